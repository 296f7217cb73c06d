"""Not collected by pytest (by hand on the GPU box: `python tests/soak_random_scenes.py 96 600`): the random-scene fuzz of
test_parity_gpu.py::test_random_scenes_parity over a seed range of one's choosing; prints every seed whose frame or counters
differ from the oracle's.  With a third argument "mix" every seed also draws a data path (auto / dense / brick / LDS /
cells), a device format (fp32 / the reference's texel) and an over-relaxation factor of its own."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import volumetricraytracer_amd as v  # noqa: E402
from oracle.binding import OracleScene  # noqa: E402
from test_parity_gpu import STAT_KEYS, TOL, gpu_render  # noqa: E402
from volumetricraytracer_amd import workloads as scenes  # noqa: E402

from volumetricraytracer_amd import _abi  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
mix = len(sys.argv) > 3 and sys.argv[3] == "mix"
bad = []
r = v.VHipRenderer()  # ONE renderer for all scenes: volumes, textures and tables are replaced scene after scene
assert r.Start()
for seed in range(lo, hi):
    sc, p = scenes.random_scene(seed)
    what = ""
    if mix:
        rng = np.random.RandomState(seed + 77777)
        p.path = int(rng.choice([_abi.PATH_AUTO, _abi.PATH_DENSE, _abi.PATH_BRICK, _abi.PATH_BRICK_LDS, _abi.PATH_CELLS]))
        fmt = int(rng.choice([_abi.FORMAT_F32, _abi.FORMAT_TEXEL16]))
        p.k_relax = float(rng.choice([0.7, 1.0, 1.4, 1.7, 2.0]))
        for vol in sc.volumes():
            vol.set_device_format(fmt)
        what = f" path {p.path} format {fmt} k_relax {p.k_relax}"
    img, t = gpu_render(r, sc, p)
    # the full closest hit's passes form (what a block of frames runs): the same bits and counters as the one kernel
    p3 = _abi.vrt_params.from_buffer_copy(p)
    p3.flags |= _abi.FLAG_FULL_THREE_PASS
    img3, t3 = gpu_render(r, sc, p3)
    if not np.array_equal(img, img3) or any(t[k] != t3[k] for k in STAT_KEYS):
        bad.append(seed)
        print(f"seed {seed}: passes form differs from the one kernel in {np.count_nonzero(img != img3)} values", {k: (t[k], t3[k]) for k in STAT_KEYS if t[k] != t3[k]}, flush=True)
    ref, st = OracleScene(sc).render(p, threads=16)
    keys = STAT_KEYS if len(sc.Objects) == 1 else ("primary_rays", "shadow_rays", "bounce_rays", "hits")
    err = float(np.abs(img - ref).max())
    if np.isnan(img).any() or err > TOL or any(t[k] != st[k] for k in keys):
        bad.append(seed)
        print(f"seed {seed}: max err {err:.3g} mode {p.mode} objects {len(sc.Objects)}{what}", {k: (t[k], st[k]) for k in keys if t[k] != st[k]}, flush=True)
    if seed % 50 == 0:
        print(f"... seed {seed}", flush=True)
r.Stop()
print(f"seeds {lo}..{hi - 1}: {len(bad)} mismatches {bad}")
sys.exit(1 if bad else 0)
