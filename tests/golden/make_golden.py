"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

The reference holds no golden vectors for this path, and it cannot run here, so
these are produced by the build's own oracle (itself pinned analytically in
tests/test_oracle_pins.py).  They exist so that (a) later edits cannot silently move the oracle
and (b) the GPU path can be checked on the GPU box against data that was frozen on another
machine.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from volumetricraytracer_amd import workloads as scenes  # noqa: E402
import volumetricraytracer_amd as v  # noqa: E402

CASES = {
    # name: (scene builder, kwargs, width, height, max_steps, shadow)
    "config2_64x36": (scenes.config2_sphere, dict(resolution=6, env=16), 64, 36, 128, False),
    "config3_96x54": (scenes.config3_torus, dict(resolution=6, env=16), 96, 54, 255, True),
    "config5_96x54": (scenes.config5_instances, dict(resolution=5, env=16), 96, 54, 255, True),
    # round 2: Voxelizer shell volume (two-level empty-space table), as fp32 and as the reference's 16-bit texel
    "config3vox_96x54": (scenes.config3_voxelized, dict(resolution=6, env=16), 96, 54, 255, True),
    "config3vox_texel16_96x54": (scenes.config3_voxelized, dict(resolution=6, env=16, device_format=1), 96, 54, 255, True),
}


def build_case(case):
    fn, kw, w, h, ms, shadow = case
    sc = fn(**kw)
    p = v.default_params(w, h, scenes.min_cell(sc), ms, shadow=shadow)
    return sc, p


def render_case(case):
    from oracle.binding import OracleScene

    sc, p = build_case(case)
    return OracleScene(sc).render(p)


if __name__ == "__main__":
    for name, case in CASES.items():
        img, st = render_case(case)
        stats = np.array([st[k] for k in ("primary_rays", "shadow_rays", "primary_steps", "shadow_steps", "hits")],
                         dtype=np.int64)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), image=img.astype(np.float32), stats=stats)
        print(name, img.shape, st)
