"""Not collected by pytest (by hand on the GPU box: `python tests/soak_voxelizer.py 0 200`): random triangle soups — from a
handful of large triangles to thousands of tiny, thin, near-degenerate and duplicated ones — through the device Voxelizer
(vrt_voxelize_mesh) and through the C++ CPU converter: every density and material bit for bit, the same triangles skipped."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import volumetricraytracer_amd as v  # noqa: E402
from volumetricraytracer_amd import voxelizer as vx  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
r = v.VHipRenderer()
assert r.Start()
for seed in range(lo, hi):
    rng = np.random.RandomState(seed)
    kind = seed % 4
    n_tri = int(rng.choice([3, 20, 200, 3000]))
    if kind == 0:    # big triangles anywhere in the unit box
        tri = rng.uniform(-0.5, 0.5, (n_tri, 3, 3))
    elif kind == 1:  # tiny triangles (smaller than a cell) scattered about
        c = rng.uniform(-0.5, 0.5, (n_tri, 1, 3))
        tri = c + rng.uniform(-0.004, 0.004, (n_tri, 3, 3))
    elif kind == 2:  # needles: two vertices nearly coincide; some exactly degenerate
        a = rng.uniform(-0.5, 0.5, (n_tri, 1, 3))
        b = rng.uniform(-0.5, 0.5, (n_tri, 1, 3))
        tri = np.concatenate([a, a + rng.uniform(-1e-4, 1e-4, (n_tri, 1, 3)), b], axis=1)
        tri[::7, 1] = tri[::7, 0]
    else:            # a closed mesh with duplicated faces
        pos, _, idx = vx.torus_mesh(0.4, 0.15, int(rng.randint(8, 40)), int(rng.randint(6, 20)))
        tri = pos[idx.reshape(-1, 3)]
        tri = np.concatenate([tri, tri[:: 5]], axis=0)
    pos = np.ascontiguousarray(tri.reshape(-1, 3), np.float32)
    idx = np.arange(len(pos), dtype=np.uint32)
    resolution = int(rng.choice([4, 5, 6, 7]))
    p, be = vx.importer_space(pos)
    cpu = vx.convert_mesh(p, idx, be, f"soak_{resolution}")
    skipped = r.voxelize_mesh(2, p, idx, resolution, cpu.VolumeExtends)
    gpu = r.download_volume(2, resolution, cpu.VolumeExtends)
    same = np.array_equal(gpu.density, cpu.density) and np.array_equal(gpu.material_id, cpu.material_id)
    if not same:
        bad.append(seed)
        d = gpu.density != cpu.density
        print(f"seed {seed}: kind {kind} triangles {len(idx) // 3} resolution {resolution}: {int(d.sum())} voxels differ, "
              f"max |diff| {float(np.abs(gpu.density - cpu.density)[d].max()) if d.any() else 0.0}", flush=True)
    if seed % 20 == 0:
        print(f"... seed {seed} (device skipped {skipped})", flush=True)
r.Stop()
print(f"seeds {lo}..{hi - 1}: {len(bad)} mismatches {bad}")
sys.exit(1 if bad else 0)
