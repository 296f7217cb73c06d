#!/usr/bin/env python3
"""Interleaved A/B of march-kernel variants in ONE process (guide §5.4 rule 24): data paths and
blockIdx→tile maps, kernel time from the hipEvent pairs around each launch.
Usage: python tools/ab_variants.py [workload] [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from volumetricraytracer_amd import workloads as scenes  # noqa: E402
import volumetricraytracer_amd as v  # noqa: E402
from volumetricraytracer_amd import _abi  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "c3"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sc, W, H, max_steps, shadow, label = bench.build_workload(workload)
variants = {}
for pname, path in (("dense", _abi.PATH_DENSE), ("brick", _abi.PATH_BRICK), ("lds", _abi.PATH_BRICK_LDS)):
    for mname, m in (("supertile", 0), ("band", 1), ("linear", 2)):
        p = v.default_params(W, H, scenes.min_cell(sc), max_steps, shadow=shadow, path=path)
        p.flags = m
        variants[f"{pname}/{mname}"] = p

r = v.VHipRenderer()
assert r.Start()
r.SetSceneToRender(sc)
r.ResizeRenderOutput(W, H)
r.SyncWithScene()
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
times = {k: [] for k in variants}
ref = None
for rd in range(rounds + 1):
    for k, p in variants.items():
        for _ in range(3):
            r.render_rows(p, 0, H, out.data_ptr(), stream)
        torch.cuda.synchronize()
        if rd > 0:
            times[k].extend(r.timing_history(3))
        elif ref is None:
            ref = out.cpu().numpy().copy()
        else:
            assert np.array_equal(out.cpu().numpy(), ref), f"{k}: pixels differ between variants"
t = r.last_timing()
print(f"workload {label}: {t['primary_rays'] + t['shadow_rays']} rays, {t['primary_steps'] + t['shadow_steps']} samples")
for k, ts in sorted(times.items(), key=lambda kv: np.median(kv[1])):
    print(f"{k:18s} median {np.median(ts) * 1e3:8.1f} us   min {np.min(ts) * 1e3:8.1f} us")
r.Stop()
