#!/usr/bin/env python3
"""Upper bound of what a per-tile beam pre-pass could gain.

A beam pre-pass marches ONE wave-uniform cone per 8x8 tile through the empty space in front of the surface and hands the 64
lanes a common, later start.  Whatever its design, it cannot save more than ALL the positions a camera ray skips before its
first sample, at no cost of its own.  This probe measures exactly that bound on the chip: the A/B library
(lib/ab_tstart.so, -DVRT_AB_TSTART) first RECORDS, per lane, the ray parameter of the camera ray's first sampled position,
then renders the same frames again with every camera ray STARTED there (its leading skips removed, the hand-off costing one
4-byte load per lane).  Same frames with and without, interleaved, block launches (throughput) and lone frames (latency).
Run:  VRT_LIB=volumetricraytracer_amd/lib/ab_tstart.so python tools/beam_upper_bound.py [c3|cover]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import volumetricraytracer_amd as v  # noqa: E402
from volumetricraytracer_amd import _abi, workloads  # noqa: E402

assert "ab_tstart" in os.environ.get("VRT_LIB", ""), "run with VRT_LIB=.../ab_tstart.so"
RECORD, USE = 1 << 20, 2 << 20
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
sc, W, H, ms, sh, label = bench.build_workload("c3")
if which == "cover":
    ext = float(sc.volumes()[0].VolumeExtends)
    sc = workloads.config3_voxelized(8, 256, distance=0.6 * ext)
    label = "config 3's volume, camera inside its box (every wave marches)"
B = 96
r = v.VHipRenderer()
assert r.Start()
r.SetSceneToRender(sc)
r.ResizeRenderOutput(W, H)
r.SyncWithScene()
cams = r.camera_array(workloads.orbit_cameras(sc, B))
buf = torch.empty((B, H, W, 4), dtype=torch.float32, device="cuda")


def params(flag):
    p = v.default_params(W, H, workloads.min_cell(sc), ms, shadow=sh)
    p.flags |= flag
    return p


def block(flag, n=B, start=0):
    r.render_block(params(flag), n, buf.data_ptr(), H * W * 16, 0, cameras=(cams, start))


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print(label)
# throughput: one launch of 96 frames
block(RECORD)
torch.cuda.synchronize()
ref = buf[B // 2].clone()
res = {"base": [], "lead_skips_free": []}
for rd in range(4):
    res["base"].append(timed(lambda: block(0), 5))
    res["lead_skips_free"].append(timed(lambda: block(USE), 5))
block(USE)
torch.cuda.synchronize()
diff = (buf[B // 2] - ref).abs().max().item()
for k, x in res.items():
    print(f"  block of {B} frames   {k:16s} {np.median(x) / B * 1e3:7.2f} us/frame  (rounds {[round(y / B * 1e3, 2) for y in x]})")
print(f"  upper bound of a beam pre-pass on throughput: {100 * (1 - np.median(res['lead_skips_free']) / np.median(res['base'])):.1f} %   (max |pixel difference| {diff:.2e})")
# latency: one frame per launch, waited for
f0 = B // 2
block(RECORD, 1, f0)
torch.cuda.synchronize()
lat = {"base": [], "lead_skips_free": []}
for rd in range(4):
    for k, flag in (("base", 0), ("lead_skips_free", USE)):
        ts = []
        for _ in range(30):
            block(flag, 1, f0)
            torch.cuda.synchronize()
            ts.append(r.launch_history(1)[0][0])
        lat[k].append(float(np.median(ts)))
for k, x in lat.items():
    print(f"  lone frame (kernel)   {k:16s} {np.median(x) * 1e3:7.2f} us")
print(f"  upper bound of a beam pre-pass on the lone frame: {100 * (1 - np.median(lat['lead_skips_free']) / np.median(lat['base'])):.1f} %")
r.Stop()
