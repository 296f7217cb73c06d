#!/usr/bin/env python3
"""Diagnostic timeline of one march_kernel launch (VRT_FLAG_DIAG_TIMELINE build): when and where
every wave ran.  Prints occupancy over time, per-XCD finish times, wave-duration histogram.
Usage: python tools/timeline.py [workload] [tile_map]"""
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
tile_map = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if workload.startswith("torus"):  # torus6 / torus7 / torus8: the config-3 scene at another grid resolution
    res = int(workload[5:])
    sc, W, H, max_steps, shadow, label = scenes.config3_torus(res, 256, distance=190.0), 1920, 1080, 255, True, f"torus SDF at 2^{res} cells"
else:
    sc, W, H, max_steps, shadow, label = bench.build_workload(workload)
path = {"auto": 0, "dense": 1, "brick": 2, "lds": 3}[os.environ.get("VRT_PATH", "auto")]
p = v.default_params(W, H, scenes.min_cell(sc), max_steps, shadow=shadow, path=path)
p.flags = tile_map | _abi.FLAG_DIAG_TIMELINE
r = v.VHipRenderer()
assert r.Start()
r.SetSceneToRender(sc)
r.ResizeRenderOutput(W, H)
r.SyncWithScene()
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
for _ in range(5):
    r.render_rows(p, 0, H, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
t = r.last_timing()
rec = r.wave_records(0)
dg = r.wave_records(1)
r.Stop()
live = rec[:, 0] > 0  # waves that own at least one pixel
start = dg[:, 0].astype(np.int64)
end = dg[:, 1].astype(np.int64)
t0 = start[live].min()
s_us = (start - t0) / 100.0
e_us = (end - t0) / 100.0
dur = e_us - s_us
xcc = dg[:, 3] & 0xF
maxit = dg[:, 4].astype(np.int64)
mem_cyc = dg[:, 5].astype(np.int64)
loop_cyc = dg[:, 6].astype(np.int64)
iters = dg[:, 7].astype(np.int64)
fetches = dg[:, 2].astype(np.int64)
span = e_us[live].max()
print(f"{label}: kernel {t['kernel_ms']*1e3:.1f} us (event, diagnostic build), waves {len(rec)} (live {live.sum()}), span {span:.1f} us")
print("wave duration us: mean %.2f  p50 %.2f  p90 %.2f  p99 %.2f  max %.2f" % (dur[live].mean(), *np.percentile(dur[live], [50, 90, 99]), dur[live].max()))
steps = (rec[:, 3] + rec[:, 4]).astype(np.int64)
for lo, hi in ((0, 1), (1, 200), (200, 1000), (1000, 3000), (3000, 10**9)):
    m = live & (steps >= lo) & (steps < hi)
    if m.any():
        print(f"  waves with {lo:5d}<=samples<{hi:<10d}: {m.sum():6d}  mean dur {dur[m].mean():7.2f} us  sum {dur[m].sum()/1e3:8.2f} ms")
print("wave-time total %.1f ms -> mean occupancy %.1f waves/CU over the span" % (dur[live].sum() / 1e3, dur[live].sum() / span / 256))
nb = 20
edges = np.linspace(0, span, nb + 1)
occ = [(np.minimum(e_us[live], edges[i + 1]) - np.maximum(s_us[live], edges[i])).clip(min=0).sum() / (edges[i + 1] - edges[i]) / 256 for i in range(nb)]
print("occupancy (waves/CU) per %.1f-us slice: %s" % (edges[1], " ".join(f"{o:.1f}" for o in occ)))
for x in range(8):
    m = live & (xcc == x)
    if m.any():
        print(f"  XCC {x}: waves {m.sum():6d}  last end {e_us[m].max():7.1f} us  wave-time {dur[m].sum()/1e3:7.2f} ms  samples {steps[m].sum()}")
m = live & (iters > 60)
if m.any():
    print("waves with >60 loop iterations: %d" % m.sum())
    print("  cycles per loop iteration (stamped region): mean %.0f  p10 %.0f  p90 %.0f" % ((loop_cyc[m] / iters[m]).mean(), *np.percentile(loop_cyc[m] / iters[m], [10, 90])))
    print("  of which tap fetches:                       mean %.0f  p10 %.0f  p90 %.0f" % ((mem_cyc[m] / iters[m]).mean(), *np.percentile(mem_cyc[m] / iters[m], [10, 90])))
    print("  wave us per iteration (wall):               mean %.3f" % (dur[m] / iters[m]).mean())
    print("  iterations whose taps came back within 450 cycles (all lanes hit L1/L2): %.1f %%" % (100.0 * fetches[m].sum() / iters[m].sum()))
    tail = m & (e_us > 0.6 * span)
    if tail.any():
        print("  waves ending in the tail (%d): cycles/iter %.0f, load+interp %.0f, us/iter %.3f" % (
            tail.sum(), (loop_cyc[tail] / iters[tail]).mean(), (mem_cyc[tail] / iters[tail]).mean(), (dur[tail] / iters[tail]).mean()))
late = np.argsort(-np.where(live, e_us, -1))[:6]
print("latest-finishing waves: " + ", ".join(f"(start {s_us[i]:.1f}, dur {dur[i]:.1f}, samples {steps[i]}, chain {maxit[i]}, iters {iters[i]})" for i in late))
