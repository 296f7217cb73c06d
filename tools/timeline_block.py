#!/usr/bin/env python3
"""Diagnostic timeline of ONE fused march launch (VRT_FLAG_DIAG_TIMELINE build, vrt_render_block): when every wave of every frame
of the block ran.  Prints the occupancy over the launch, when each frame's first / last wave started and ended (how the frames
overlap inside the launch), and the tail.  The diagnostic build stamps every loop iteration and is slower than the product kernel.
Usage: python tools/timeline_block.py [workload] [frames per launch]"""
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
G = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sc, W, H, max_steps, shadow, label = bench.build_workload(workload)
p = v.default_params(W, H, scenes.min_cell(sc), max_steps, shadow=shadow)
p.flags |= _abi.FLAG_DIAG_TIMELINE
r = v.VHipRenderer()
assert r.Start()
r.SetSceneToRender(sc)
r.ResizeRenderOutput(W, H)
r.SyncWithScene()
cams = scenes.orbit_cameras(sc, G)
out = torch.empty((G, H, W, 4), dtype=torch.float32, device="cuda:0")
for _ in range(3):
    r.render_block(p, G, out.data_ptr(), H * W * 16, 0, cameras=cams)
torch.cuda.synchronize()
ms, fr = r.launch_history(1)[0]
rec = r.wave_records(2).reshape(G, -1, 8)
dg = r.wave_records(3).reshape(G, -1, 8)
r.Stop()
live = rec[:, :, 0] > 0
start, end = dg[:, :, 0].astype(np.int64), dg[:, :, 1].astype(np.int64)
t0 = start[live].min()
s_us, e_us = (start - t0) / 100.0, (end - t0) / 100.0
span = e_us[live].max()
print(f"{label}: ONE launch of {fr} frames, {ms * 1e3:.1f} us by its event pair (diagnostic build), span of the wave stamps {span:.1f} us = {span / G:.1f} us per frame")
dur = (e_us - s_us)[live]
print(f"waves {live.sum()}  wave-time {dur.sum() / 1e3:.1f} ms -> mean occupancy {dur.sum() / span / 256:.1f} of 32 waves per CU over the launch")
nb = 24
edges = np.linspace(0, span, nb + 1)
occ = [(np.minimum(e_us[live], edges[i + 1]) - np.maximum(s_us[live], edges[i])).clip(min=0).sum() / (edges[i + 1] - edges[i]) / 256 for i in range(nb)]
print("occupancy (waves/CU) per %.1f-us slice: %s" % (edges[1], " ".join(f"{o:.1f}" for o in occ)))
print("frame: first wave starts / last wave starts / last wave ends (us)   frames in flight at its first start")
for f in range(G):
    m = live[f]
    fs, ls, le = s_us[f][m].min(), s_us[f][m].max(), e_us[f][m].max()
    inflight = sum(1 for g in range(G) if s_us[g][live[g]].min() <= fs < e_us[g][live[g]].max())
    print(f"  {f:3d}: {fs:8.1f} {ls:8.1f} {le:8.1f}    {inflight}")
last = e_us[G - 1][live[G - 1]].max()
prev = max(e_us[f][live[f]].max() for f in range(G - 1)) if G > 1 else 0.0
tail_start = np.percentile(s_us[live], 99.9)
print(f"tail: the launch's last wave START at {s_us[live].max():.1f} us, END at {span:.1f} us: {span - s_us[live].max():.1f} us with nothing left to dispatch")
