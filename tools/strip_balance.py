#!/usr/bin/env python3
"""Per-rank march time and work of the multi-GPU frames, measured on ONE GPU by launching every rank's tile in turn (the
tiles are independent; only the gather is missing).  Shows the load balance of interleaved strips against contiguous row
tiles: the SAME frame split N ways (strong scaling, the default) or, with a third argument "weak", the frame grown with N.
Usage: python tools/strip_balance.py [workload] [strip_rows] [weak]"""
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
from volumetricraytracer_amd.tiles import strip_layout, tile_rows  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "c3"
strip_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 8
weak = len(sys.argv) > 3 and sys.argv[3] == "weak"
sc, W0, H0, max_steps, shadow, label = bench.build_workload(workload)
r = v.VHipRenderer()
assert r.Start()
r.SetSceneToRender(sc)
r.SyncWithScene()
stream = torch.cuda.current_stream().cuda_stream
print(label + f"; strips of {strip_rows} rows; " + ("weak" if weak else "strong") + " scaling")
for world in (1, 2, 4, 8):
    W, H = (int(round(W0 * world ** 0.5)), int(round(H0 * world ** 0.5))) if weak else (W0, H0)
    p = v.default_params(W, H, scenes.min_cell(sc), max_steps, shadow=shadow)
    p.flags |= _abi.FLAG_OUTPUT_RGBA8
    _, per = strip_layout(H, world, strip_rows)
    tile = torch.zeros((max(per * strip_rows, (H + world - 1) // world), W), dtype=torch.int32, device="cuda:0")
    res = {}
    for mode in ("strips", "rows"):
        ms, rays, samples = [], [], []
        for g in range(world):
            for _ in range(6):
                if mode == "strips":
                    r.render_strips(p, strip_rows, g, world, per, tile.data_ptr(), stream)
                else:
                    _, row0, rows = tile_rows(H, world, g)
                    r.render_rows(p, row0, rows, tile.data_ptr(), stream)
            torch.cuda.synchronize()
            ms.append(float(np.mean(r.timing_history(5))))
            t = r.last_timing()
            rays.append(t["primary_rays"] + t["shadow_rays"])
            samples.append(t["primary_steps"] + t["shadow_steps"])
        res[mode] = (ms, rays, samples)
        print(f"  N={world} {W}x{H} {mode:6s}: kernel us per rank " + " ".join(f"{m*1e3:.0f}" for m in ms) +
              f" | max {max(ms)*1e3:.0f} | rays/rank {min(rays)}..{max(rays)} | samples/rank max/mean {max(samples)/(sum(samples)/len(samples)):.3f}"
              f" | march-bound frame rate {sum(rays)/max(ms)/1e6:.1f} Grays/s")
r.Stop()
