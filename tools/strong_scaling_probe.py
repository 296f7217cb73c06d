#!/usr/bin/env python3
"""What ONE rank of an N-way strong-scaled frame can sustain: rank 0's interleaved strips of the config-3 frame (every N-th
32-row strip, RGBA8 tile) rendered with K frames in flight (K streams, blocks of 8 frames per vrt_render_block call), no gather — the per-rank march rate that bounds the N-GPU job.
Run once per GPU_MAX_HW_QUEUES setting (the HIP runtime reads it at start-up).
Usage: [GPU_MAX_HW_QUEUES=8] python tools/strong_scaling_probe.py [workload]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
import volumetricraytracer_amd as v  # noqa: E402
from volumetricraytracer_amd import _abi, workloads  # noqa: E402
from volumetricraytracer_amd.tiles import strip_layout  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "c3"
sc, W, H, max_steps, shadow, label = bench.build_workload(workload)
p = v.default_params(W, H, workloads.min_cell(sc), max_steps, shadow=shadow)
p.flags |= _abi.FLAG_OUTPUT_RGBA8
r = v.VHipRenderer()
assert r.Start()
r.SetSceneToRender(sc)
r.ResizeRenderOutput(W, H)
r.SyncWithScene()
print(f"{label}; GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}; strips of {os.environ.get('PROBE_STRIP_ROWS', '32')} rows")
print("ranks  K  ms/frame(rank 0 alone)  frames/s  -> whole-job Grays/s if every rank keeps that rate")
full = None
SR = int(os.environ.get("PROBE_STRIP_ROWS", "32"))  # rows per interleaved strip
NS = [int(x) for x in os.environ.get("PROBE_RANKS", "1,2,4,8").split(",")]
KS = [int(x) for x in os.environ.get("PROBE_FRAMES_IN_FLIGHT", "1,2,3,4,6,8").split(",")]
for n in NS:
    _, per = strip_layout(H, n, SR)
    for K in KS:
        streams = [torch.cuda.Stream() for _ in range(K)]
        G = int(os.environ.get("PROBE_BLOCK_FRAMES", "8"))  # frames per vrt_render_block call, like bench.py with N > 1
        tiles = [torch.zeros((G, per * SR, W, 4), dtype=torch.uint8, device="cuda:0") for _ in range(K)]

        def run(steps):
            for i in range(0, steps, G):
                b = (i // G) % K
                r.render_block(p, min(G, steps - i), tiles[b].data_ptr(), per * SR * W * 4, streams[b].cuda_stream, strips=(SR, 0, n, per))
            torch.cuda.synchronize()

        run(2 * K * G)
        steps, dt = 4 * K * G if 4 * K * G > 304 else 304, 1e9
        for _ in range(3):  # best of three
            t0 = time.perf_counter()
            run(steps)
            dt = min(dt, (time.perf_counter() - t0) / steps)
        if full is None:
            if n != 1:
                r.render_strips(p, 32, 0, 1, strip_layout(H, 1, 32)[1], torch.zeros((strip_layout(H, 1, 32)[1] * 32, W, 4), dtype=torch.uint8, device="cuda:0").data_ptr(), 0)
                torch.cuda.synchronize()
            t = r.last_timing()
            full = t["primary_rays"] + t["shadow_rays"]  # n = 1: the whole frame's rays
        print(f"  {n}    {K}   {dt * 1e3:.4f}   {1 / dt:9.0f}   {full / dt / 1e9:7.2f}")
r.Stop()
