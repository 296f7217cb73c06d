#!/usr/bin/env python3
"""Benchmark of the ray-march hot path (BASELINE.json metric: Mrays/s + ms/frame at 1080p over a
256^3 SDF volume, 1/2/4/8 MI355X).

A "step" is one full frame.  With N ranks (one process per GPU, launched by torch.distributed.run)
the frame is cut into 32-row strips dealt round-robin to the ranks (interleaved row tiles: contiguous
tiles would put every object row on the middle GPUs); every rank marches its strips into a compact
device tile with ONE launch and the tiles are gathered onto rank 0 with one RCCL gather
(torch.distributed backend "nccl") that overlaps with the next frame's march (two tile buffers);
rank 0 un-shuffles the gathered strips into frame order.  The exchange format is R8G8B8A8_UNORM, the
reference's own back-buffer precision (DXConstants.cpp:21): at 16 B/pixel rank 0's seven inbound
xGMI links, not the march, would set the frame time.

Scaling is WEAK by default: the per-GPU ray count is fixed at the 1080p frame of the metric and the
frame grows with N at constant 16:9 aspect (N=1 1920x1080, N=2 2715x1527, N=4 3840x2160 = config 4's
frame, N=8 5431x3055), same camera, same scene.  A 1080p frame is ~0.19 ms of GPU work, most of it
the latency-bound tail of a few hundred grazing rays, so splitting THAT frame 8 ways
cannot scale; `--scaling strong --workload c4` gives the fixed-4K-frame split of config 4.

Prints ONE JSON line on rank 0 (see the driver contract), extended with "roofline" and
"cpu_baseline".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E (spec)


def build_workload(name: str):
    """Returns (scene, width, height, max_steps, shadow, label)."""
    import scenes

    if name == "c3":
        sc = scenes.bench_config3()
        return sc, 1920, 1080, 255, True, "config3: 256^3 voxelized glTF-style mesh (torus, 16384 triangles), 1920x1080, shadow ray on"
    if name == "c3sdf":
        sc = scenes.config3_torus(8, 256, distance=190.0)
        return sc, 1920, 1080, 255, True, "config3 (analytic SDF variant): 256^3 torus SDF, 1920x1080, shadow ray on"
    if name == "c2":
        sc = scenes.config2_sphere(6, 256)
        return sc, 1280, 720, 128, False, "config2: 64^3 SDF sphere, 1280x720, 128 max steps"
    if name == "c4":
        sc = scenes.bench_config3()
        return sc, 3840, 2160, 255, True, "config4: 256^3 voxelized mesh, 3840x2160, row tiles + RCCL gather"
    if name == "c5":
        sc = scenes.config5_instances(7, 256)
        return sc, 1920, 1080, 255, True, "config5: 8 instanced 128^3 volumes + skybox, 1920x1080, AABB BVH"
    raise SystemExit(f"unknown workload {name}")


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c3sdf", "c4", "c5"])
    ap.add_argument("--path", default="auto", choices=["auto", "dense", "brick", "lds"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N>1: weak = frame grows with N (fixed rays per GPU); strong = the workload's own frame split N ways")
    ap.add_argument("--output", default="auto", choices=["auto", "f32", "rgba8"],
                    help="tile pixel format; auto = float4 on one GPU, RGBA8 (the exchange format) on several")
    ap.add_argument("--strip-rows", type=int, default=32, help="N>1: rows per interleaved strip; 0 = contiguous row tiles")
    ap.add_argument("--frames-in-flight", type=int, default=2, choices=[1, 2, 3, 4],
                    help="frames launched before the first one must have finished, each on its own HIP stream and tile buffer "
                         "(the reference keeps 3 in flight, DXConstants.cpp:23)")
    ap.add_argument("--tile-map", default="supertile", choices=["supertile", "band", "linear"], help="blockIdx -> tile map (speed only)")
    ap.add_argument("--skip-empty", action="store_true",
                    help="VRT_FLAG_SKIP_EMPTY: no samples in bricks the leap table declares empty (same pixels, fewer samples "
                         "and therefore fewer algorithmic bytes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target CPU time of the baseline sample")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import scenes
    import volumetricraytracer_amd as v
    from volumetricraytracer_amd import _abi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus={args.gpus}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # VRT_BENCH_BACKEND=gloo: rehearsal of the N>1 code path on a ONE-GPU box (every rank marches on cuda:0,
    # tiles are staged through host memory and gathered over gloo).  Never a measurement.
    rehearsal = os.environ.get("VRT_BENCH_BACKEND", "nccl") == "gloo"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    sc, W, H, max_steps, shadow, label = build_workload(args.workload)
    if world > 1 and args.scaling == "weak":
        W, H = int(round(W * world ** 0.5)), int(round(H * world ** 0.5))
        label += f" -- weak-scaled to {W}x{H} for {world} GPUs"
    rgba8 = args.output == "rgba8" or (args.output == "auto" and world > 1)
    strip_rows = args.strip_rows if world > 1 else 0
    path = {"auto": _abi.PATH_AUTO, "dense": _abi.PATH_DENSE, "brick": _abi.PATH_BRICK, "lds": _abi.PATH_BRICK_LDS}[args.path]
    p = v.default_params(W, H, scenes.min_cell(sc), max_steps, shadow=shadow, path=path)
    if rgba8:
        p.flags |= _abi.FLAG_OUTPUT_RGBA8
    p.flags |= {"supertile": 0, "band": 1, "linear": 2}[args.tile_map]
    if args.skip_empty:
        p.flags |= _abi.FLAG_SKIP_EMPTY

    r = v.VHipRenderer(devices=(local_rank,))
    if not r.Start():
        raise SystemExit("VHipRenderer.Start() failed")
    r.SetSceneToRender(sc)
    r.ResizeRenderOutput(W, H)
    r.SyncWithScene()

    from volumetricraytracer_amd.tiles import FrameGather

    # tile of this rank + (rank 0) the gathered frames
    pix = torch.uint8 if rgba8 else torch.float32
    # Frames in flight: K tile buffers, K HIP streams.  A frame's last ~90 us are a few hundred latency-bound waves
    # on an otherwise idle chip; the next frame's march fills it.  Within a buffer, frame i+K follows
    # frame i in stream order, so a tile is never overwritten before it has been consumed.
    K = args.frames_in_flight
    fg = FrameGather(H, W, world, rank, torch.device("cpu") if rehearsal else dev, dtype=pix, buffers=K, strip_rows=strip_rows)
    row0, rows = fg.row0, fg.rows
    march_tiles = [torch.zeros_like(x, device=dev) for x in fg.tiles] if rehearsal else fg.tiles
    pending = [None] * K
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(K - 1)]
    stream = streams[0]

    # Rank 0 un-shuffles the gathered strips on a stream of its own, so that the copy does not sit between two frames of
    # a march stream: it only has to be finished before the NEXT gather into the same frame buffer starts.
    unshuffle = world > 1 and rank == 0 and strip_rows > 0
    copy_stream = torch.cuda.Stream(device=dev) if unshuffle else None
    unshuffled = [None] * K  # event: frames[b] has been copied out and may be overwritten

    def step(i: int) -> None:
        b = i % K
        with torch.cuda.stream(streams[b]):
            if pending[b] is not None:
                pending[b].wait()  # tile buffer b is free again (its gather finished); stream b waits, not the host
                pending[b] = None
            if strip_rows > 0:
                r.render_strips(p, strip_rows, rank, world, fg.strips_per, march_tiles[b].data_ptr(), streams[b].cuda_stream)
            else:
                r.render_rows(p, row0, rows, march_tiles[b].data_ptr(), streams[b].cuda_stream)
            if rehearsal:
                fg.tiles[b].copy_(march_tiles[b])
            if world > 1:
                if unshuffled[b] is not None:
                    streams[b].wait_event(unshuffled[b])
                pending[b] = fg.gather(b, async_op=True)  # RCCL gather over xGMI, overlaps the following frames' march
        if unshuffle:
            with torch.cuda.stream(copy_stream):
                pending[b].wait()  # the copy stream (not the host) waits for this gather (gloo rehearsal: the host does)
                fg.unshuffle(b)    # gathered [rank, strip] order -> frame order, one strided device copy
                unshuffled[b] = torch.cuda.Event()
                unshuffled[b].record(copy_stream)

    def drain() -> None:
        for b in range(K):
            with torch.cuda.stream(streams[b]):
                if pending[b] is not None:
                    pending[b].wait()
                    pending[b] = None
        torch.cuda.synchronize()

    def barrier() -> None:
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    drain()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        cdev = torch.device("cpu") if rehearsal else dev
        te = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    t = r.last_timing()  # this rank's tile, last frame (every frame is identical)
    kms = r.timing_history(min(args.steps, 200))
    verified = None
    if os.environ.get("VRT_BENCH_VERIFY") and rank == 0 and args.steps > 0:
        # the gathered (and un-shuffled) frame must be the frame one GPU renders alone, bit for bit
        whole = torch.empty((H, W, 4), dtype=pix, device=dev)
        r.render_rows(p, 0, H, whole.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        got = fg.frame((args.steps - 1) % K)
        verified = bool(torch.equal(got.cpu(), whole.cpu()))
        if not verified:
            raise SystemExit("[bench] gathered frame differs from the single-GPU frame")
        t = dict(t)  # keep the tile's counters (the verification launch overwrote last_timing)
    counts = torch.tensor([t["primary_rays"], t["shadow_rays"], t["primary_steps"], t["shadow_steps"], t["hits"]],
                          dtype=torch.float64, device=torch.device("cpu") if rehearsal else dev)
    if world > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    primary, shadow_rays, psteps, ssteps, hits = [float(x) for x in counts.tolist()]
    rays_per_frame = primary + shadow_rays
    ms_per_step = elapsed / max(args.steps, 1) * 1e3
    value = rays_per_frame * args.steps / elapsed / 1e6 if args.steps > 0 else 0.0

    if rank == 0:
        # roofline of the march kernel on THIS rank's tile: algorithmic bytes / mean
        # kernel time from the hipEvent pairs recorded on the launch stream around every launch
        alg_bytes = v.algorithmic_bytes(t, 4 if rgba8 else 16)
        k_ms = float(np.mean(kms)) if kms else float("nan")
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if kms else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == args.workload and tj.get("n_gpus") == world:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {
            "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "kernel": "march_kernel", "kernel_ms": round(k_ms, 4), "algorithmic_bytes_per_launch": int(alg_bytes),
            "samples_per_launch": int(t["primary_steps"] + t["shadow_steps"]),
        }
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(sc, p, args.cpu_seconds)
        out = {
            "metric": "Mrays/sec at 1080p, 256^3 SDF volume" if args.workload in ("c3", "c3sdf") else "Mrays/sec",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": label, "width": W, "height": H, "volume": f"{sc.volumes()[0].N - 1}^3 cells",
                       "max_steps": max_steps, "shadow": bool(shadow), "data_path": args.path,
                       "output": "rgba8 (R8G8B8A8_UNORM tiles; march and shading in f32)" if rgba8 else "f32 (float4)",
                       "frames_in_flight": K, "skip_empty": bool(args.skip_empty),
                       "parallelism": ("1 GPU" if world == 1 else
                                       (f"{strip_rows}-row interleaved strips" if strip_rows else "contiguous row tiles") +
                                       f" x{world} + " + ("gloo gather, REHEARSAL on one GPU" if rehearsal else "RCCL gather to rank 0")),
                       "rays_per_frame": int(rays_per_frame), "samples_per_ray": round((psteps + ssteps) / max(rays_per_frame, 1), 2)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if verified is not None:
            out["gathered_frame_equals_single_gpu_frame"] = verified
        print(json.dumps(out), flush=True)

    r.Stop()
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(sc, p, target_seconds: float):
    """The scalar oracle (kind "port") on this host's cores, on a bounded sample of the same
    workload: whole frames of the same scene and parameters, repeated until about
    `target_seconds` of wall time have passed (reduced resolution if one frame alone would take
    longer; Mrays/s is resolution-independent to first order)."""
    from oracle.binding import OracleScene
    from volumetricraytracer_amd import _abi

    cores = os.cpu_count() or 1
    o = OracleScene(sc)
    q = _abi.vrt_params.from_buffer_copy(p)
    # calibrate on 1/64 of the pixels
    q.width, q.height = max(p.width // 8, 1), max(p.height // 8, 1)
    t0 = time.perf_counter()
    _, st = o.render(q, threads=cores)
    dt = max(time.perf_counter() - t0, 1e-4)
    rate = (st["primary_rays"] + st["shadow_rays"]) / dt
    scale = min(1.0, (rate * target_seconds / max(p.width * p.height, 1)) ** 0.5)
    q.width, q.height = max(int(p.width * scale), 1), max(int(p.height * scale), 1)
    rays, frames = 0, 0
    t0 = time.perf_counter()
    while True:
        _, st = o.render(q, threads=cores)
        rays += st["primary_rays"] + st["shadow_rays"]
        frames += 1
        dt = time.perf_counter() - t0
        if dt >= target_seconds or frames >= 10000:
            break
    q1 = _abi.vrt_params.from_buffer_copy(p)
    q1.width, q1.height = max(p.width // 4, 1), max(p.height // 4, 1)
    t1 = time.perf_counter()
    _, st1 = o.render(q1, threads=1)
    dt1 = time.perf_counter() - t1
    return {
        "value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": f"{frames} frame(s) of the same scene and march parameters at {q.width}x{q.height} "
                  f"({rays} rays, {dt:.1f} s wall on {cores} threads)",
        "single_thread_value": round((st1["primary_rays"] + st1["shadow_rays"]) / dt1 / 1e6, 3),
        "single_thread_sample": f"one {q1.width}x{q1.height} frame, {dt1:.1f} s",
    }


if __name__ == "__main__":
    main()
