#!/usr/bin/env python3
"""Benchmark of the ray-march hot path (BASELINE.json metric: Mrays/s + ms/frame at 1080p over a
256^3 SDF volume, 1/2/4/8 MI355X).

A "step" is one pass of the hot path over one batch of input: a batch of 96 full frames of the workload (default: BASELINE
config 3 — 1920x1080, 256^3 voxelized mesh, shadow ray on), consecutive views of a camera on a short orbit through the
workload's own view (0.25 degrees apart) — frames of a moving camera, not copies of one frame; `ms_per_frame` =
`ms_per_step` / 96 is in the line too.  The batch is rendered by vrt_render_block: ONE march launch per block of frames
(the kernel's grid has a frame axis; the dispatcher back-fills the wave slots a frame's latency-bound tail leaves empty with the
next frame's waves).  One GPU: ONE stream, ONE launch per step.

With N ranks (one process per GPU) the SAME frame is split N ways — strong scaling, as the metric and config 4 define it: the
frame is cut into 8-row strips dealt round-robin to the ranks (contiguous tiles would put every object row on the middle GPUs);
every rank marches its strips of a block of 48 frames into compact device tiles with ONE launch, then ONE RCCL collective per
block (torch.distributed backend "nccl") assembles the frames, overlapping the next block's march on a second stream; the
assembling rank un-shuffles the strips into frame order.  `--exchange rotate` (default): frame g of a block is assembled on rank
g // (24 / N) — one all-to-all per block, every xGMI link of the node in use; `--exchange gather`: every frame on rank 0
(ncclGather; its inbound links then bound the job) — the other one is measured as a leg of the same run.  The
exchange format is R8G8B8A8_UNORM, the reference's own back-buffer precision (DXConstants.cpp:21).  `--scaling weak` (frame grows
with N, fixed rays per GPU) is kept as an option; it is never the default.

Launching: `python bench.py --gpus N` starts the N ranks itself (a child `python -m torch.distributed.run`,
spawned BEFORE this process touches the GPU); under `torch.distributed.run` (WORLD_SIZE set) it is a rank.
A rank whose WORLD_SIZE differs from --gpus exits non-zero.

Prints ONE JSON line on rank 0 (driver contract) extended with:
  roofline        the contract's fields (algorithmic bytes of one launch / its event-timed duration / 8 TB/s) PLUS what the
                  counters say the kernel is bound by: vector-instruction issue fraction, mean occupancy, measured HBM bytes
                  (PMC; from profiles/counters_latest.json when it was taken on this very kernel source with these settings),
                  the HBM floor of a frame, and the lone frame's latency model (longest dependent chain x time per position)
  cpu_baseline    the scalar oracle on this host's cores (a reported baseline)
  latency         one frame per launch, one launch in flight: ms per frame as an application waiting for each frame sees it
  scale_anchor    the ONE-GPU rate with the settings an N-GPU run uses (RGBA8 tiles, 2 streams x 48 frames per launch): the
                  like-for-like base of the scaling curve; N > 1 lines carry speedup_vs_anchor (anchor re-measured on rank 0)
  end_to_end      vrt_render_begin/_end: march + copy of the frame to pinned host memory, pipelined
  config4         the same scene at 3840x2160 split the same N ways (BASELINE config 4), with its own anchor
  reference_texel_format   the same frames with the volume kept as the reference's 16-bit texel (opt-in device format)
  full_closest_hit         the same frames with one point light in the scene: the full closest hit, in passes and as one kernel
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E (spec)
# tools/microbench/gather.hip / gather16.hip on MI355X (re-measured on the round-4 build: profiles/r04_gather_microbench.txt): trilinear samples/s the chip sustains for
# 4 x dwordx2 taps per lane at random cells, by where the bricks live; l1_coherent_lanes: the same with the 64 lanes of a wave on the 3x3 cells
# an 8x8-pixel tile covers (lanes that share a cell share its lines; 552 / 561 / 540 from L1 / L2 / the 256^3 pool)
NATIVE_PROBE_DEADLINE_S = float(os.environ.get("VRT_BENCH_NATIVE_PROBE_DEADLINE_S", "120"))  # vrt_comm_init + a 64-byte exchange and gather; RCCL's bootstrap takes a few seconds
# FALLBACK only (a library without vrt_debug_gather_ceiling): since round 5 the ceilings are measured in the run itself, on the run's own box
GATHER_CEILING_GSAMPLES = {"l1": 264.0, "l2": 209.0, "mall_hbm": 126.0, "l1_int16": 298.0, "l1_coherent_lanes": 552.0}


def measure_gather_ceilings(r, _abi):
    """The march's inner operation in isolation on THIS box and build (vrt_debug_gather_ceiling, ~10 ms each): G trilinear samples per second
    for independent cells from an L1- / L2-resident pool and from the pool of a 256^3 volume (fp32 bricks), for int16 bricks, and with the 64
    lanes of a wave on the 3x3 cells of an 8x8-pixel tile.  Returns (table, True) or (the round-4 constants, False)."""
    try:
        return {"l1": round(r.gather_ceiling(_abi.FORMAT_F32, False, 32), 1),
                "l2": round(r.gather_ceiling(_abi.FORMAT_F32, False, 4096), 1),
                "mall_hbm": round(r.gather_ceiling(_abi.FORMAT_F32, False, 262144), 1),
                "l1_int16": round(r.gather_ceiling(_abi.FORMAT_TEXEL16, False, 32), 1),
                "l1_coherent_lanes": round(r.gather_ceiling(_abi.FORMAT_F32, True, 32), 1),
                "l1_int16_coherent_lanes": round(r.gather_ceiling(_abi.FORMAT_TEXEL16, True, 32), 1)}, True
    except Exception as e:  # noqa: BLE001
        print(f"[bench] vrt_debug_gather_ceiling unavailable ({e!r}): round-4 constants", file=sys.stderr)
        return dict(GATHER_CEILING_GSAMPLES), False


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c3sdf", "c3light", "c3cover", "c3dropin", "c4", "c5"],
                    help="c3 = BASELINE config 3 (the metric); c3light = the same with one point light: the full closest-hit kernel "
                         "(the reference's default render mode once a scene has a point light)")
    ap.add_argument("--path", default="auto", choices=["auto", "dense", "brick", "lds", "cells"])
    ap.add_argument("--format", default="auto", choices=["auto", "f32", "texel16"],
                    help="device volume format: f32 bricks, or the reference's 16-bit texel (sign + 15-bit |d|*100)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N>1: strong = the workload's own frame split N ways (the metric); weak = frame grows with N")
    ap.add_argument("--output", default="auto", choices=["auto", "f32", "rgba8"],
                    help="tile pixel format; auto = float4 on one GPU, RGBA8 (the exchange format) on several")
    ap.add_argument("--strip-rows", type=int, default=8,
                    help="N>1: rows per interleaved strip (8: one wave row; 1080 rows deal out with 0.7 %% padding at N = 8, 15 %% with 32); 0 = contiguous row tiles")
    ap.add_argument("--frames-in-flight", type=int, default=0, choices=[0, 1, 2, 3, 4, 5, 6, 8],
                    help="HIP streams that carry blocks of frames (each with its own tile buffer); 0 = 1 on one GPU (one march launch "
                         "covers a whole block of frames: the launch itself keeps the frames in flight), 2 on several (a block's gather "
                         "overlaps the next block's march)")
    ap.add_argument("--frames-per-step", type=int, default=96,
                    help="frames in the batch ONE step renders: consecutive views of a camera on a short orbit (0.25 degrees apart)")
    ap.add_argument("--launches-per-step", type=int, default=0,
                    help="passes over the batch that make ONE step (each pass = --frames-per-step frames); 0 = default: 30 on the main "
                         "line (a step is then 2 880 frames, about 0.11 s on one MI355X: the driver's --steps 20 --warmup 5 times 2 s after "
                         "0.5 s of warm-up, long enough for clocks to settle and for SMI samples to see the GPU busy), 1 with --steps <= 2")
    ap.add_argument("--block-frames", type=int, default=0,
                    help="frames per vrt_render_block call = per march launch (up to 48; one event pair per launch); 0 = default")
    ap.add_argument("--per-frame-launches", action="store_true",
                    help="A/B: vrt_render_block issues one march launch per frame (VRT_FLAG_BLOCK_PER_FRAME) instead of one per block")
    ap.add_argument("--k-relax", type=float, default=0.0,
                    help="over-relaxation factor of the sphere trace (vrt_params.k_relax); 0 = the renderer's default (1.7), 1 = plain")
    ap.add_argument("--no-hit-polish", action="store_true",
                    help="A/B: VRT_FLAG_NO_HIT_POLISH — closest hits stay where the cone threshold stopped the ray (rounds 1-3)")
    ap.add_argument("--tile-map", default="supertile", choices=["supertile", "band", "linear"], help="blockIdx -> tile map (speed only)")
    ap.add_argument("--exchange", default="rotate", choices=["rotate", "gather"],
                    help="N>1: where the frames of a block are assembled: rotate = frame g on rank g // (block / N), one all-to-all "
                         "per block (every xGMI link carries tiles); gather = every frame on rank 0 (ncclGather; rank 0's inbound "
                         "links bound the job).  The other one is measured as a leg of the same run")
    ap.add_argument("--gather", default="native", choices=["torch", "native"],
                    help="N>1: the collective's implementation: the C-ABI's own vrt_gather_tiles / vrt_exchange_tiles (the PRODUCT's path: "
                         "ncclGather / grouped ncclSend+ncclRecv on the march stream, RCCL resolved by dlopen; default — falls back to "
                         "torch.distributed, and says so in `collective`, when librccl cannot be resolved or vrt_comm_init fails), or "
                         "torch.distributed (RCCL under torch).  Whichever is not the main line is cross-checked after the timed region")
    ap.add_argument("--native-check", action="store_true",
                    help="N>1 with --gather torch: also bring up the C-ABI's own RCCL communicator and run a few frames through "
                         "vrt_gather_tiles after the timed region (native_gather_check)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the latency / end_to_end / config4 legs")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target CPU time of the baseline sample")
    ap.add_argument("--launch-check", action="store_true",
                    help="CPU-only check of the N-rank launch path: ranks rendezvous over gloo, gather synthetic tiles "
                         "through FrameGather and report; no march, no GPU (never a measurement)")
    return ap.parse_args(argv)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n: int, argv) -> int:
    """Start the n ranks as a child `python -m torch.distributed.run` and return its exit code.  Called before
    this process has made any GPU call (a process that initialised the GPU must never be replaced or forked)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    env.setdefault("GPU_MAX_HW_QUEUES", "8")  # 8 frames in flight per rank want 8 hardware queues (default 4)
    return subprocess.run(cmd, env=env).returncode


def build_workload(name: str):
    """Returns (scene, width, height, max_steps, shadow, label)."""
    from volumetricraytracer_amd import workloads

    if name == "c3":
        sc = workloads.bench_config3()
        return sc, 1920, 1080, 255, True, "config3: 256^3 voxelized glTF-style mesh (torus, 16384 triangles), 1920x1080, shadow ray on"
    if name == "c3light":
        import copy

        import volumetricraytracer_amd as v
        sc = copy.copy(workloads.bench_config3())
        sc.PointLights = [v.VPointLight(Position=(120.0, 60.0, 140.0), Color=(1.0, 0.9, 0.8, 1.0), IlluminationStrength=40.0)]
        return sc, 1920, 1080, 255, True, "config3 + one point light: 256^3 voxelized mesh, 1920x1080, directional + point light with their shadow rays (full closest-hit kernel)"
    if name == "c3dropin":
        # config 3 as a drop-in user of the C++ adaptor gets it: the reference's 1x1 default normal texel on the material (mode Interp, the
        # reference's 16-bit volume texel, B8G8R8A8 frames and the two reference-artefact flags are set in main()) — the `drop_in_defaults`
        # leg as a workload of its own, for the profiler
        import copy

        base = workloads.bench_config3()
        sc = copy.copy(base)
        vol = copy.copy(base.volumes()[0])
        vol.Material = copy.copy(vol.Material)
        vol.Material.NormalTexture = workloads.reference_default_normal_texel()
        sc.Objects = [copy.copy(o) for o in base.Objects]
        sc.Objects[0].Volume = vol
        return sc, 1920, 1080, 255, True, ("config3 with the C++ adaptor's defaults: mode Interp, reference 16-bit volume texel, the reference's 1x1 default normal texel, "
                                          "B8G8R8A8 frames, VRT_FLAG_REFERENCE_VIEW_VECTOR | _BOUNDARY_TEXELS (the lean kernel's REF instantiation)")
    if name == "c3cover":
        # config 3's volume with the camera INSIDE its box, 0.6 extents from the centre: every wave marches (the `full_coverage` leg as a workload
        # of its own, for the profiler)
        base = workloads.bench_config3()
        sc = workloads.config3_voxelized(8, 256, distance=0.6 * float(base.volumes()[0].VolumeExtends))
        sc.Objects[0].Volume = base.volumes()[0]
        return sc, 1920, 1080, 255, True, "config3's 256^3 voxelized mesh, camera inside the volume's box (every wave marches), 1920x1080, shadow ray on"
    if name == "c3sdf":
        sc = workloads.config3_torus(8, 256, distance=190.0)
        return sc, 1920, 1080, 255, True, "config3 (analytic SDF variant): 256^3 torus SDF, 1920x1080, shadow ray on"
    if name == "c2":
        sc = workloads.config2_sphere(6, 256)
        return sc, 1280, 720, 128, False, "config2: 64^3 SDF sphere, 1280x720, 128 max steps"
    if name == "c4":
        sc = workloads.bench_config3()
        return sc, 3840, 2160, 255, True, "config4: 256^3 voxelized mesh, 3840x2160, row tiles + RCCL gather"
    if name == "c5":
        sc = workloads.config5_instances(7, 256)
        return sc, 1920, 1080, 255, True, "config5: 8 instanced 128^3 volumes + skybox, 1920x1080, AABB BVH"
    raise SystemExit(f"unknown workload {name}")


def kernel_source_hash() -> str:
    """sha256 of the kernel + device-struct sources: ties a PMC traffic figure to the code it was measured on."""
    h = hashlib.sha256()
    for rel in ("volumetricraytracer_amd/csrc/vrt_kernels.hip", "volumetricraytracer_amd/csrc/vrt_device.h"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def traffic_key(args, world: int, K: int, G: int, rgba8: bool) -> dict:
    return {"workload": args.workload, "n_gpus": world, "path": args.path, "format": args.format, "tile_map": args.tile_map,
            "streams": K, "frames_per_launch": G, "rgba8": bool(rgba8), "k_relax": args.k_relax, "frames_per_step": args.frames_per_step,
            "hit_polish": not args.no_hit_polish,
            "kernel_source_sha": kernel_source_hash()}


def measured_counters(key: dict):
    """Per-launch PMC figures of the march kernel (HBM bytes, vector instructions, wave quad-cycles, GPU cycles) from the
    passes of tools/r03_profile.sh, or None when the committed figures were not taken on this kernel source with these
    settings."""
    tpath = os.path.join(ROOT, "profiles", "counters_latest.json")
    try:
        tj = json.load(open(tpath))
    except Exception:
        return None
    if all(tj.get("key", {}).get(k) == v for k, v in key.items()):
        return tj
    return None


class _StreamEvent:
    """The part of torch.distributed's Work object the pipeline uses, for work enqueued on a HIP stream: wait() makes the
    CURRENT stream wait for everything that was in `stream` when this object was made."""

    def __init__(self, torch, stream):
        self._torch = torch
        self._ev = torch.cuda.Event()
        self._ev.record(stream)

    def wait(self):
        self._torch.cuda.current_stream().wait_event(self._ev)


def block_plan(frames: int, G: int, K: int):
    """Sizes of the vrt_render_block calls that issue exactly `frames` frames over K streams: whole rounds of K blocks of G
    frames, then the rest dealt as evenly as possible over the K streams (no stream ends with a long queue while the others
    idle).  Block i goes to stream i % K."""
    rounds, rest = divmod(max(frames, 0), G * K)
    plan = [G] * (rounds * K) + [rest // K + (1 if k < rest % K else 0) for k in range(K)]
    return [n for n in plan if n > 0]


class Pipeline:
    """K streams, each marching one frame at a time: K frames in flight.  Frames are issued in BLOCKS of up to G frames per
    call on one stream (vrt_render_block: G launches back to back, one event pair per block instead of one per frame), blocks
    round-robin over the streams.  N > 1: one gather per block (G tiles per rank in one collective, enqueued behind the
    block's marches and overlapping the other streams' marches); rank 0 un-shuffles a gathered block of strips into frame
    order with one strided copy on a stream of its own.  No cross-stream dependency on the march path: on this runtime
    an event wait between streams costs several microseconds of queue time (profiles/r02_launch_overhead.txt)."""

    def __init__(self, r, p, W, H, world, rank, dev, rgba8, strip_rows, K, rehearsal, native=False, block_frames=8, cameras=None,
                 n_cameras=0, rotate=False):
        import torch

        from volumetricraytracer_amd.tiles import FrameGather

        self.torch, self.r, self.p, self.world, self.rank, self.K = torch, r, p, world, rank, K
        self.G = G = max(int(block_frames), 1)
        self.strip_rows, self.rehearsal, self.native = strip_rows, rehearsal, native and not rehearsal and world > 1
        pix = torch.uint8 if rgba8 else torch.float32
        self.fg = FrameGather(H, W, world, rank, torch.device("cpu") if rehearsal else dev, dtype=pix, buffers=K, strip_rows=strip_rows,
                              frames_per_gather=G, rotate_roots=rotate and world > 1)
        self.march_tiles = [torch.zeros_like(x, device=dev) for x in self.fg.tiles] if rehearsal else self.fg.tiles
        self.frame_bytes = self.fg.rows_per * W * (4 if rgba8 else 16)
        # (the current stream + K-1 pool streams: with HIP's default 4 hardware queues this arrangement lands on distinct queues;
        # K pool streams measured 40 % slower at K = 3, profiles/r02_strong_scaling_probe.txt)
        self.streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(K - 1)]
        self.pending = [None] * K
        self.unshuffle = world > 1 and strip_rows > 0 and (rank == 0 or self.fg.rotate)
        self.copy_stream = torch.cuda.Stream(device=dev) if self.unshuffle else None
        self.unshuffled = [None] * K
        self.blocks = 0
        if self.native:
            # the byte counts this pipeline's collectives are going to carry, agreed on by all ranks up front (a rank with another block size
            # would otherwise hang the first collective inside RCCL)
            tb = self.fg.tiles[0].numel() * self.fg.tiles[0].element_size()
            r.comm_expect_sizes(0 if self.fg.rotate else tb, tb // world if self.fg.rotate else 0)
        self.xchg_events = []  # native exchange only: (before, after) timing events around every block's collective on its march stream
        self.last = (0, 0)  # (buffer, frame within the block) of the last frame issued
        # the batch's cameras, twice over (a block may wrap around the end of the batch): frame i uses camera i % n_cameras
        self.cameras, self.n_cameras, self.frame_no = cameras, n_cameras, 0

    def run(self, steps: int) -> None:
        """Issue exactly `steps` FRAMES: blocks of G on stream 0, 1, ... K-1, 0, ...; the last round is dealt evenly."""
        torch, fg, r = self.torch, self.fg, self.r
        # whole rounds of K blocks of G frames; the last (partial) round is dealt evenly over the K streams, so that no stream
        # ends with a long queue while the others idle
        for n in block_plan(steps, self.G, self.K):
            b = self.blocks % self.K
            st = self.streams[b]
            with torch.cuda.stream(st):
                if self.pending[b] is not None:
                    self.pending[b].wait()  # this buffer's previous block has been gathered: stream b waits, not the host
                    self.pending[b] = None
                if self.unshuffled[b] is not None:
                    st.wait_event(self.unshuffled[b])
                cams = (self.cameras, self.frame_no % self.n_cameras) if self.cameras is not None else None
                if self.strip_rows > 0:
                    r.render_block(self.p, n, self.march_tiles[b].data_ptr(), self.frame_bytes, st.cuda_stream,
                                   strips=(self.strip_rows, self.rank, self.world, fg.strips_per), cameras=cams)
                else:
                    r.render_block(self.p, n, self.march_tiles[b].data_ptr(), self.frame_bytes, st.cuda_stream, rows=(fg.row0, fg.rows),
                                   cameras=cams)
                self.frame_no += n
                if self.rehearsal:
                    fg.tiles[b].copy_(self.march_tiles[b])
                if self.world > 1:
                    if self.native:  # ncclGather right behind the block's marches on the same stream; "pending" = an event after it
                        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        ea.record(st)
                        fg.native_gather(r, b, st.cuda_stream)
                        eb.record(st)
                        self.xchg_events.append((ea, eb))
                        if len(self.xchg_events) > 4096:
                            del self.xchg_events[:2048]
                        self.pending[b] = _StreamEvent(torch, st)
                    else:
                        self.pending[b] = fg.gather(b, async_op=True)  # RCCL gather over xGMI, overlaps the other streams' marches
            if self.unshuffle:
                with torch.cuda.stream(self.copy_stream):
                    self.pending[b].wait()  # the copy stream (not the host) waits for this gather (gloo rehearsal: the host does)
                    fg.unshuffle(b)         # gathered [rank, frame, strip] order -> frame order, one strided device copy
                    if self.unshuffled[b] is None:
                        self.unshuffled[b] = torch.cuda.Event()
                    self.unshuffled[b].record(self.copy_stream)
            self.last = (b, n - 1)
            self.blocks += 1

    def exchange_ms(self, last_n: int):
        """Mean duration (ms) of the last `last_n` blocks' native collectives on this rank (call after drain()); None without them."""
        ev = self.xchg_events[-max(last_n, 1):]
        if not ev:
            return None
        return float(sum(a.elapsed_time(b) for a, b in ev) / len(ev))

    def last_frame(self):
        """The assembled last frame issued, on the rank that assembles it (rank 0, or the frame's root with rotating roots);
        None elsewhere."""
        return self.fg.frame(*self.last)

    def drain(self) -> None:
        torch = self.torch
        for b in range(self.K):
            with torch.cuda.stream(self.streams[b]):
                if self.pending[b] is not None:
                    self.pending[b].wait()
                    self.pending[b] = None
        torch.cuda.synchronize()


def timed_run(pipe: Pipeline, steps: int, warmup: int, world: int, cdev, frames_per_step: int = 1) -> float:
    """W untimed steps, then exactly `steps` steps (of frames_per_step frames each) bracketed by barrier + synchronize; MAX over
    ranks (seconds).  Every timed run starts at the first camera of the batch."""
    import torch
    import torch.distributed as dist

    def barrier() -> None:
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    pipe.run(warmup * frames_per_step)
    pipe.drain()
    barrier()
    t0 = time.perf_counter()
    pipe.run(steps * frames_per_step)
    pipe.drain()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    return elapsed


def multi_gpu_summary(value, anchor_v, rotate, exchange_other, world, W, H, bytes_per_pixel, rays_per_frame, other_steps):
    """The N > 1 line's top-level account of the exchange: both exchanges side by side with their speed-up over
    the one-GPU anchor, and what the link model says about them.  north_star asks for ">= 6x at 8 GPUs" with "an RCCL gather ... for
    the final image": a gather onto ONE rank moves (N-1)/N of every frame over that rank's N-1 inbound xGMI links (~77 GB/s per
    direction each, ~80 % reached by RCCL), which bounds the whole job whatever the march does; rotating roots (one all-to-all per
    block) spread the same bytes over every link of the node."""
    main_name, other_name = ("rotate", "gather") if rotate else ("gather", "rotate")
    ex = {main_name: {"value": round(value, 2), "speedup_vs_anchor": round(value / anchor_v, 3) if anchor_v else None, "main_line": True}}
    if isinstance(exchange_other, dict) and exchange_other.get("value"):
        ex[other_name] = {"value": exchange_other["value"], "speedup_vs_anchor": round(exchange_other["value"] / anchor_v, 3) if anchor_v else None,
                          "main_line": False, "steps": other_steps}
    frame_bytes = W * H * bytes_per_pixel
    link_gbs, eff = 77.0, 0.8
    cap_fps = world * (link_gbs * 1e9 * eff) / frame_bytes  # (N-1)/N of a frame over N-1 links
    return {
        "exchanges": ex,
        "link_model": {
            "gather_to_rank0_cap_Mrays_per_s": round(cap_fps * rays_per_frame / 1e6, 1),
            "gather_to_rank0_cap_speedup_vs_anchor": round(cap_fps * rays_per_frame / 1e6 / anchor_v, 2) if anchor_v else None,
            "assumes": f"{frame_bytes} B per assembled frame, {world - 1} inbound xGMI links x {link_gbs} GB/s x {eff} into the root",
            "rotating_roots": "the same bytes over all N(N-1) directed links: not link-bound at these rates"},
        "north_star_6x_at_8_gpus": ({k: (x["speedup_vs_anchor"] is not None and x["speedup_vs_anchor"] >= 6.0) for k, x in ex.items()}
                                    if world == 8 else None),
    }


def launch_check(args) -> None:
    """The N-rank plumbing without a march (CPU, gloo): rendezvous, strip layout, gather, un-shuffle, max-over-ranks."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from volumetricraytracer_amd.tiles import FrameGather, strip_frame_rows

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    H, W, sr = 77, 16, 8
    t0 = time.perf_counter()
    ok = True
    for rot in (False, True):  # gather to rank 0, and rotating roots (one all-to-all per block)
        G = 2 * world if rot else 3  # a block of G frames per collective, like the GPU pipeline
        fg = FrameGather(H, W, world, rank, torch.device("cpu"), dtype=torch.uint8, buffers=1, strip_rows=sr if world > 1 else 0,
                         frames_per_gather=G, rotate_roots=rot)

        def tag(rows0, rows, g):  # a pixel value that names its frame row and its frame of the block
            return ((torch.arange(rows0, rows0 + rows, dtype=torch.int32) + 7 * g) % 251).to(torch.uint8)[:, None, None]

        for _ in range(max(args.steps, 1)):
            for g in range(G):
                if world == 1:
                    fg.tile(0, g)[:H] = tag(0, H, g)
                    continue
                for local0, frame0, rows in strip_frame_rows(H, world, rank, sr):
                    fg.tile(0, g)[local0:local0 + rows] = tag(frame0, rows, g)
            if world > 1:
                fg.gather(0, async_op=True).wait()
                fg.unshuffle(0)
        mine = [g for g in range(G) if fg.root_of(g) == rank]
        good = all(bool(np.array_equal(fg.frame(0, g)[:, 0, 0].numpy(), ((np.arange(H) + 7 * g) % 251).astype(np.uint8))) for g in mine)
        flag = torch.tensor([1 if good else 0], dtype=torch.int32)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = ok and bool(flag.item())
    te = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
    if rank == 0:
        # (the N > 1 line's exchange account, assembled from synthetic figures: its keys are what a SCALE record must carry)
        summary = multi_gpu_summary(400000.0, 62000.0, True, {"value": 140000.0}, max(world, 2), 1920, 1080, 4, 2.4e6, 3)
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_joined": world, "gathered_frame_ok": ok,
                          "elapsed_s": round(float(te.item()), 4), "multi_gpu_line_keys": sorted(summary),
                          "multi_gpu_line_sample": summary}), flush=True)
        if not ok:
            raise SystemExit(3)
    if world > 1:
        dist.destroy_process_group()


def main() -> None:
    args = parse_args()
    n = max(args.gpus, 1)
    if "WORLD_SIZE" not in os.environ and n > 1:
        # not under a launcher: start the ranks ourselves, before anything here touches the GPU
        raise SystemExit(self_launch(n, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != n:
        print(f"[bench] rank {rank}: WORLD_SIZE={world} but --gpus={n}: refusing to measure a different job", file=sys.stderr)
        raise SystemExit(2)
    if args.launch_check:
        launch_check(args)
        return
    if world > 1:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # read by the HIP runtime at start-up: before torch is imported

    import numpy as np
    import torch
    import torch.distributed as dist

    import volumetricraytracer_amd as v
    from volumetricraytracer_amd import _abi, workloads

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # VRT_BENCH_BACKEND=gloo: rehearsal of the N>1 code path on a ONE-GPU box (every rank marches on cuda:0,
    # tiles are staged through host memory and gathered over gloo).  Never a measurement.
    rehearsal = os.environ.get("VRT_BENCH_BACKEND", "nccl") == "gloo"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if rehearsal else dev
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        joined = torch.ones(1, dtype=torch.int32, device=cdev)
        dist.all_reduce(joined)
        if int(joined.item()) != n:
            raise SystemExit(f"[bench] only {int(joined.item())} of {n} ranks joined")

    sc, W, H, max_steps, shadow, label = build_workload(args.workload)
    if world > 1 and args.scaling == "weak":
        W, H = int(round(W * world ** 0.5)), int(round(H * world ** 0.5))
        label += f" -- weak-scaled to {W}x{H} for {world} GPUs"
    elif world > 1:
        label += f" -- the same {W}x{H} frame split over {world} GPUs (strong scaling)"
    dropin = args.workload == "c3dropin"
    rgba8 = args.output == "rgba8" or (args.output == "auto" and world > 1) or dropin
    strip_rows = args.strip_rows if world > 1 else 0
    rotate = world > 1 and args.exchange == "rotate"
    exchange_fallback = None
    if rotate:
        # the all-to-all is what the default exchange stands on: try it once on a few bytes, and fall back to the gather (said in the
        # line) rather than lose the whole run if this backend cannot do it — every rank fails or passes the same way
        try:
            probe_in = torch.arange(world * 4, dtype=torch.uint8, device=cdev)
            probe_out = torch.zeros_like(probe_in)
            dist.all_to_all_single(probe_out, probe_in)
            if not rehearsal:
                torch.cuda.synchronize()
            # chunk s of the output is chunk `rank` of rank s's input: bytes 4*rank .. 4*rank+3 from every rank
            ok = bool((probe_out.view(world, 4).cpu() == (torch.arange(4, dtype=torch.uint8) + 4 * rank)[None, :]).all())
        except Exception as e:  # noqa: BLE001
            ok, exchange_fallback = False, f"all_to_all_single unavailable ({e!r}): gather to rank 0 instead"
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if not bool(flag.item()):
            rotate = False
            exchange_fallback = exchange_fallback or "all_to_all_single failed on another rank: gather to rank 0 instead"
    # One GPU: ONE stream — a march launch covers a whole block of frames, the launch itself keeps the frames in flight (the
    # reference keeps 3 on its swap chain, DXConstants.cpp:23).  Several: 2 streams, so that a block's collective overlaps the
    # next block's march.  (No dependence on how HIP maps streams to hardware queues any more: profiles/r03_fused_launch_sweep.txt.)
    K = args.frames_in_flight or (1 if world == 1 else MULTI_STREAMS)
    path = {"auto": _abi.PATH_AUTO, "dense": _abi.PATH_DENSE, "brick": _abi.PATH_BRICK, "lds": _abi.PATH_BRICK_LDS, "cells": _abi.PATH_CELLS}[args.path]
    fmt = {"auto": _abi.FORMAT_TEXEL16 if dropin else workloads.BENCH_VOLUME_FORMAT, "f32": _abi.FORMAT_F32, "texel16": _abi.FORMAT_TEXEL16}[args.format]
    for vol in sc.volumes():
        vol.set_device_format(fmt)

    def params(w, h, as_rgba8=None):
        q = v.default_params(w, h, workloads.min_cell(sc), max_steps, shadow=shadow, path=path)
        if args.k_relax > 0.0:
            q.k_relax = args.k_relax
        if rgba8 if as_rgba8 is None else as_rgba8:
            q.flags |= _abi.FLAG_OUTPUT_RGBA8
        if args.per_frame_launches:
            q.flags |= _abi.FLAG_BLOCK_PER_FRAME
        if args.no_hit_polish:
            q.flags |= _abi.FLAG_NO_HIT_POLISH
        q.flags |= {"supertile": 0, "band": 1, "linear": 2}[args.tile_map]
        if dropin:  # VHipRenderer's defaults (csrc/host/HipRenderer.h)
            q.mode, q.max_bounces = _abi.MODE_INTERP, 2
            q.flags |= _abi.FLAG_OUTPUT_BGRA8 | _abi.FLAG_REFERENCE_VIEW_VECTOR | _abi.FLAG_REFERENCE_BOUNDARY_TEXELS
        return q

    p = params(W, H)
    r = v.VHipRenderer(devices=(local_rank,))
    if not r.Start():
        raise SystemExit("VHipRenderer.Start() failed")
    r.SetSceneToRender(sc)
    r.ResizeRenderOutput(W, H)
    r.SyncWithScene()

    native_ready, native_error, probe_abandoned = False, None, False
    # VRT_BENCH_NATIVE_PROBE=1 runs this block in a rehearsal too: two ranks on ONE GPU is something RCCL refuses or waits on for
    # ever, which is exactly the failure the deadline below exists for (tests/test_parity_gpu.py walks it)
    if world > 1 and (not rehearsal or os.environ.get("VRT_BENCH_NATIVE_PROBE") == "1"):
        # the C-ABI's own communicator — the product's exchange path, and the main line's by default (an N > 1 run
        # must measure vrt_comm_init / vrt_exchange_tiles / vrt_gather_tiles, not torch.distributed): rank 0 makes the id,
        # torch.distributed carries it to the others
        # ... inside a thread with a deadline: a bootstrap that never returns (it has never run on N > 1 GPUs) must cost the run its
        # native path, not the run — ctypes drops the GIL in the call, every rank times out alike, and the flag below agrees on it
        # ... and on a CONTEXT OF ITS OWN: a probe that is abandoned at its deadline leaves its thread inside vrt_comm_init or
        # the probe exchange for good; that thread must not share a vrt_ctx with the measurement (vrt_destroy under it = use after free).
        # The probe context is never stopped once abandoned; the main context only initialises its communicator after the probe has
        # come back.  If the probe's GPU collective itself is stuck, no device-wide synchronize would ever return: the run then prints
        # what it knows and leaves with os._exit (no destructors over a thread that is still inside RCCL).
        def fresh_id():
            idt = torch.zeros(_abi.VRT_COMM_ID_BYTES, dtype=torch.uint8, device=cdev)
            err = None
            try:
                if rank == 0:
                    idt.copy_(torch.frombuffer(bytearray(v.VHipRenderer.comm_unique_id()), dtype=torch.uint8))
            except Exception as e:  # noqa: BLE001
                err = repr(e)
            dist.broadcast(idt, 0)
            return bytes(idt.cpu().numpy().tobytes()), err

        id_bytes, native_error = fresh_id()
        probe_stream = torch.cuda.Stream(device=dev)
        probe_result = {}
        probe_ctx = v.VHipRenderer(devices=(local_rank,))

        def native_probe():
            try:
                torch.cuda.set_device(dev)
                if not probe_ctx.Start():
                    raise RuntimeError("probe context: Start() failed")
                probe_ctx.comm_init(world, rank, id_bytes)
                # the product's exchange on a few bytes before the run stands on it: chunk d of my buffer goes to rank d
                # (vrt_exchange_tiles), every rank's tile lands rank-major on rank 0 (vrt_gather_tiles)
                cb = 64
                with torch.cuda.stream(probe_stream):
                    snd = (torch.arange(world, dtype=torch.int32, device=dev)[:, None] + 16 * rank).to(torch.uint8).expand(world, cb).contiguous()
                    rcv = torch.zeros_like(snd)
                    tile = torch.full((cb,), 100 + rank, dtype=torch.uint8, device=dev)
                    got = torch.zeros((world, cb), dtype=torch.uint8, device=dev)
                    st0 = probe_stream.cuda_stream
                    probe_ctx.exchange_tiles(snd.data_ptr(), rcv.data_ptr(), cb, st0)
                    probe_ctx.gather_tiles(tile.data_ptr(), got.data_ptr() if rank == 0 else 0, cb, 0, st0)
                    probe_stream.synchronize()
                    want = (16 * torch.arange(world, dtype=torch.int32)[:, None] + rank).to(torch.uint8).expand(world, cb)
                    ok_x = bool((rcv.cpu() == want).all())
                    ok_g = rank != 0 or bool((got.cpu() == (100 + torch.arange(world, dtype=torch.int32))[:, None].to(torch.uint8)).all())
                if not (ok_x and ok_g):
                    raise RuntimeError(f"native exchange probe delivered wrong bytes (exchange ok {ok_x}, gather ok {ok_g})")
                probe_result["ok"] = True
            except Exception as e:  # reported, never fatal: torch.distributed remains
                probe_result["error"] = repr(e)

        if native_error is None:
            import threading
            th = threading.Thread(target=native_probe, daemon=True)
            th.start()
            th.join(NATIVE_PROBE_DEADLINE_S)
            if th.is_alive():
                probe_abandoned = True
                native_error = f"vrt_comm_init / the probe exchange did not return within {NATIVE_PROBE_DEADLINE_S} s"
                if not probe_stream.query():
                    # the probe's collective sits on the device and does not finish: nothing that synchronises the device can return
                    print(json.dumps({"error": "native exchange probe hung ON THE DEVICE: " + native_error, "n_gpus": world, "rank": rank}), flush=True)
                    os._exit(3)
            else:
                native_ready, native_error = bool(probe_result.get("ok")), probe_result.get("error")
                probe_ctx.Stop()  # the probe came back: its context (and communicator) can go
        flag = torch.tensor([1 if native_ready else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        native_ready = bool(flag.item()) and not rehearsal
        if native_ready:
            # the measurement's own communicator, on the context the measurement uses: the same call sequence just worked on this box
            try:
                id2, err2 = fresh_id()
                if err2:
                    raise RuntimeError(err2)
                r.comm_init(world, rank, id2)
            except Exception as e:  # noqa: BLE001
                native_ready, native_error = False, repr(e)
            flag = torch.tensor([1 if native_ready else 0], dtype=torch.int32, device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            native_ready = bool(flag.item())
    use_native = args.gather == "native" and native_ready
    native_fallback = None
    if args.gather == "native" and world > 1 and not native_ready and (not rehearsal or native_error):
        # never lose the run: torch.distributed carries the exchange instead, and the line says so
        native_fallback = f"vrt_comm_init unavailable ({native_error}): torch.distributed carries the exchange"
        print(f"[bench] {native_fallback}", file=sys.stderr)

    # One STEP = one batch of B frames: consecutive views of a camera on a short orbit through the workload's own view
    # (frames of a moving camera, not B copies of one frame, so a frame does not find its predecessor's lines in L2).
    B = max(args.frames_per_step, 1)
    # the main line's step: L passes over the batch (L launches of B frames on one GPU)
    L = args.launches_per_step if args.launches_per_step > 0 else (30 if args.steps > 2 else 1)
    # frames per vrt_render_block call = per march launch: one GPU the whole step (96: a launch's latency-bound tail is paid once per
    # launch; beyond 48 frames the cameras are copied to the device ahead of the launch instead of travelling in the kernarg segment);
    # several GPUs: about 48 (a multiple of N), exchanged as ONE block per collective
    G = min(args.block_frames or (_abi.MAX_LAUNCH_FRAMES if world == 1 else multi_block_frames(world)), B)
    if rotate and G % world != 0:
        raise SystemExit(f"[bench] --exchange rotate needs --block-frames to be a multiple of the {world} ranks")
    cams = workloads.orbit_cameras(sc, B)
    cam_arr = r.camera_array(cams + cams)
    KEYS = ("primary_rays", "shadow_rays", "bounce_rays", "primary_steps", "shadow_steps", "hits", "exhausted_rays")

    def batch_counts(pp, w, h, alone=False):
        """The counters of one batch (every camera once, this rank's rows — the whole frame with alone=True —, untimed), summed
        over the ranks."""
        from volumetricraytracer_amd.tiles import strip_layout, tile_rows
        tot = {k: 0.0 for k in KEYS}
        is8 = bool(pp.flags & _abi.FLAG_OUTPUT_RGBA8)
        if alone:
            rows_here, kw = h, {"rows": (0, h)}
        elif strip_rows > 0:
            per = strip_layout(h, world, strip_rows)[1]
            rows_here, kw = per * strip_rows, {"strips": (strip_rows, rank, world, per)}
        else:
            _, row0, rows_here = tile_rows(h, world, rank)
            kw = {"rows": (row0, rows_here)}
        scratch = torch.empty((max(rows_here, 1), w, 4), dtype=torch.uint8 if is8 else torch.float32, device=dev)
        for f in range(B):
            r.render_block(pp, 1, scratch.data_ptr(), scratch.numel() * scratch.element_size(), 0, cameras=(cam_arr, f), **kw)
            torch.cuda.synchronize()
            tt = r.last_timing()
            for k in KEYS:
                tot[k] += tt[k]
        c = torch.tensor([tot[k] for k in KEYS], dtype=torch.float64, device=cdev)
        if world > 1 and not alone:
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
        return dict(zip(KEYS, (float(x) for x in c.tolist())))

    def pipeline(pp, w, h, k, native, g=None, rot=None):
        return Pipeline(r, pp, w, h, world, rank, dev, rgba8, strip_rows, k, rehearsal, native=native, block_frames=g or G, cameras=cam_arr,
                        n_cameras=B, rotate=rotate if rot is None else rot)

    ceilings, ceilings_measured = measure_gather_ceilings(r, _abi) if rank == 0 else (dict(GATHER_CEILING_GSAMPLES), False)
    pipe = pipeline(p, W, H, K, use_native)
    elapsed = timed_run(pipe, args.steps, args.warmup, world, cdev, B * L)
    xchg_ms = pipe.exchange_ms(n_timed_blocks := len(list(block_plan(args.steps * B * L, G, K)))) if world > 1 else None
    # the event-timed march launches of the timed region: (ms, frames the launch covered); only whole blocks count
    n_timed = len(list(block_plan(args.steps * B * L, G, K)))  # launches of the timed region (the warm-up's come before them)
    hist = [(ms, fr) for ms, fr in r.launch_history(min(max(n_timed, 1), 200)) if ms > 0.0]
    fpl = max((fr for _, fr in hist), default=1)  # frames per launch
    kms = [ms for ms, fr in hist if fr == fpl]

    def single_gpu_frame(pp, w, h, cam_index):
        whole = torch.empty((h, w, 4), dtype=torch.uint8 if pp.flags & _abi.FLAG_OUTPUT_RGBA8 else torch.float32, device=dev)
        r.render_block(pp, 1, whole.data_ptr(), whole.numel() * whole.element_size(), 0, cameras=(cam_arr, cam_index), rows=(0, h))
        torch.cuda.synchronize()
        return whole

    def all_ok(ok: bool) -> bool:
        if world == 1:
            return ok
        f = torch.tensor([1 if ok else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        return bool(f.item())

    def last_frame_ok(pl, pp, w, h) -> bool:
        """The assembled (exchanged and un-shuffled) last frame of a pipeline is the frame ONE GPU renders alone from that
        camera, bit for bit — checked on the rank that holds it, agreed on by all."""
        ok = True
        got = pl.last_frame()
        if got is not None:
            ok = bool(torch.equal(got.cpu(), single_gpu_frame(pp, w, h, (pl.frame_no - 1) % B).cpu()))
        return all_ok(ok)

    verified = None
    if args.steps > 0 and (world > 1 or os.environ.get("VRT_BENCH_VERIFY")):
        # always on for N > 1 (it is one frame, outside the timed region)
        verified = last_frame_ok(pipe, p, W, H)
        if not verified and use_native:
            # The product's own RCCL exchange has never run on N > 1 GPUs before this job: if the frame it assembled is wrong, say so
            # loudly in the line and measure the job with torch.distributed's collective instead of losing the whole scaling record
            native_fallback = "vrt_exchange_tiles / vrt_gather_tiles assembled a frame that DIFFERS from the single-GPU frame: main line re-run through torch.distributed"
            print(f"[bench] {native_fallback}", file=sys.stderr)
            use_native = False
            del pipe
            pipe = pipeline(p, W, H, K, False)
            elapsed = timed_run(pipe, args.steps, args.warmup, world, cdev, B * L)
            hist = [(ms, fr) for ms, fr in r.launch_history(min(max(n_timed, 1), 200)) if ms > 0.0]
            fpl = max((fr for _, fr in hist), default=1)
            kms = [ms for ms, fr in hist if fr == fpl]
            verified = last_frame_ok(pipe, p, W, H)
        if not verified:
            raise SystemExit("[bench] the assembled frame differs from the single-GPU frame")

    per_rank = None
    if world > 1:
        mine = torch.tensor([float(np.mean(kms)) if kms else 0.0, xchg_ms if xchg_ms is not None else -1.0], dtype=torch.float64, device=cdev)
        hi, lo = mine.clone(), mine.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        per_rank = {"march_ms_per_launch": {"rank0": round(float(mine[0]), 4), "max": round(float(hi[0]), 4), "min": round(float(lo[0]), 4)},
                    "exchange_ms_per_block": ({"rank0": round(float(mine[1]), 4), "max": round(float(hi[1]), 4), "min": round(float(lo[1]), 4)}
                                              if float(lo[1]) >= 0.0 else None),
                    "frames_per_launch": G, "streams": K,
                    "note": "event-timed on each rank's march stream over the timed region; exchange: the native collective only (vrt_exchange_tiles / "
                            "vrt_gather_tiles on the march stream); with two streams a block's collective overlaps the other stream's march"}
    cnt = batch_counts(p, W, H)  # whole job, one batch
    rays_per_batch = cnt["primary_rays"] + cnt["shadow_rays"]
    rays_per_step = rays_per_batch * L
    rays_per_frame = rays_per_batch / B
    ms_per_step = elapsed / max(args.steps, 1) * 1e3
    value = rays_per_step * args.steps / elapsed / 1e6 if args.steps > 0 else 0.0
    # mean frame of the batch as this rank's launch sees it (algorithmic bytes of ONE launch)
    t = {k: cnt[k] / B / world for k in KEYS}

    # ---- extra legs (outside the timed region) ------------------------------------------------------------------
    def leg(build, run):
        """An extra leg must never cost the main line: its allocations (which differ per rank — rank 0 holds the gathered
        frames) are agreed on by all ranks before anyone enters the leg's collectives; a failure is reported in the leg's key."""
        obj, err = None, None
        try:
            obj = build()
        except Exception as e:  # noqa: BLE001
            err = repr(e)
        if not all_ok(err is None):
            return {"error": err or "another rank could not set this leg up"}
        try:
            return run(obj)
        except Exception as e:  # noqa: BLE001
            return {"error": repr(e)}

    def anchor_leg(w, h):
        """The like-for-like base of the scaling curve: ONE GPU renders the whole w x h frame alone with the settings an N-GPU run
        uses — RGBA8 tiles, MULTI_STREAMS streams x 48 frames per launch — and no exchange.  N > 1: measured on rank 0 while the
        other ranks wait, in the same run on the same hardware as the N-GPU figure."""
        out = None
        if rank == 0:
            pa = params(w, h, as_rgba8=True)
            asteps = max(min(args.steps, 10), 1)  # (a few tens of milliseconds: a shorter run reads several per cent low while the clocks ramp)
            try:
                pl = Pipeline(r, pa, w, h, 1, 0, dev, True, 0, MULTI_STREAMS, False, block_frames=min(ANCHOR_BLOCK_FRAMES, B), cameras=cam_arr, n_cameras=B)
                ea = timed_run(pl, asteps, max(min(args.warmup, 3), 1), 1, cdev, B)
                ca = batch_counts(pa, w, h, alone=True)
                out = {"value": round((ca["primary_rays"] + ca["shadow_rays"]) * asteps / ea / 1e6, 2), "unit": "Mrays/s",
                       "ms_per_frame": round(ea / (asteps * B) * 1e3, 4), "n_gpus": 1,
                       "settings": f"whole {w}x{h} frame on one GPU, RGBA8 tiles, {MULTI_STREAMS} streams x {min(ANCHOR_BLOCK_FRAMES, B)} frames per launch, no exchange"}
                del pl
            except Exception as e:  # noqa: BLE001
                out = {"error": repr(e)}
        if world > 1:
            dist.barrier()
        return out

    native_check = exchange_other = None
    if world > 1 and not args.no_extra_legs and args.steps > 0:
        osteps = max(min(args.steps, 3), 1)

        def run_other(other):
            eo = timed_run(other, osteps, 1, world, cdev, B)
            return {"ms_per_frame": round(eo / (osteps * B) * 1e3, 4), "value": round(rays_per_batch * osteps / eo / 1e6, 2), "unit": "Mrays/s",
                    "last_frame_equals_single_gpu_frame": last_frame_ok(other, p, W, H)}

        if native_ready:  # the other implementation of the same collective, a few steps: same pixels, and its frame time
            native_check = leg(lambda: pipeline(p, W, H, K, not use_native), run_other)
            if isinstance(native_check, dict):
                native_check["collective"] = "torch.distributed" if use_native else "native (vrt_gather_tiles / vrt_exchange_tiles on the march stream)"
        G_other = G if (rotate or G % world == 0) else multi_block_frames(world)
        exchange_other = leg(lambda: pipeline(p, W, H, K, use_native, G_other, not rotate), run_other)
        if isinstance(exchange_other, dict):
            exchange_other["exchange"] = exchange_label(not rotate, world, G_other, rehearsal)
    latency = end_to_end = config4 = scale_anchor = None
    if not args.no_extra_legs and args.steps > 0:
        lsteps = max(min(args.steps, 4), 1)  # batches

        def run_latency(p1):
            # one frame per launch, one launch in flight: what an application that waits for every frame sees
            e1 = timed_run(p1, lsteps, 1, world, cdev, B)
            g1 = world if rotate else 1
            k1 = [x / g1 for x, fr in r.launch_history(200) if x > 0.0 and fr == g1]
            return {"streams": 1, "frames_per_launch": g1, "ms_per_frame": round(e1 / (lsteps * B) * 1e3, 4),
                    "value": round(rays_per_batch * lsteps / e1 / 1e6, 2), "unit": "Mrays/s",
                    "kernel_ms_per_frame": round(float(np.mean(k1)), 4) if k1 else None}

        latency = leg(lambda: pipeline(p, W, H, 1, use_native, world if rotate else 1), run_latency)
        if world == 1:
            def e2e(_):
                # the frame as the host gets it, in the bench's pixel format and in the reference's own back-buffer precision
                # (R8G8B8A8_UNORM, DXConstants.cpp:21): a quarter of the bytes over PCIe
                # (240 frames after 40 untimed ones: the first few dozen frames into freshly pinned host memory, or behind another leg's
                # freed buffers, read up to 40 % high — tools/e2e_leg_probe.py)
                end_to_end_leg(r, p, rays_per_frame, 40)
                out = end_to_end_leg(r, p, rays_per_frame, 240)
                p8 = _abi.vrt_params.from_buffer_copy(p)
                p8.flags |= _abi.FLAG_OUTPUT_RGBA8
                if not rgba8:
                    end_to_end_leg(r, p8, rays_per_frame, 40)
                out["rgba8"] = end_to_end_leg(r, p8, rays_per_frame, 240) if not rgba8 else None
                return out

            end_to_end = leg(lambda: None, e2e)
        # (the anchor runs behind the end-to-end leg: placed behind the anchor's two-stream pipeline, the frames-in-flight leg read 0.28 ms
        # per RGBA8 frame; at every earlier point of the process, and alone in tools/e2e_leg_probe.py, 0.167)
        if args.scaling == "strong":
            scale_anchor = anchor_leg(W, H)
        if args.workload in ("c3", "c4") and args.scaling == "strong":
            W4, H4 = 3840, 2160
            p4 = params(W4, H4)

            def build4():
                r.ResizeRenderOutput(W4, H4)
                return pipeline(p4, W4, H4, K, use_native)

            def run4(pipe4):
                s4 = max(min(args.steps, 4), 1)
                e4 = timed_run(pipe4, s4, 1, world, cdev, B)
                ok4 = last_frame_ok(pipe4, p4, W4, H4) if world > 1 else None
                c4 = batch_counts(p4, W4, H4)
                v4 = (c4["primary_rays"] + c4["shadow_rays"]) * s4 / e4 / 1e6
                a4 = anchor_leg(W4, H4)
                return {"workload": f"config4: 256^3 voxelized mesh, 3840x2160 split over {world} GPU(s)" +
                                    (f", {strip_rows}-row interleaved strips + " + exchange_label(rotate, world, G, rehearsal) if world > 1 else ""),
                        "ms_per_frame": round(e4 / (s4 * B) * 1e3, 4), "value": round(v4, 2), "unit": "Mrays/s",
                        "steps": s4, "frames_per_step": B, "streams": K, "frames_per_launch": G,
                        "assembled_frame_equals_single_gpu_frame": ok4, "scale_anchor": a4,
                        "speedup_vs_anchor": round(v4 / a4["value"], 3) if world > 1 and a4 and a4.get("value") else None}

            config4 = leg(build4, run4)
            r.ResizeRenderOutput(W, H)

    marching = no_cull = None
    if world == 1 and args.steps > 0:
        # How many of a frame's primary rays belong to waves that march at all: 4 out of 5 waves of the frame lie outside the
        # host's cull rectangle (or miss the volume's active box) and go straight to the sky.  From the per-wave records of the
        # batch's middle frame.  And the same frames with the rectangle switched off (every wave looks at the scene).
        single_gpu_frame(p, W, H, B // 2)
        rec = r.wave_records(0).astype(np.int64)
        busy = (rec[:, 3] + rec[:, 4]) > 0
        marching = {"primary_rays_in_marching_waves": int(rec[busy, 0].sum()), "primary_rays": int(rec[:, 0].sum()),
                    "waves_marching": int(busy.sum()), "waves": int((rec[:, 0] > 0).sum())}
        if not args.no_extra_legs:
            pn = params(W, H)
            pn.flags |= _abi.FLAG_NO_CULL_RECT
            pipen = pipeline(pn, W, H, K, False)
            sn = max(min(args.steps, 4), 1)
            en = timed_run(pipen, sn, 1, world, cdev, B)
            no_cull = {"ms_per_frame": round(en / (sn * B) * 1e3, 4), "value": round(rays_per_batch * sn / en / 1e6, 2), "unit": "Mrays/s",
                       "what": "the same frames with VRT_FLAG_NO_CULL_RECT: every wave loads the scene and slab-tests its rays"}
            del pipen

    texel_leg = None
    if not args.no_extra_legs and args.steps > 0 and world == 1 and fmt == _abi.FORMAT_F32 and args.path == "auto":
        # the same frames with the volume kept as the reference's own 16-bit texel (RDXVoxelVolume.cpp:399-421), cell records
        for vol in sc.volumes():
            vol.set_device_format(_abi.FORMAT_TEXEL16)
        r.SyncWithScene()
        pt = params(W, H)
        pt.path = _abi.PATH_CELLS
        pipet = pipeline(pt, W, H, K, False)
        st_ = max(min(args.steps, 20), 1)
        et = timed_run(pipet, st_, min(args.warmup, 5), world, cdev, B)
        ct = batch_counts(pt, W, H)
        texel_leg = {"volume_format": "reference texel (sign + 15-bit |d|*100) as 16-byte cell records, --format texel16 --path cells",
                     "ms_per_frame": round(et / (st_ * B) * 1e3, 4),
                     "value": round((ct["primary_rays"] + ct["shadow_rays"]) * st_ / et / 1e6, 2), "unit": "Mrays/s", "streams": K, "frames_per_launch": G}
        del pipet
        for vol in sc.volumes():
            vol.set_device_format(fmt)

    full_leg = None
    if not args.no_extra_legs and args.steps > 0 and world == 1 and args.workload == "c3":
        # the same frames with ONE point light in the scene: the full closest hit (the reference's default render mode as soon as a
        # scene has a point / spot light, a mirroring material or a texture, SH/Raytracing_NoTex.hlsl:41-139) instead of the
        # directional-light kernel.  A block of frames runs it in passes (camera-ray march / light shadow rays + shading / mirror
        # bounces, vrt_kernels.hip primary_pass_kernel); VRT_FLAG_FULL_ONE_KERNEL: the one kernel a lone frame gets.
        import copy
        scl = copy.copy(sc)
        scl.PointLights = [v.VPointLight(Position=(120.0, 60.0, 140.0), Color=(1.0, 0.9, 0.8), IlluminationStrength=40.0)]
        r.SetSceneToRender(scl)
        r.SyncWithScene()
        sf = max(min(args.steps, 10), 1)
        cf = batch_counts(p, W, H)
        full_leg = {"scene": "config 3 + one point light", "frames_per_launch": G, "streams": K, "unit": "Mrays/s"}
        for name, flag in (("passes", 0), ("one_kernel", _abi.FLAG_FULL_ONE_KERNEL)):
            pf = params(W, H)
            pf.flags |= flag
            pipef = pipeline(pf, W, H, K, False)
            ef = timed_run(pipef, sf, min(args.warmup, 3), world, cdev, B)
            full_leg[name] = {"ms_per_frame": round(ef / (sf * B) * 1e3, 4),
                              "value": round((cf["primary_rays"] + cf["shadow_rays"] + cf["bounce_rays"]) * sf / ef / 1e6, 2)}
            del pipef
        r.SetSceneToRender(sc)
        r.SyncWithScene()

    dropin_leg = None
    if not args.no_extra_legs and args.steps > 0 and world == 1 and args.workload == "c3":
        # What a drop-in user gets: the C++ adaptor's DEFAULTS — render mode Interp (the reference's default),
        # the volume as the reference's 16-bit texel, its 1x1 default normal texel on the material, B8G8R8A8 frames, the two reference-artefact
        # flags, MaxBounces 2 — on the benchmark's frames, as a block (one launch per 96 frames) and as a lone frame (one frame per launch, one
        # in flight).  Next to it, on the same box: the lean kernel in the same pixel and volume format (NoTex mode, no texel, no flags) and the
        # form round 4 ran this state in (the texel as an IMAGE: the full closest hit).  A 1x1 texture is a constant folded into the lean
        # kernel's REF instantiation: `kernel_form` says which kernel ran.
        def run_dropin(_):
            for vol in sc.volumes():
                vol.set_device_format(_abi.FORMAT_TEXEL16)
            mat = sc.volumes()[0].Material
            out_fmt = _abi.FLAG_OUTPUT_RGBA8 | _abi.FLAG_OUTPUT_BGRA8
            sd = max(min(args.steps, 10), 1)

            def one(label, mode, flags, normal_tex, g, steps_):
                mat.NormalTexture = normal_tex
                r.SetSceneToRender(sc)
                r.SyncWithScene()
                pd = params(W, H, as_rgba8=True)
                pd.mode, pd.max_bounces = mode, 2
                pd.flags |= out_fmt | flags
                pl = Pipeline(r, pd, W, H, 1, 0, dev, True, 0, 1, False, block_frames=g, cameras=cam_arr, n_cameras=B)
                e = timed_run(pl, steps_, 1, 1, cdev, B)
                form = r.last_kernel_form()
                c = batch_counts(pd, W, H, alone=True)
                del pl
                return {"what": label, "frames_per_launch": g, "ms_per_frame": round(e / (steps_ * B) * 1e3, 4),
                        "value": round((c["primary_rays"] + c["shadow_rays"] + c["bounce_rays"]) * steps_ / e / 1e6, 2), "unit": "Mrays/s",
                        "kernel_form": {"full_closest_hit": bool(form & _abi.FORM_FULL), "passes": bool(form & _abi.FORM_PASSES),
                                        "lean_ref_instantiation": bool(form & _abi.FORM_LEAN_REF), "textured": bool(form & _abi.FORM_TEXTURED)}}

            ref_flags = _abi.FLAG_REFERENCE_VIEW_VECTOR | _abi.FLAG_REFERENCE_BOUNDARY_TEXELS
            texel = workloads.reference_default_normal_texel()
            image = np.ascontiguousarray(np.tile(texel, (2, 2, 1)))  # the same texel as a 2x2 IMAGE: what round 4's host made of it
            try:
                od = {"settings": "mode Interp, VRT_FORMAT_TEXEL16, the reference's 1x1 default normal texel bound, B8G8R8A8 frames, "
                                  "VRT_FLAG_REFERENCE_VIEW_VECTOR | _BOUNDARY_TEXELS, max_bounces 2 (VHipRenderer's defaults)",
                      "block": one("adaptor defaults, one launch per batch", _abi.MODE_INTERP, ref_flags, texel, B, sd),
                      "lone_frame": one("adaptor defaults, one frame per launch, one in flight", _abi.MODE_INTERP, ref_flags, texel, 1, max(sd // 3, 1)),
                      "lean_same_formats": {"block": one("lean kernel: Interp_NoTex, no texel, no flags; same volume and pixel formats", _abi.MODE_INTERP_NOTEX, 0, None, B, sd),
                                            "lone_frame": one("the same, one frame per launch", _abi.MODE_INTERP_NOTEX, 0, None, 1, max(sd // 3, 1))},
                      "round4_form": {"block": one("the texel as a 2x2 image: full closest hit in passes (round 4's path for this state)", _abi.MODE_INTERP, ref_flags, image, B, sd),
                                      "lone_frame": one("the same, one frame per launch (one kernel)", _abi.MODE_INTERP, ref_flags, image, 1, max(sd // 3, 1))}}
                od["block_over_lean"] = round(od["block"]["value"] / od["lean_same_formats"]["block"]["value"], 3)
                od["lone_frame_over_lean"] = round(od["lean_same_formats"]["lone_frame"]["ms_per_frame"] / od["lone_frame"]["ms_per_frame"], 3)
                od["block_over_round4_form"] = round(od["block"]["value"] / od["round4_form"]["block"]["value"], 3)
            finally:
                mat.NormalTexture = None
                for vol in sc.volumes():
                    vol.set_device_format(fmt)
                r.SetSceneToRender(sc)
                r.SyncWithScene()
            return od

        dropin_leg = leg(lambda: None, run_dropin)

    dynamic_leg = None
    if not args.no_extra_legs and args.steps > 0 and world == 1 and args.workload == "c3":
        # Per-frame scene state (vrt_block::scenes): BASELINE config 5's 8 instances ALL moving from frame to frame — the reference moves
        # objects every frame and rebuilds its TLAS every frame (RendererEngineInstance.cpp:111-130, DXRenderer.cpp:809-825) — still
        # ONE march launch per batch (instances, BVH, lights, cull rectangle packed per frame on the host, one copy ahead of the
        # launch), next to the same batch over the standing scene (frame B/2's) with only the camera moving.
        def run_dynamic(_):
            sc5 = workloads.config5_instances(7, 256)
            for vol in sc5.volumes():
                vol.set_device_format(fmt)
            r.SetSceneToRender(sc5)
            r.SyncWithScene()
            frames5 = workloads.moving_instances(sc5, B)
            arr5 = r.scene_array(frames5)
            still = r.scene_array([frames5[B // 2]])
            cams5 = r.camera_array(workloads.orbit_cameras(sc5, B))
            p5 = v.default_params(W, H, workloads.min_cell(sc5), max_steps, shadow=shadow, path=path)
            buf = torch.empty((B, H, W, 4), dtype=torch.float32, device=dev)
            fb = H * W * 16
            out5 = {"scene": "config 5 (8 instanced 128^3 volumes + skybox, BVH), 1920x1080", "frames_per_launch": B, "unit": "Mrays/s"}
            sd = max(min(args.steps, 10), 1)
            lib = _abi.load()
            for name in ("static_scene", "per_frame_scenes"):
                dyn = name == "per_frame_scenes"
                if not dyn:
                    _abi.check(lib.vrt_scene_set(r._ctx, still), "vrt_scene_set")
                rays = 0.0
                for f in range(B):  # counters of the batch, frame by frame (untimed)
                    r.render_block(p5, 1, buf.data_ptr(), fb, 0, **({"scenes": (arr5, f)} if dyn else {"cameras": (cams5, f)}))
                    torch.cuda.synchronize()
                    tt = r.last_timing()
                    rays += tt["primary_rays"] + tt["shadow_rays"]
                kw5 = {"scenes": (arr5, 0)} if dyn else {"cameras": (cams5, 0)}
                for _ in range(2):
                    r.render_block(p5, B, buf.data_ptr(), fb, 0, **kw5)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(sd):
                    r.render_block(p5, B, buf.data_ptr(), fb, 0, **kw5)
                torch.cuda.synchronize()
                e5 = time.perf_counter() - t0
                out5[name] = {"ms_per_frame": round(e5 / (sd * B) * 1e3, 4), "value": round(rays * sd / e5 / 1e6, 2),
                              "launches_per_batch": len([1 for _, fr in r.launch_history(1) if fr == B])}
            out5["per_frame_over_static"] = round(out5["per_frame_scenes"]["value"] / out5["static_scene"]["value"], 3)
            return out5

        dynamic_leg = leg(lambda: None, run_dynamic)
        r.SetSceneToRender(sc)
        r.SyncWithScene()

    coverage_leg = None
    if not args.no_extra_legs and args.steps > 0 and world == 1 and args.workload == "c3":
        # Every pixel marches: the same 256^3 volume with the camera INSIDE its box, 0.6 extents from the centre,
        # looking at it — no sky wave, no wave outside the cull rectangle; the case that stresses the texture path hardest.
        def run_coverage(_):
            ext = float(sc.volumes()[0].VolumeExtends)
            scc = workloads.config3_voxelized(8, 256, distance=0.6 * ext, device_format=fmt)
            scc.Objects[0].Volume = sc.volumes()[0]  # the resident volume (no second upload)
            r.SetSceneToRender(scc)
            r.SyncWithScene()
            camsc = r.camera_array(workloads.orbit_cameras(scc, B))
            buf = torch.empty((B, H, W, 4), dtype=torch.float32, device=dev)
            fb = H * W * 16
            tot = {k: 0.0 for k in KEYS}
            busy_waves = waves = 0
            for f in range(B):
                r.render_block(p, 1, buf.data_ptr(), fb, 0, cameras=(camsc, f))
                torch.cuda.synchronize()
                tt = r.last_timing()
                for k in KEYS:
                    tot[k] += tt[k]
                if f == B // 2:
                    rec = r.wave_records(0).astype(np.int64)
                    busy_waves, waves = int(((rec[:, 3] + rec[:, 4]) > 0).sum()), int((rec[:, 0] > 0).sum())
            sd = max(min(args.steps, 10), 1)
            for _ in range(2):
                r.render_block(p, B, buf.data_ptr(), fb, 0, cameras=(camsc, 0))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(sd):
                r.render_block(p, B, buf.data_ptr(), fb, 0, cameras=(camsc, 0))
            torch.cuda.synchronize()
            ec = time.perf_counter() - t0
            kc = [ms for ms, fr in r.launch_history(sd) if ms > 0.0 and fr == B]
            samples_c = tot["primary_steps"] + tot["shadow_steps"]
            rays_c = tot["primary_rays"] + tot["shadow_rays"]
            outc = {"scene": f"config 3's volume, camera inside its box at 0.6 extents ({0.6 * ext:.1f}) from the centre: every wave marches",
                    "ms_per_frame": round(ec / (sd * B) * 1e3, 4), "value": round(rays_c * sd / ec / 1e6, 2), "unit": "Mrays/s",
                    "waves_marching": busy_waves, "waves": waves, "samples_per_ray": round(samples_c / max(rays_c, 1), 2),
                    "hits_per_frame": int(tot["hits"] / B), "frames_per_launch": B}
            if kc:
                gs = samples_c / (float(np.mean(kc)) * 1e-3) / 1e9
                ge = (samples_c + 6.0 * tot["hits"]) / (float(np.mean(kc)) * 1e-3) / 1e9  # + the 6 trilinear evaluations of every hit's normal
                outc.update({"kernel_ms_per_launch": round(float(np.mean(kc)), 4), "gsamples_per_s": round(gs, 2), "gevaluations_per_s": round(ge, 2),
                             "limiter_frac": round(ge / ceilings["l1_int16" if fmt == _abi.FORMAT_TEXEL16 else "l1"], 4),
                             "limiter_frac_coherent_lanes": round(ge / ceilings["l1_coherent_lanes"], 4),
                             "roofline_frac_algorithmic": round(v.algorithmic_bytes({k: tot[k] for k in KEYS}, 16) / (float(np.mean(kc)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
            return outc

        coverage_leg = leg(lambda: None, run_coverage)
        r.SetSceneToRender(sc)
        r.SyncWithScene()

    if rank == 0:
        alg_bytes = v.algorithmic_bytes(t, 4 if rgba8 else 16) * fpl  # of ONE launch: fpl frames
        gather_key = "l1_int16" if fmt == _abi.FORMAT_TEXEL16 else "l1"  # the active data path's taps
        k_ms = float(np.mean(kms)) if kms else float("nan")
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if kms else 0.0
        samples = int(t["primary_steps"] + t["shadow_steps"]) * fpl
        evals = samples + int(6 * t["hits"]) * fpl
        psteps, ssteps = cnt["primary_steps"], cnt["shadow_steps"]
        key = traffic_key(args, world, K, G, rgba8)
        pmc = measured_counters(key)
        traffic = pmc.get("hbm_bytes_per_launch") if pmc else None
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(sc, p, args.cpu_seconds)
        coherent_key = "l1_int16_coherent_lanes" if (fmt == _abi.FORMAT_TEXEL16 and "l1_int16_coherent_lanes" in ceilings) else "l1_coherent_lanes"
        roofline = {
            # The contract's fields are what the contract defines: ALGORITHMIC bytes (32 B per trilinear sample + 192 B per hit +
            # the pixel store) of one launch / its mean event-timed duration / the HBM peak.  Everything in `measured_in_run` was measured by
            # THIS process on THIS box; everything in `replayed_from_profiles` comes from the committed PMC passes of tools/r05_profile.sh
            # (keyed by the kernel sources' hash and the run's settings) and was NOT measured by this run.
            "bound": "the L1 texture path on depth-incoherent lanes, reached through a latency-limited pipeline: the march runs at "
                     "measured_in_run.limiter_frac of a dependency-free gather of the same bricks timed in this run, and what is missing is the latency of "
                     "a position's dependent chain (cell -> table word -> taps -> step) at the hardware's cap of 8 waves per SIMD.  Sensitivities measured "
                     "on this kernel: 11 % fewer vector instructions per position +1.2 %, half the loads per sample +1.5 %, a third fewer samples +3.4 %, "
                     "24 instead of 32 waves per CU -16 % (profiles/r05_ab_step_asm.txt, r05_data_paths.txt, r04_ab_cell_activity_mask.txt, "
                     "r03_occupancy_sweep_raw.txt); not hbm (replayed_from_profiles.hbm_measured_frac), not mfma (unused)",
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_key": key,
            "frac_note": "frac = algorithmic bytes (32 B per sample, cache-served or not) / launch time / 8 TB/s: an accounting of samples, not of bytes that "
                         "cross the HBM interface (96 % of the taps are L1 hits); it MAY EXCEED 1.  The roof the kernel is under is measured_in_run.limiter_*; "
                         "the physical HBM share is replayed_from_profiles.hbm_measured_frac",
            "kernel": "primary_pass_kernel + light_pass_kernel (the full closest hit of a block of frames in passes; one duration per launch = both)" if args.workload == "c3light" else "march_kernel",
            "kernel_ms": round(k_ms, 4), "frames_per_launch": fpl, "algorithmic_bytes_per_launch": int(alg_bytes),
            "samples_per_launch": samples,
            "measured_in_run": {
                "kernel_ms": round(k_ms, 4),
                "gsamples_per_s": round(samples / (k_ms * 1e-3) / 1e9, 2) if kms else None,
                # every trilinear evaluation the launch makes — march and polish samples AND the 6 of every hit's normal (SURVEY §8d counts
                # them: 192 B per hit) — per second
                "trilinear_evaluations_per_launch": evals,
                "gevaluations_per_s": round(evals / (k_ms * 1e-3) / 1e9, 2) if kms else None,
                # the march's inner operation in isolation, on this box, right before the timed region (vrt_debug_gather_ceiling):
                # independent cells per lane (by where the bricks live; fp32 and int16 bricks) and the 64 lanes of a wave on the 3x3 cells
                # of an 8x8-pixel tile
                "gather_ceiling_gsamples_per_s": ceilings, "gather_ceiling_measured_in_run": ceilings_measured,
                "limiter": "L1-served trilinear gather (the texture path)",
                "limiter_ceiling_gsamples_per_s": ceilings[gather_key],
                "limiter_frac": round(evals / (k_ms * 1e-3) / 1e9 / ceilings[gather_key], 4) if kms else None,
                # ... and against a gather whose 64 lanes are tile-coherent (the march's lanes lie between the two)
                "limiter_ceiling_coherent_lanes_gsamples_per_s": ceilings[coherent_key],
                "limiter_frac_coherent_lanes": round(evals / (k_ms * 1e-3) / 1e9 / ceilings[coherent_key], 4) if kms else None,
                # measured in rounds 4 and 5 (profiles/r04_ab_cell_activity_mask.txt, r05_data_paths.txt, r05_ab_placement.txt): the evaluation
                # RATE is the march's speed against the gather microbenchmark, not what its time is made of — a third fewer samples for
                # bit-identical frames: +3.4 %; half the loads per sample: +1.5 %; no re-placement of lanes or bricks changes the 19 lines per load
                "limiter_note": "limiter_frac is the march's evaluation rate against an L1-served gather of independent cells measured in this run; it is a "
                                "yardstick, not headroom: fewer samples (+3.4 % for a third fewer), fewer loads (+1.5 % for half) and re-placed lanes or "
                                "bricks (-0.4 .. -9.5 %) were all measured, and the 19 cache lines per wave-level load come from the depth spread of a "
                                "wave's lanes, not from their screen footprint",
            },
        }
        # (kept at the top level too: the driver's records of earlier rounds read them there)
        roofline["limiter_frac"] = roofline["measured_in_run"]["limiter_frac"]
        roofline["limiter_ceiling_gsamples_per_s"] = roofline["measured_in_run"]["limiter_ceiling_gsamples_per_s"]
        roofline["limiter_frac_coherent_lanes"] = roofline["measured_in_run"]["limiter_frac_coherent_lanes"]
        if pmc and kms:
            # physical picture, from the keyed PMC passes (profiles/counters_latest.json): what really crosses the HBM interface, how busy
            # the texture path and the vector issue are.  REPLAYED: taken by the profiler on this kernel source with these settings, not by this run
            cyc = pmc.get("gpu_cycles_per_launch")  # GRBM_GUI_ACTIVE / 8 XCDs, taken under the counters' own (serialised) run
            clock = float(pmc.get("clock_ghz") or 2.4)
            valu = pmc.get("SQ_INSTS_VALU")
            roofline["replayed_from_profiles"] = {
                "tag": pmc.get("tag"), "file": "profiles/counters_latest.json", "kernel_source_sha": key["kernel_source_sha"],
                "what": "rocprofv3 --pmc passes of this very command on this kernel source (tools/r05_profile.sh); combined with THIS run's kernel_ms where a rate is formed",
                "hbm_bytes_per_launch": traffic,
                "hbm_measured_GBps": round(traffic / (k_ms * 1e-3) / 1e9, 2) if traffic else None,
                "hbm_measured_frac": round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                "hbm_floor_us_per_frame": round(traffic / fpl / (HBM_PEAK_GBS * 1e9) * 1e6, 2) if traffic else None,
                # wave-level vector instructions x 2 cycles (a wave64 op on a 32-lane SIMD, MI355X_MICROARCH.md) / 1024 SIMDs / clock /
                # the launch's duration in THIS run
                "valu_issue_frac": round(valu * 2.0 / 1024.0 / (clock * 1e9) / (k_ms * 1e-3), 4) if valu else None,
                "valu_insts_per_launch": int(valu) if valu else None, "clock_ghz": round(clock, 3),
                "occupancy_mean_waves_per_cu": pmc.get("occupancy_mean_waves_per_cu"),
                # busy cycles of the texture path's two units summed over the CUs / 256 / the launch's GPU cycles (in the counters' own run)
                "td_busy_frac": pmc.get("td_busy_frac"), "ta_busy_frac": pmc.get("ta_busy_frac"),
                # cache-line accesses the L1 (TCP) serves per CU and cycle (TCP_TOTAL_CACHE_ACCESSES_sum / 256 CUs / GPU cycles) and per
                # wave-level vector-memory read instruction (.. / SQ_INSTS_VMEM_RD): ~4 if the 64 lanes of a load read one brick
                "l1_cache_line_accesses_per_cu_cycle": pmc.get("l1_cache_line_accesses_per_cu_cycle"),
                "lines_per_vmem_instr": pmc.get("lines_per_vmem_instr"),
                "valu_lane_utilisation": pmc.get("valu_lane_utilisation"),
                "gpu_cycles_per_launch": cyc,
            }
        if latency and latency.get("kernel_ms_per_frame") and cpu and cpu.get("chain_positions_max"):
            # the lone frame: its longest chain of dependent positions (oracle, same frame) x the time a position takes
            roofline["latency_model"] = {"chain_positions_max": cpu["chain_positions_max"],
                                         "lone_frame_kernel_us": round(latency["kernel_ms_per_frame"] * 1e3, 1),
                                         "us_per_position": round(latency["kernel_ms_per_frame"] * 1e3 / cpu["chain_positions_max"], 3)}
        out = {
            "metric": "Mrays/sec at 1080p, 256^3 SDF volume" if args.workload in ("c3", "c3sdf") else "Mrays/sec",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "ms_per_frame": round(ms_per_step / (B * L), 4), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": label, "width": W, "height": H, "volume": f"{sc.volumes()[0].N - 1}^3 cells",
                       "volume_format": {_abi.FORMAT_F32: "f32 bricks (512 B per 4^3 cells)",
                                         _abi.FORMAT_TEXEL16: "reference texel: sign + 15-bit |d|*100, 16-bit bricks (256 B per 4^3 cells)"}[fmt],
                       "max_steps": max_steps, "shadow": bool(shadow), "k_relax": round(float(p.k_relax), 3), "data_path": args.path,
                       "output": "rgba8 (R8G8B8A8_UNORM tiles; march and shading in f32)" if rgba8 else "f32 (float4)",
                       "step": f"{L} pass(es) over a batch of {B} frames (consecutive views of a camera orbiting the workload's view, 0.25 degrees "
                               f"apart): {B * L} frames per step, one march launch per {G} frames",
                       "frames_per_step": B * L, "frames_per_batch": B, "launches_per_step": (B * L + G - 1) // G, "streams": K, "frames_per_launch": G,
                       "parallelism": ("1 GPU" if world == 1 else
                                       (f"{strip_rows}-row interleaved strips" if strip_rows else "contiguous row tiles") +
                                       f" x{world} + " + exchange_label(rotate, world, G, rehearsal)),
                       "rays_per_step": int(rays_per_step), "rays_per_frame": int(rays_per_frame),
                       # (the waves of a frame that lie outside the host's cull rectangle go straight to the sky: their pixels count as
                       # primary rays, like the reference's DispatchRays(W, H) counts them; how many rays sit in waves that march:)
                       "marching": marching,
                       "samples_per_ray": round((psteps + ssteps) / max(rays_per_batch, 1), 2)},
            "roofline": roofline, "cpu_baseline": cpu,
            "latency": latency, "scale_anchor": scale_anchor, "end_to_end": end_to_end, "config4": config4, "reference_texel_format": texel_leg, "full_closest_hit": full_leg,
            "no_cull_rect": no_cull, "dynamic_scene": dynamic_leg, "full_coverage": coverage_leg, "drop_in_defaults": dropin_leg,
        }
        if world > 1:
            out["speedup_vs_anchor"] = round(value / scale_anchor["value"], 3) if scale_anchor and scale_anchor.get("value") else None
            out["collective"] = ("native (vrt_gather_tiles / vrt_exchange_tiles on the march stream)" if use_native else
                                 "torch.distributed (" + ("gloo, REHEARSAL" if rehearsal else "RCCL") + ")")
            out["native_gather_check"] = native_check if native_check is not None else ({"error": native_error} if native_error else None)
            out["other_exchange"] = exchange_other
            if exchange_fallback:
                out["exchange_fallback"] = exchange_fallback
            if native_fallback:
                out["collective_fallback"] = native_fallback
            out.update(multi_gpu_summary(value, scale_anchor.get("value") if scale_anchor else None, rotate, exchange_other, world, W, H,
                                         4 if rgba8 else 16, rays_per_frame, max(min(args.steps, 3), 1)))
            # what the block's time is made of on a rank, to be read against the link model at a glance: this rank's event-timed march per
            # launch (max / min over the ranks), and — native exchange only: it runs on the march stream — the collective's own duration
            out["per_rank"] = per_rank
            out["scaling_curve"] = ("this line is ONE point: a 1 -> N curve exists only where the driver's SCALE record measured N = 1, 2, 4, 8 back to back; "
                                    "no round of this build has had more than one physical GPU")
        if verified is not None:
            out["assembled_frame_equals_single_gpu_frame"] = verified
        if probe_abandoned:
            out["native_probe_abandoned"] = True  # its thread is still inside vrt_comm_init / RCCL on a context of its own: the process leaves without destructors
        print(json.dumps(out), flush=True)

    r.Stop()
    if world > 1:
        dist.destroy_process_group()
    if probe_abandoned:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


MULTI_STREAMS = 2          # N > 1: streams per rank (a block's collective overlaps the next block's march)
ANCHOR_BLOCK_FRAMES = 48   # frames per launch of the one-GPU scale anchor (= the N-GPU block for N = 2, 4, 8)


def multi_block_frames(world: int) -> int:
    """Frames per block (= per march launch and per collective) of an N-GPU run: about 48, a multiple of N (rotating roots deal
    whole frames to the ranks).  A rank's eighth of 24 frames is 130 us of march with a 60-us tail behind it; per-rank probe
    (profiles/r03_strong_scaling_probe.txt, N = 8, 2 streams): 24 frames per launch 0.0055 ms/frame, 48 0.0053, 96 0.0051."""
    return world * max(1, round(48 / world))


def exchange_label(rotate: bool, world: int, G: int, rehearsal: bool) -> str:
    how = "gloo, REHEARSAL on one GPU" if rehearsal else "RCCL"
    if rotate:
        return f"one all-to-all ({how}) per block of {G} frames: frame g of a block is assembled on rank g // {max(G // world, 1)}"
    return f"one gather ({how}) to rank 0 per block of {G} frames"


def end_to_end_leg(r, p, rays_per_frame: float, steps: int, slots: int = 3):
    """The frame as a host application gets it: vrt_render_begin / vrt_render_end with the reference's three frame slots
    (FrameCount, DXConstants.cpp:23) — march, then the copy of the frame into pinned host memory, the next frames' marches
    overlapping the copy."""
    for i in range(slots):
        r.render_begin(i % slots, p)
    for i in range(slots):
        r.render_end(i % slots, p, copy=False)
    t0 = time.perf_counter()
    for i in range(steps):
        if i >= slots:
            r.render_end(i % slots, p, copy=False)
        r.render_begin(i % slots, p)
    for i in range(max(steps - slots, 0), steps):
        r.render_end(i % slots, p, copy=False)
    dt = time.perf_counter() - t0
    bpp = 4 if p.flags & 8 else 16
    return {"value": round(rays_per_frame * steps / dt / 1e6, 2), "unit": "Mrays/s", "ms_per_frame": round(dt / steps * 1e3, 4),
            "includes": f"march + D2H of the {p.width}x{p.height} frame ({bpp} B/pixel) into pinned host memory, "
                        f"vrt_render_begin/_end with {slots} frame slots (scene re-sent every frame)"}


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
    # the longest chain of dependent march positions a lane of the GPU frame runs (primary ray + the rays after it): one whole
    # frame at full size with the oracle's debug position image (the oracle visits exactly the positions the kernel visits)
    chain = None
    try:
        import ctypes as C

        import numpy as np

        from oracle import binding
        lib = binding.load()
        lib.vrto_debug_set_steps_image.argtypes = [C.c_void_p]
        img = np.zeros((p.height, p.width), np.uint32)
        lib.vrto_debug_set_steps_image(img.ctypes.data)
        try:
            o.render(p, threads=cores)
        finally:
            lib.vrto_debug_set_steps_image(None)
        chain = int(((img & 0xFFFF).astype(np.int64) + (img >> 16).astype(np.int64)).max())
    except Exception:  # noqa: BLE001
        chain = None
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
        "chain_positions_max": chain,
    }


if __name__ == "__main__":
    main()
