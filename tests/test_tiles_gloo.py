"""Multi-rank row-tile path on CPU: world_size 2 and 3 over gloo.  Each rank renders its tile
(here with the CPU oracle — on the GPU the tile comes from vrt_render_rows), the tiles are gathered
with the same FrameGather bench.py uses, and rank 0 must hold exactly the single-rank frame."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def quantize_rgba8(img):
    """The kernel's VRT_FLAG_OUTPUT_RGBA8 rule restated: (uint)(min(c,1)*255 + 0.5), fp32 mul then add."""
    c = np.minimum(np.asarray(img, dtype=np.float32), np.float32(1.0))
    return (c * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)


def _strip_worker(rank, world, port, height, width, strip_rows, as_u8, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    from volumetricraytracer_amd import workloads as scenes
    import volumetricraytracer_amd as v
    from oracle.binding import OracleScene
    from volumetricraytracer_amd.tiles import FrameGather, strip_frame_rows

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = scenes.config5_instances(5, 16)
        p = v.default_params(width, height, scenes.min_cell(sc), 255, shadow=True)
        fg = FrameGather(height, width, world, rank, torch.device("cpu"), dtype=torch.uint8 if as_u8 else torch.float32,
                         strip_rows=strip_rows)
        o = OracleScene(sc)
        for b in range(2):
            for local0, frame0, rows in strip_frame_rows(height, world, rank, strip_rows):
                tile, _ = o.render(p, frame0, rows)  # on the GPU: one vrt_render_strips launch for all strips
                fg.tiles[b][local0:local0 + rows] = torch.from_numpy(quantize_rgba8(tile) if as_u8 else tile)
            fg.gather(b, async_op=True).wait()
            fg.unshuffle(b)
        if rank == 0:
            np.save(out_path, fg.frame(1).numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, height, width, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    from volumetricraytracer_amd import workloads as scenes
    import volumetricraytracer_amd as v
    from oracle.binding import OracleScene
    from volumetricraytracer_amd.tiles import FrameGather

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = scenes.config5_instances(5, 16)
        p = v.default_params(width, height, scenes.min_cell(sc), 255, shadow=True)
        fg = FrameGather(height, width, world, rank, torch.device("cpu"))
        for b in range(2):  # both buffers, like the pipelined bench loop
            if fg.rows > 0:
                tile, _ = OracleScene(sc).render(p, fg.row0, fg.rows)
                fg.tiles[b][: fg.rows] = torch.from_numpy(tile)
            work = fg.gather(b, async_op=True)
            work.wait()
        if rank == 0:
            np.save(out_path, fg.frame(1).numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 54), (3, 50), (2, 1)])
def test_row_tiles_gather_matches_single_rank(tmp_path, oracle_lib, world, height):
    import torch.multiprocessing as mp

    from volumetricraytracer_amd import workloads as scenes
    import volumetricraytracer_amd as v
    from oracle.binding import OracleScene

    width = 96
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), height, width, out), nprocs=world, join=True)
    sc = scenes.config5_instances(5, 16)
    p = v.default_params(width, height, scenes.min_cell(sc), 255, shadow=True)
    ref, _ = OracleScene(sc).render(p)
    got = np.load(out)
    assert got.shape == (height, width, 4)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("world,height,strip_rows,as_u8", [(2, 54, 8, False), (3, 50, 4, False), (2, 37, 16, True)])
def test_interleaved_strips_gather_matches_single_rank(tmp_path, oracle_lib, world, height, strip_rows, as_u8):
    """Strips dealt round-robin to the ranks (ragged last strip, empty strip slots), gathered and
    un-shuffled: rank 0 holds exactly the single-rank frame (float, or the RGBA8 exchange format)."""
    import torch.multiprocessing as mp

    from volumetricraytracer_amd import workloads as scenes
    import volumetricraytracer_amd as v
    from oracle.binding import OracleScene

    width = 96
    out = str(tmp_path / "frame.npy")
    mp.spawn(_strip_worker, args=(world, _free_port(), height, width, strip_rows, as_u8, out), nprocs=world, join=True)
    sc = scenes.config5_instances(5, 16)
    p = v.default_params(width, height, scenes.min_cell(sc), 255, shadow=True)
    ref, _ = OracleScene(sc).render(p)
    if as_u8:
        ref = quantize_rgba8(ref)
    got = np.load(out)
    assert got.shape == (height, width, 4) and got.dtype == ref.dtype
    assert np.array_equal(got, ref)


def test_strip_layout_partition():
    from volumetricraytracer_amd.tiles import strip_frame_rows, strip_layout

    for h in (0, 1, 31, 32, 33, 1080, 2160, 3054):
        for w in (1, 2, 3, 4, 8):
            for sr in (1, 16, 32):
                total, per = strip_layout(h, w, sr)
                assert total == (h + sr - 1) // sr and per * w >= total and (per - 1) * w < max(total, 1)
                cover = []
                for r in range(w):
                    for local0, frame0, rows in strip_frame_rows(h, w, r, sr):
                        assert local0 % sr == 0 and local0 // sr < per and 0 < rows <= sr
                        assert (frame0 // sr) % w == r  # strip index dealt round-robin
                        cover += list(range(frame0, frame0 + rows))
                assert sorted(cover) == list(range(h))
    assert strip_layout(3054, 8, 32) == (96, 12)  # 8x the 1080p ray count at 16:9 -> 12 strips of 32 rows per GPU
    with pytest.raises(ValueError):
        strip_layout(10, 2, 0)


def test_tile_rows_partition():
    from volumetricraytracer_amd.tiles import tile_rows

    for h in (0, 1, 7, 135, 1080, 2160):
        for w in (1, 2, 3, 4, 8):
            cover = []
            for r in range(w):
                rows_per, row0, rows = tile_rows(h, w, r)
                assert rows_per == (h + w - 1) // w and 0 <= rows <= rows_per
                cover += list(range(row0, row0 + rows))
            assert cover == list(range(h))
    assert tile_rows(1080, 8, 3) == (135, 405, 135)  # BASELINE: 1080p on 8 GPUs -> 135 rows each
    assert tile_rows(2160, 8, 7) == (270, 1890, 270)  # config 4: 4K -> 270 rows each
    with pytest.raises(ValueError):
        tile_rows(10, 2, 2)


@pytest.mark.parametrize("world,height,strip_rows", [(8, 3055, 32), (8, 1080, 32), (4, 2160, 32), (5, 77, 8)])
def test_unshuffle_puts_every_strip_where_it_belongs(world, height, strip_rows):
    """Rank 0's un-shuffle at the bench's real shapes (8 ranks, 3055 rows, ...) without processes: the gathered buffer is
    filled the way the ranks' compact tiles arrive (rank-major, strip after strip), every row tagged with its frame row."""
    import torch

    from volumetricraytracer_amd.tiles import FrameGather, strip_frame_rows

    width = 3
    fg = FrameGather(height, width, world, 0, torch.device("cpu"), dtype=torch.float32, buffers=1, strip_rows=strip_rows)
    fg.frames[0].fill_(-1.0)
    for rank in range(world):
        tile = fg.frames[0][rank * fg.rows_per:(rank + 1) * fg.rows_per]
        for local0, frame0, rows in strip_frame_rows(height, world, rank, strip_rows):
            tile[local0:local0 + rows] = torch.arange(frame0, frame0 + rows, dtype=torch.float32)[:, None, None]
    fg.unshuffle(0)
    got = fg.frame(0)
    assert got.shape == (height, width, 4)
    assert torch.equal(got[:, 0, 0], torch.arange(height, dtype=torch.float32))


@pytest.mark.parametrize("world,height,strip_rows,G", [(8, 1080, 32, 8), (4, 2160, 32, 3), (3, 50, 4, 5), (2, 54, 0, 4), (8, 1080, 0, 8)])
def test_block_of_frames_per_gather(world, height, strip_rows, G):
    """frames_per_gather = G: a buffer holds G frames' tiles and ONE gather moves them (bench.py, N > 1).  The gathered buffer
    is filled the way the ranks' blocks arrive (rank-major, frame after frame, strip after strip), every row tagged with
    (frame of the block, frame row): tile(b, g) addresses the right slice and frame(b, g) is frame g in row order."""
    import torch

    from volumetricraytracer_amd.tiles import FrameGather, strip_frame_rows, tile_rows

    width = 2
    fg = FrameGather(height, width, world, 0, torch.device("cpu"), dtype=torch.float32, buffers=1, strip_rows=strip_rows, frames_per_gather=G)
    assert fg.tiles[0].shape[0] == G * fg.rows_per and fg.tile(0, G - 1).data_ptr() == fg.tiles[0][(G - 1) * fg.rows_per:].data_ptr()
    fg.frames[0].fill_(-1.0)
    for rank in range(world):
        block = fg.frames[0][rank * G * fg.rows_per:(rank + 1) * G * fg.rows_per]
        for g in range(G):
            tile = block[g * fg.rows_per:(g + 1) * fg.rows_per]
            if strip_rows > 0:
                spans = strip_frame_rows(height, world, rank, strip_rows)
            else:
                _, row0, rows = tile_rows(height, world, rank)
                spans = [(0, row0, rows)] if rows > 0 else []
            for local0, frame0, rows in spans:
                tile[local0:local0 + rows] = (1000.0 * g + torch.arange(frame0, frame0 + rows, dtype=torch.float32))[:, None, None]
    fg.unshuffle(0)
    for g in range(G):
        got = fg.frame(0, g)
        assert got.shape == (height, width, 4)
        assert torch.equal(got[:, 1, 2], 1000.0 * g + torch.arange(height, dtype=torch.float32)), g


def _rotate_worker(rank, world, port, height, width, strip_rows, m, out_dir):
    """One rank of the rotating-root exchange over gloo: tiles of a block of G = m x world frames, every row tagged with
    (frame of the block, frame row); after ONE all-to-all and the un-shuffle this rank holds ITS m frames whole."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from volumetricraytracer_amd.tiles import FrameGather, strip_frame_rows, tile_rows

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        G = m * world
        fg = FrameGather(height, width, world, rank, torch.device("cpu"), dtype=torch.float32, buffers=2, strip_rows=strip_rows,
                         frames_per_gather=G, rotate_roots=True)
        assert fg.rotate and fg.m == m and [fg.root_of(g) for g in range(G)] == [g // m for g in range(G)]
        if strip_rows > 0:
            spans = strip_frame_rows(height, world, rank, strip_rows)
        else:
            _, row0, rows = tile_rows(height, world, rank)
            spans = [(0, row0, rows)] if rows > 0 else []
        for b in range(2):
            for g in range(G):
                tile = fg.tile(b, g)
                tile.fill_(-1.0)
                for local0, frame0, rows in spans:
                    tile[local0:local0 + rows] = (100000.0 * b + 1000.0 * g + torch.arange(frame0, frame0 + rows, dtype=torch.float32))[:, None, None]
            fg.gather(b, async_op=True).wait()
            fg.unshuffle(b)
        ok = True
        for b in range(2):
            for g in range(G):
                got = fg.frame(b, g)
                if fg.root_of(g) != rank:
                    ok = ok and got is None
                    continue
                want = 100000.0 * b + 1000.0 * g + torch.arange(height, dtype=torch.float32)
                ok = ok and got.shape == (height, width, 4) and bool(torch.equal(got[:, width - 1, 3], want))
        np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.array([1 if ok else 0]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,height,strip_rows,m", [(2, 54, 8, 2), (3, 50, 4, 1), (2, 37, 0, 3), (3, 1080, 8, 2)])
def test_rotating_roots_exchange_assembles_every_frame_on_its_rank(tmp_path, world, height, strip_rows, m):
    """FrameGather(rotate_roots=True): ONE all_to_all_single per block of m x world frames; frame g ends up whole on rank
    g // m (strips un-shuffled there), with ragged last strips, empty strip slots and contiguous tiles; nobody else has it."""
    import torch.multiprocessing as mp

    mp.spawn(_rotate_worker, args=(world, _free_port(), height, 5, strip_rows, m, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        assert int(np.load(str(tmp_path / f"ok{rank}.npy"))[0]) == 1, rank


def test_rotating_roots_reject_a_block_that_does_not_deal_out():
    import torch

    from volumetricraytracer_amd.tiles import FrameGather

    with pytest.raises(ValueError):
        FrameGather(16, 4, 3, 0, torch.device("cpu"), frames_per_gather=4, rotate_roots=True)
    # one rank: nothing to rotate
    fg = FrameGather(16, 4, 1, 0, torch.device("cpu"), frames_per_gather=4, rotate_roots=True)
    assert not fg.rotate and fg.root_of(3) == 0 and fg.frame(0, 3).shape == (16, 4, 4)


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with WORLD_SIZE unset (how the driver calls it) must start two ranks itself: the launch
    path without a march (--launch-check: gloo rendezvous, strip layout, gather, un-shuffle, max-over-ranks), runnable
    without a GPU.  A rank whose WORLD_SIZE differs from --gpus is refused with a non-zero exit code."""
    import json
    import subprocess

    env = {k: val for k, val in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    bench = os.path.join(ROOT, "bench.py")
    res = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "2", "--launch-check"], env=env, capture_output=True,
                         text=True, timeout=300, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert out == {**out, "launch_check": True, "n_gpus": 2, "ranks_joined": 2, "gathered_frame_ok": True}
    # the N > 1 line states both exchanges, their speed-up over the one-GPU anchor and the link model's cap for the rank-0 gather
    assert out["multi_gpu_line_keys"] == ["exchanges", "link_model", "north_star_6x_at_8_gpus"]
    sample = out["multi_gpu_line_sample"]
    assert set(sample["exchanges"]) == {"rotate", "gather"} and sample["exchanges"]["rotate"]["main_line"] and not sample["exchanges"]["gather"]["main_line"]
    assert abs(sample["exchanges"]["rotate"]["speedup_vs_anchor"] - 400000.0 / 62000.0) < 1e-3
    # (N-1)/N of every 8.3-MB frame over the root's N-1 inbound links: N * 77 GB/s * 0.8 / frame bytes frames per second
    want = 2 * 77e9 * 0.8 / (1920 * 1080 * 4) * 2.4e6 / 1e6 / 62000.0
    assert abs(sample["link_model"]["gather_to_rank0_cap_speedup_vs_anchor"] - want) < 0.01
    bad = subprocess.run([sys.executable, bench, "--gpus", "2", "--launch-check"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                         capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert bad.returncode == 2 and "refusing" in bad.stderr


def test_bench_cli_parses_and_prints_help():
    """bench.py --help (a percent sign in a help string once broke it) and the defaults the driver relies on."""
    import subprocess

    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and "--frames-per-step" in res.stdout and "--gpus" in res.stdout
    sys.path.insert(0, ROOT)
    import bench

    a = bench.parse_args([])
    assert (a.gpus, a.frames_per_step, a.strip_rows, a.scaling, a.workload) == (1, 96, 8, "strong", "c3")
    a = bench.parse_args(["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert (a.gpus, a.steps, a.warmup) == (8, 20, 5)


def test_bench_block_plan_issues_exactly_the_frames_asked_for():
    """bench.py times EXACTLY K steps: the blocks of a run add up to steps x frames per step, none is larger than the block
    size, whole rounds come first and the last round is dealt evenly."""
    sys.path.insert(0, ROOT)
    import bench

    for frames in (0, 1, 5, 63, 64, 65, 20 * 64, 100 * 64 + 7):
        for G, K in ((2, 3), (8, 8), (1, 1), (8, 3), (64, 8)):
            plan = bench.block_plan(frames, G, K)
            assert sum(plan) == frames and all(1 <= n <= G for n in plan)
            full = frames // (G * K) * K
            assert plan[:full] == [G] * full
            tail = plan[full:]
            assert len(tail) <= K and (not tail or max(tail) - min(tail) <= 1)
    assert bench.block_plan(100, 8, 8) == [8] * 8 + [5, 5, 5, 5, 4, 4, 4, 4]
