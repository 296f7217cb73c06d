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


def _worker(rank, world, port, height, width, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import scenes
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

    import scenes
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
