"""Row-tile partition of a frame across ranks and the gather of the tiles onto rank 0 — the
multi-GPU step of the hot path (the reference is single-adapter, NodeMask = 0).

Every rank renders rows [row0, row0+rows) of the same frame from global pixel coordinates, so the
tiles are seamless.  The tiles are equal-sized buffers (rows_per = ceil(H / world)), gathered with
ONE collective (`torch.distributed.gather`; backend "nccl" is RCCL over xGMI on MI355X, "gloo" on
CPU for tests) and trimmed to H rows on the destination."""
from __future__ import annotations

from typing import List, Optional, Tuple


def tile_rows(height: int, world: int, rank: int) -> Tuple[int, int, int]:
    """(rows_per, row0, rows) of `rank`: contiguous tiles of ceil(H/world) rows; trailing ranks may
    get fewer (or zero) rows when world does not divide H."""
    if height < 0 or world < 1 or not 0 <= rank < world:
        raise ValueError("bad tile request")
    rows_per = (height + world - 1) // world
    row0 = min(rank * rows_per, height)
    rows = max(0, min(rows_per, height - row0))
    return rows_per, row0, rows


class FrameGather:
    """Owns the destination frame(s) on rank 0 and issues the gather of one tile per rank."""

    def __init__(self, height: int, width: int, world: int, rank: int, device, dtype=None, buffers: int = 2):
        import torch

        self.height, self.width, self.world, self.rank = height, width, world, rank
        self.rows_per, self.row0, self.rows = tile_rows(height, world, rank)
        dtype = dtype or torch.float32
        self.tiles = [torch.zeros((self.rows_per, width, 4), dtype=dtype, device=device) for _ in range(buffers)]
        self.frames: Optional[List] = None
        if world > 1 and rank == 0:
            self.frames = [torch.empty((world * self.rows_per, width, 4), dtype=dtype, device=device) for _ in range(buffers)]

    def gather(self, b: int, async_op: bool = False):
        """Gather tile buffer `b` of every rank into frame buffer `b` on rank 0."""
        import torch.distributed as dist

        if self.world == 1:
            return None
        glist = None
        if self.rank == 0:
            glist = [self.frames[b][k * self.rows_per:(k + 1) * self.rows_per] for k in range(self.world)]
        return dist.gather(self.tiles[b], glist, dst=0, async_op=async_op)

    def frame(self, b: int):
        """The assembled H x W x 4 frame (rank 0 only; for world == 1 it is the tile itself)."""
        if self.world == 1:
            return self.tiles[b][: self.height]
        if self.rank != 0:
            return None
        return self.frames[b][: self.height]
