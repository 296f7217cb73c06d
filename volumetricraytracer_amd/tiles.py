"""Partition of a frame across ranks and the gather of the tiles onto rank 0 — the multi-GPU step
of the hot path (the reference is single-adapter, NodeMask = 0).

Two partitions, both seamless because every rank renders from global pixel coordinates:

* rows   — rank g renders the contiguous rows [row0, row0+rows) (`vrt_render_rows`);
* strips — the frame is cut into strips of `strip_rows` rows and rank g renders strips g, g+n, g+2n …
           into a compact tile (`vrt_render_strips`).  Contiguous tiles put every object row on the
           middle GPUs and sky on the outer ones; interleaving evens the load (SURVEY §8e "risk").

The tiles are equal-sized buffers gathered with ONE collective: `torch.distributed.gather` (backend
"nccl" is RCCL over xGMI on MI355X, "gloo" on CPU for tests), or — `native_gather` — the C-ABI's own
`vrt_gather_tiles` (ncclGather on the march stream, no torch in the data path).  For strips rank 0 then
un-shuffles the gathered [rank, strip] order into frame order with one strided device copy.  A buffer may hold
a BLOCK of several frames' tiles (`frames_per_gather`): one collective then moves the whole block — what
bench.py does, because at a few tens of microseconds per frame the per-call cost of a collective bounds the job.

`rotate_roots`: the frames of a block are assembled on DIFFERENT ranks — frame g of a block of G = m x world frames on rank
g // m — with ONE all-to-all per block (`torch.distributed.all_to_all_single`, or `vrt_exchange_tiles`: grouped
ncclSend / ncclRecv).  A gather onto one rank moves (world - 1) / world of every frame over that rank's inbound xGMI links:
8.3 MB of RGBA8 per 1080p frame against about 0.5 TB/s of inbound links caps the job near 60 000 frames/s whatever the march
does, while 8 GPUs march 160 000; spread over all ranks the same bytes use every link of the node.  Each frame still ends up
whole on one GPU."""
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


def strip_layout(height: int, world: int, strip_rows: int) -> Tuple[int, int]:
    """(strips in the frame, strips per rank).  Every rank owns the same number of strip slots; slots
    whose rows fall beyond the frame stay empty (the kernel skips them)."""
    if height < 0 or world < 1 or strip_rows < 1:
        raise ValueError("bad strip request")
    total = (height + strip_rows - 1) // strip_rows
    return total, (total + world - 1) // world


def strip_frame_rows(height: int, world: int, rank: int, strip_rows: int) -> List[Tuple[int, int, int]]:
    """[(local_row0, frame_row0, rows)] of the strips rank `rank` renders (clipped to the frame)."""
    if not 0 <= rank < world:
        raise ValueError("bad rank")
    _, per = strip_layout(height, world, strip_rows)
    out = []
    for s in range(per):
        fr = (s * world + rank) * strip_rows
        rows = max(0, min(strip_rows, height - fr))
        if rows > 0:
            out.append((s * strip_rows, fr, rows))
    return out


class FrameGather:
    """Owns this rank's tile buffers and, on rank 0, the gathered frame(s); issues the gather of one
    tile per rank.  `strip_rows` = 0: contiguous row tiles; > 0: interleaved strips of that height.
    `dtype` float32 → [rows, W, 4] float RGBA; uint8 → [rows, W, 4] R8G8B8A8."""

    def __init__(self, height: int, width: int, world: int, rank: int, device, dtype=None, buffers: int = 2,
                 strip_rows: int = 0, frames_per_gather: int = 1, rotate_roots: bool = False):
        """frames_per_gather = G > 1: a buffer holds a BLOCK of G frames' tiles ([G, rows, W, 4]) and one gather moves the whole
        block (fewer, larger collectives: at a few tens of microseconds per frame the per-call cost of a collective is what
        bounds an N-GPU job); tile(b, g) / frame(b, g) address frame g of block b.
        rotate_roots (G a multiple of world, m = G / world): frame g of a block is assembled on rank g // m by one all-to-all
        per block (exchange()); every rank then holds ITS m frames: frame(b, g) is valid on rank root_of(g) only."""
        import torch

        self.height, self.width, self.world, self.rank = height, width, world, rank
        self.rotate = bool(rotate_roots) and world > 1
        if self.rotate and (frames_per_gather < world or frames_per_gather % world != 0):
            raise ValueError("rotate_roots needs a block of a multiple of `world` frames")
        self.strip_rows = int(strip_rows)
        if self.strip_rows > 0:
            self.total_strips, self.strips_per = strip_layout(height, world, self.strip_rows)
            self.rows_per = self.strips_per * self.strip_rows
            self.row0, self.rows = 0, self.rows_per  # compact tile; see strip_frame_rows for the frame rows
        else:
            self.rows_per, self.row0, self.rows = tile_rows(height, world, rank)
        dtype = dtype or torch.float32
        G = self.G = max(int(frames_per_gather), 1)
        self.tiles = [torch.zeros((G * self.rows_per, width, 4), dtype=dtype, device=device) for _ in range(buffers)]
        self.frames: Optional[List] = None  # gathered tiles, rank-major
        self.final: Optional[List] = None   # strips only: frame-ordered copies
        self._glist = None      # per buffer: the views torch.distributed.gather receives into (built once, not per frame)
        self._unshuffle = None  # per buffer: (source view, destination view) of the strip un-shuffle
        self.m = G // world if self.rotate else G  # frames of a block this rank assembles
        if self.rotate:
            m = self.m
            # received: [source rank, my frame of the block, rows of a tile] — the layout a gather of my m frames would give
            self.frames = [torch.zeros((world * m * self.rows_per, width, 4), dtype=dtype, device=device) for _ in range(buffers)]
            if self.strip_rows > 0:
                self.final = [torch.zeros((m * world * self.rows_per, width, 4), dtype=dtype, device=device) for _ in range(buffers)]
                n, per, sr = world, self.strips_per, self.strip_rows
                self._unshuffle = [(f.view(n, m, per, sr, width, 4).permute(1, 2, 0, 3, 4, 5), g.view(m, per, n, sr, width, 4))
                                   for f, g in zip(self.frames, self.final)]
        elif world > 1 and rank == 0:
            # gathered: [rank, frame of the block, rows of a tile]
            self.frames = [torch.zeros((world * G * self.rows_per, width, 4), dtype=dtype, device=device) for _ in range(buffers)]
            self._glist = [[f[k * G * self.rows_per:(k + 1) * G * self.rows_per] for k in range(world)] for f in self.frames]
            if self.strip_rows > 0:
                # frame order: [frame of the block, strip slot, rank, rows of a strip]
                self.final = [torch.zeros((G * world * self.rows_per, width, 4), dtype=dtype, device=device) for _ in range(buffers)]
                n, per, sr = world, self.strips_per, self.strip_rows
                self._unshuffle = [(f.view(n, G, per, sr, width, 4).permute(1, 2, 0, 3, 4, 5), g.view(G, per, n, sr, width, 4))
                                   for f, g in zip(self.frames, self.final)]

    def root_of(self, g: int) -> int:
        """The rank that assembles frame g of a block."""
        return g // self.m if self.rotate else 0

    def gather(self, b: int, async_op: bool = False):
        """Gather tile buffer `b` of every rank into frame buffer `b` on rank 0 — or, with rotate_roots, exchange the block so
        that every rank receives all tiles of its own m frames (one all-to-all)."""
        import torch.distributed as dist

        if self.world == 1:
            return None
        if self.rotate:
            return dist.all_to_all_single(self.frames[b], self.tiles[b], async_op=async_op)
        return dist.gather(self.tiles[b], self._glist[b] if self.rank == 0 else None, dst=0, async_op=async_op)

    def native_gather(self, renderer, b: int, stream: int = 0) -> None:
        """The same gather through the C-ABI (vrt_gather_tiles: ncclGather of the raw tile bytes), enqueued on HIP stream
        `stream` behind the march that produced tile buffer `b`.  The renderer must have joined a communicator
        (VHipRenderer.comm_init) of `world` ranks."""
        tile = self.tiles[b]
        if self.world == 1:
            return
        if self.rotate:  # vrt_exchange_tiles: chunk d of my tile buffer (my tiles of rank d's frames) goes to rank d
            renderer.exchange_tiles(tile.data_ptr(), self.frames[b].data_ptr(), tile.numel() * tile.element_size() // self.world, stream)
            return
        frame_ptr = self.frames[b].data_ptr() if self.rank == 0 and self.frames is not None else 0
        renderer.gather_tiles(tile.data_ptr(), frame_ptr, tile.numel() * tile.element_size(), 0, stream)

    def unshuffle(self, b: int) -> None:
        """Strips, rank 0: gathered [rank, strip, row] order → frame order, one strided copy on the
        current stream (after the gather of buffer `b` has completed)."""
        if self.strip_rows == 0 or self.world == 1 or (self.rank != 0 and not self.rotate):
            return
        src, dst = self._unshuffle[b]
        dst.copy_(src)

    def tile(self, b: int, g: int = 0):
        """This rank's tile of frame g of buffer (block) b."""
        return self.tiles[b][g * self.rows_per:(g + 1) * self.rows_per]

    def frame(self, b: int, g: int = 0):
        """The assembled H x W x 4 frame g of buffer (block) b (rank 0 only; for strips call unshuffle(b) first)."""
        if self.world == 1:
            return self.tile(b, g)[: self.height]
        full = self.world * self.rows_per
        if self.rotate:  # my j-th frame of the block
            if self.root_of(g) != self.rank:
                return None
            j = g - self.rank * self.m
            if self.strip_rows > 0:
                return self.final[b][j * full: j * full + self.height]
            return self.frames[b].view(self.world, self.m, self.rows_per, self.width, 4)[:, j].reshape(full, self.width, 4)[: self.height]
        if self.rank != 0:
            return None
        if self.strip_rows > 0:
            return self.final[b][g * full: g * full + self.height]
        if self.G == 1:
            return self.frames[b][: self.height]
        # contiguous row tiles of a block: rank-major in the gathered buffer, so frame g is a strided view
        return self.frames[b].view(self.world, self.G, self.rows_per, self.width, 4)[:, g].reshape(full, self.width, 4)[: self.height]
