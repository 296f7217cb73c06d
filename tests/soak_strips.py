"""Not collected by pytest (by hand on the GPU box: `python tests/soak_strips.py 0 150`): random frame sizes, strip heights,
rank counts and pixel formats — every rank's compact tile of interleaved strips (vrt_render_strips, and a block of frames
through vrt_render_block), put back in frame order the way rank 0 does (FrameGather), must be the whole frame rendered in
one launch, bit for bit; rows beyond the frame must stay untouched."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import volumetricraytracer_amd as v  # noqa: E402
from volumetricraytracer_amd import _abi, workloads  # noqa: E402
from volumetricraytracer_amd.tiles import FrameGather  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
sc = workloads.config5_instances(5, 16)
r = v.VHipRenderer()
assert r.Start()
r.SetSceneToRender(sc)
r.SyncWithScene()
bad = []
for seed in range(lo, hi):
    rng = np.random.RandomState(seed)
    W, H = int(rng.randint(1, 300)), int(rng.randint(1, 300))
    world = int(rng.randint(1, 9))
    sr = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 32, 40]))
    rgba8 = bool(rng.randint(0, 2))
    G = int(rng.randint(1, 4))
    p = v.default_params(W, H, workloads.min_cell(sc), 255, shadow=True)
    if rgba8:
        p.flags |= _abi.FLAG_OUTPUT_RGBA8
    dt = torch.uint8 if rgba8 else torch.float32
    whole = torch.zeros((H, W, 4), dtype=dt, device="cuda:0")
    r.ResizeRenderOutput(W, H)
    r.render_rows(p, 0, H, whole.data_ptr(), 0)
    fg = FrameGather(H, W, world, 0, torch.device("cuda:0"), dtype=dt, buffers=1, strip_rows=sr, frames_per_gather=G)
    rows_per = fg.rows_per
    fb = rows_per * W * (4 if rgba8 else 16)
    ok = True
    if world > 1:
        fg.frames[0].fill_(77)
        for rank in range(world):
            blk = fg.frames[0][rank * G * rows_per:(rank + 1) * G * rows_per]  # where rank's block lands after the gather
            if rank % 2 == 0:
                r.render_block(p, G, blk.data_ptr(), fb, 0, strips=(sr, rank, world, fg.strips_per))
            else:
                for g in range(G):
                    r.render_strips(p, sr, rank, world, fg.strips_per, blk[g * rows_per:(g + 1) * rows_per].data_ptr(), 0)
        torch.cuda.synchronize()
        fg.unshuffle(0)
        for g in range(G):
            ok = ok and bool(torch.equal(fg.frame(0, g), whole))
        # padding rows (strip slots beyond the frame) keep the fill value
        pad = fg.final[0].view(G, world * rows_per, W, 4)[:, H:]
        ok = ok and (pad.numel() == 0 or bool((pad == 77).all()))
    else:
        r.render_block(p, G, fg.tiles[0].data_ptr(), fb, 0, strips=(sr, 0, 1, fg.strips_per))
        torch.cuda.synchronize()
        for g in range(G):
            ok = ok and bool(torch.equal(fg.frame(0, g), whole))
    if not ok:
        bad.append(seed)
        print(f"seed {seed}: {W}x{H} world {world} strip rows {sr} rgba8 {rgba8} frames {G}: MISMATCH", flush=True)
    if seed % 25 == 0:
        print(f"... seed {seed}", flush=True)
r.Stop()
print(f"seeds {lo}..{hi - 1}: {len(bad)} mismatches {bad}")
sys.exit(1 if bad else 0)
