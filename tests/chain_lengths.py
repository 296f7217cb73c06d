"""Not a test: a study helper run by hand (`python tests/chain_lengths.py [c3|c3sdf|c2|c5] [k_relax ...]`).

What bounds the GPU frame is the length of the dependent position chain a wave runs: a wave is an
8x8 pixel tile whose 64 lanes step together, so its march takes max-over-lanes positions for the primary rays plus
max-over-lanes positions for the rays that follow.  The oracle visits exactly the positions the kernel visits (parity
tests), so the chain lengths of a frame can be read off the oracle's debug position image without a GPU — the quickest
way to judge a change of the stepping rule (over-relaxation factor, tables) before it is written into the kernel.

Lives under tests/ because it drives the oracle, which only test infrastructure may do.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
import volumetricraytracer_amd as v  # noqa: E402
from oracle import binding  # noqa: E402
from oracle.binding import OracleScene  # noqa: E402
from volumetricraytracer_amd import workloads  # noqa: E402


def chain_stats(sc, p, threads: int = 8) -> dict:
    lib = binding.load()
    lib.vrto_debug_set_steps_image.argtypes = [C.c_void_p]
    W, H = p.width, p.height
    img = np.zeros((H, W), np.uint32)
    lib.vrto_debug_set_steps_image(img.ctypes.data)
    try:
        _, st = OracleScene(sc).render(p, threads=threads)
    finally:
        lib.vrto_debug_set_steps_image(None)
    first, rest = (img & 0xFFFF).astype(np.int64), (img >> 16).astype(np.int64)

    def tiles(a):
        Hp, Wp = (H + 7) // 8 * 8, (W + 7) // 8 * 8
        b = np.zeros((Hp, Wp), a.dtype)
        b[:H, :W] = a
        return b.reshape(Hp // 8, 8, Wp // 8, 8).transpose(0, 2, 1, 3).reshape(Hp // 8, Wp // 8, 64)

    wave = tiles(first).max(-1) + tiles(rest).max(-1)
    rays = st["primary_rays"] + st["shadow_rays"] + st["bounce_rays"]
    return {
        "samples_per_ray": (st["primary_steps"] + st["shadow_steps"]) / max(rays, 1),
        "positions_per_pixel": float((first + rest).mean()),
        "lane_chain_max": int((first + rest).max()),
        "wave_chain_max": int(wave.max()),
        "wave_chain_p99": float(np.percentile(wave, 99)),
        "wave_iterations": int(wave.sum()),
        "hits": st["hits"],
        "exhausted_rays": st["exhausted_rays"],
    }


def main(argv):
    wl = argv[1] if len(argv) > 1 else "c3"
    ks = [float(a) for a in argv[2:]] or [1.0, 1.7]
    sc, W, H, ms, sh, label = bench.build_workload(wl)
    print(label)
    for k in ks:
        p = v.default_params(W, H, workloads.min_cell(sc), ms, shadow=sh, k_relax=k)
        s = chain_stats(sc, p)
        print(f"k_relax {k:4.2f}: " + "  ".join(f"{a} {b:.2f}" if isinstance(b, float) else f"{a} {b}" for a, b in s.items()))


if __name__ == "__main__":
    main(sys.argv)
