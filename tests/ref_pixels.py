"""Shared by the CPU and GPU tests that compare a frame with tests/golden/ref_*.npz — frames as the REFERENCE's own
intersection (exact per-cell cubic root, normal at the root; tests/golden/make_ref_golden.py) would shade them, stored as
the 8-bit colours its render target holds (B8G8R8A8_UNORM, DXConstants.cpp:21)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    w, h, row0, rows = (int(x) for x in z["window"])
    return z["rgb8"], z["t"], (w, h, row0, rows)


def quantise(img):
    return np.floor(np.minimum(np.asarray(img, np.float32)[..., :3], 1.0) * 255.0 + 0.5).astype(np.uint8)


def erode(mask, n):
    """Binary erosion by n pixels (4-neighbourhood, n times); pixels outside the array count as unset."""
    m = mask.copy()
    for _ in range(n):
        p = np.pad(m, 1, constant_values=False)
        m = p[1:-1, 1:-1] & p[:-2, 1:-1] & p[2:, 1:-1] & p[1:-1, :-2] & p[1:-1, 2:]
    return m


def smooth_hits(t, height, slope=6.0, fov_deg=60.0):
    """Pixels of the reference frame that hit and whose hit distance differs from each 4-neighbour's by less than `slope` pixel
    footprints (a surface seen at up to ~80 degrees of incidence): not on a silhouette — the outer one, or an inner one where a
    near surface ends in front of a far one (a pixel there holds whichever surface its centre ray meets; a cone-terminated
    sphere-trace and an exact root may disagree by one pixel about where that edge is, test_hit_mask_is_the_reference_surface…)."""
    hit = t > 0
    foot = 2.0 * np.tan(np.radians(fov_deg) * 0.5) / float(height)
    lim = slope * foot * np.maximum(t, 0.0)
    ok = hit.copy()
    p = np.pad(t, 1, mode="edge")
    for nb in (p[:-2, 1:-1], p[2:, 1:-1], p[1:-1, :-2], p[1:-1, 2:]):
        ok &= (nb > 0) & (np.abs(nb - t) < lim)
    return ok


def compare(img, name, erosion=2):
    """img: float frame (rows, W, >= 3 channels) of the fixture's window.  Returns the fixture-relative numbers DESIGN.md §5
    quotes: over the INTERIOR of the reference's surfaces (smooth_hits eroded by `erosion` pixels) the fraction of pixels
    whose 8-bit colour differs from the reference's by more than 0, 1 and 2 steps in any channel, the largest difference,
    the mean; and over the whole window the same 'more than 1 step' fraction (silhouette pixels included)."""
    rgb8, t, (_, height, _, _) = load(name)
    assert img.shape[:2] == rgb8.shape[:2], (img.shape, rgb8.shape)
    d = np.abs(quantise(img).astype(np.int16) - rgb8.astype(np.int16)).max(-1)
    inside = erode(smooth_hits(t, height), erosion)
    di = d[inside]
    return {"interior_pixels": int(inside.sum()), "gt0": float((di > 0).mean()), "gt1": float((di > 1).mean()),
            "gt2": float((di > 2).mean()), "max": int(di.max()), "mean": float(di.mean()),
            "frame_gt1": float((d > 1).mean())}
