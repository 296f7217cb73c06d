"""Shared by the CPU and GPU tests that compare a frame with tests/golden/ref_*.npz — frames as the REFERENCE's own
intersection (exact per-cell cubic root, normal at the root; tests/golden/make_ref_golden.py) would shade them, stored as
the 8-bit colours its render target holds (B8G8R8A8_UNORM, DXConstants.cpp:21).

Two restatements of the reference's intersection exist (oracle/vrt_oracle.h) and each has its fixtures:
  ref_<case>.npz          IDEALISED: double precision, exact first root, no nudges, no octree, no budget, normalised camera direction
  ref_literal_<case>.npz  LITERAL: the shaders statement by statement in fp32 — un-normalised camera direction (offsets and the
                          shading's wo in its units), +0.01 / +0.1 nudges, collapsed-octree leaves, 2 regula-falsi + 1 secant, abs()-
                          weighted normal with out-of-bounds texels 0, 255 leaves then the red hit (`rgb8`, `t`), and the same shaders
                          fed the NORMALISED camera direction (`rgb8_norm`, `t_norm`)
`compare(img, name, against=...)` measures a frame against any of the three."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    w, h, row0, rows = (int(x) for x in z["window"])
    return z["rgb8"], z["t"], (w, h, row0, rows)


def literal_name(name):
    return "ref_literal_" + name[len("ref_"):]


LITERAL_STATS = ("rays", "iterations", "solid_start_hits", "entry_hits", "root_hits", "tail_hits", "red_hits", "rejected_reports")


def load_literal(name, normalised=False):
    """The literal restatement's frame of case `name` (the idealised fixture's name): (rgb8, t in world units, window, stats)."""
    z = np.load(os.path.join(GOLDEN, literal_name(name) + ".npz"))
    w, h, row0, rows = (int(x) for x in z["window"])
    sfx = "_norm" if normalised else ""
    return z["rgb8" + sfx], z["t" + sfx], (w, h, row0, rows), dict(zip(LITERAL_STATS, (int(x) for x in z["stats"])))


def load_against(name, against):
    if against == "idealised":
        return load(name)
    if against in ("literal", "literal_norm"):
        return load_literal(name, against == "literal_norm")[:3]
    raise ValueError(against)


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


def compare_rgb8(a8, rgb8, t, height, erosion=2):
    """Two 8-bit frames of one window; `t`, `height`: hit distances and frame height that define the interior (of the second)."""
    assert a8.shape[:2] == rgb8.shape[:2], (a8.shape, rgb8.shape)
    d = np.abs(a8[..., :3].astype(np.int16) - rgb8[..., :3].astype(np.int16)).max(-1)
    inside = erode(smooth_hits(t, height), erosion)
    di = d[inside]
    return {"interior_pixels": int(inside.sum()), "gt0": float((di > 0).mean()), "gt1": float((di > 1).mean()),
            "gt2": float((di > 2).mean()), "max": int(di.max()), "mean": float(di.mean()),
            "frame_gt1": float((d > 1).mean()), "frame_gt2": float((d > 2).mean())}


def compare(img, name, erosion=2, against="idealised"):
    """img: float frame (rows, W, >= 3 channels) of the fixture's window.  Returns the fixture-relative numbers DESIGN.md §5
    quotes: over the INTERIOR of the reference's surfaces (smooth_hits eroded by `erosion` pixels) the fraction of pixels
    whose 8-bit colour differs from the reference's by more than 0, 1 and 2 steps in any channel, the largest difference,
    the mean; and over the whole window the 'more than 1 / 2 steps' fractions (silhouette pixels included).
    against: "idealised" (ref_<case>.npz), "literal" or "literal_norm" (ref_literal_<case>.npz)."""
    rgb8, t, (_, height, _, _) = load_against(name, against)
    return compare_rgb8(quantise(img), rgb8, t, height, erosion)


def fixture_pairs(name, erosion=2):
    """What the two restatements of the reference say about each other, from the committed files alone (interior of the IDEALISED
    frame's surfaces): literal vs idealised, literal fed the normalised camera direction vs idealised, and the two literal frames."""
    ide, t, (_, height, _, _) = load(name)
    lit, tl, _, stats = load_literal(name)
    litn, tn, _, _ = load_literal(name, True)
    both = (t > 0) & (tl > 0)
    return {"literal_vs_idealised": compare_rgb8(lit, ide, t, height, erosion),
            "literal_norm_vs_idealised": compare_rgb8(litn, ide, t, height, erosion),
            "literal_vs_literal_norm": compare_rgb8(lit, litn, t, height, erosion),
            "hit_mask_differs": int(((t > 0) != (tl > 0)).sum()),
            "t_abs_diff_median": float(np.median(np.abs(t - tl)[both])) if both.any() else 0.0,
            "t_abs_diff_p99": float(np.percentile(np.abs(t - tl)[both], 99)) if both.any() else 0.0,
            "stats": stats}
