"""A/B helper (by hand on the GPU box): renders one frame of several workloads with the library VRT_LIB names (default: the
product library) and saves pixels + counters to an .npz, so that two builds of the kernel can be compared bit for bit:

    VRT_LIB=volumetricraytracer_amd/lib/libvrt_hip_variant.so python tools/ab_frames.py gpurun_out/ab_variant.npz
    python tools/ab_frames.py gpurun_out/ab_product.npz && python tools/ab_frames.py --compare gpurun_out/ab_product.npz gpurun_out/ab_variant.npz
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        if k.endswith("_timing"):
            print(k, str(a[k]), "|", str(b[k]))
        else:
            d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
            print(f"{k}: identical {np.array_equal(a[k], b[k])}  max|diff| {d.max():.3g}  pixels that differ {(d.max(-1) > 0).sum()}")
    sys.exit(0)

import bench  # noqa: E402
import volumetricraytracer_amd as v  # noqa: E402
from volumetricraytracer_amd import workloads  # noqa: E402

res = {}
for wl in ("c3", "c5", "c3cover", "c2"):
    for k_relax in (1.7, 1.0):
        sc, W, H, ms, sh, label = bench.build_workload(wl)
        p = v.default_params(W, H, workloads.min_cell(sc), ms, shadow=sh, k_relax=k_relax)
        r = v.VHipRenderer()
        assert r.Start()
        r.SetSceneToRender(sc)
        r.ResizeRenderOutput(W, H)
        r.render_begin(0, p)
        img = r.render_end(0, p)
        tm = r.last_timing()
        res[f"{wl}_k{k_relax}"] = img
        res[f"{wl}_k{k_relax}_timing"] = np.array(json.dumps({k: tm[k] for k in tm if "steps" in k or "rays" in k or "hits" in k}))
        r.Stop()
np.savez_compressed(sys.argv[1], **res)
print("saved", sys.argv[1])
