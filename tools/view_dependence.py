"""By hand on the GPU box: config 3's frame from cameras at several azimuths about the torus's axis of symmetry (the SAME picture up to the
light; without shadow rays the same work exactly) — does the march's speed depend on how the view lies to the brick records' memory
order (a 128-byte line holds the 4x4 cells of one x-slab of a brick)?   python tools/view_dependence.py [path ...] > gpurun_out/r04/view_dependence.txt"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402
import volumetricraytracer_amd as v  # noqa: E402
from volumetricraytracer_amd import _abi, workloads  # noqa: E402

paths = sys.argv[1:] or ["auto"]
sc, W, H, ms, sh, label = bench.build_workload("c3")
print(label)
B, reps = 96, 6
cam0 = sc.Camera
for path in paths:
    for shadow in (False, True):
        r = v.VHipRenderer()
        assert r.Start()
        r.SetSceneToRender(sc)
        r.ResizeRenderOutput(W, H)
        r.SyncWithScene()
        p = v.default_params(W, H, workloads.min_cell(sc), ms, shadow=shadow,
                             path={"auto": _abi.PATH_AUTO, "dense": _abi.PATH_DENSE, "brick": _abi.PATH_BRICK}[path])
        out = torch.empty((B, H, W, 4), dtype=torch.float32, device="cuda")
        for az in (0, 15, 30, 45, 60, 75, 90, 135, 180, 270):
            cams = []
            for f in range(B):
                a = math.radians(az + (f - B // 2) * 0.25)
                c, s_ = math.cos(a), math.sin(a)
                pos = (cam0.Position[0] * c - cam0.Position[1] * s_, cam0.Position[0] * s_ + cam0.Position[1] * c, cam0.Position[2])
                cams.append((pos, tuple(v.quat_mul(v.quat_from_axis_angle(v.UP, a), cam0.Rotation)), float(cam0.FOVAngle)))
            arr = r.camera_array(cams)
            st = torch.cuda.current_stream().cuda_stream
            r.render_block(p, B, out.data_ptr(), H * W * 16, st, cameras=(arr, 0))
            torch.cuda.synchronize()
            r.render_block(p, B, out.data_ptr(), H * W * 16, st, cameras=(arr, 0))  # (warm)
            torch.cuda.synchronize()
            tm = r.last_timing()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                r.render_block(p, B, out.data_ptr(), H * W * 16, st, cameras=(arr, 0))
            e1.record()
            torch.cuda.synchronize()
            msf = e0.elapsed_time(e1) / (reps * B)
            rays = tm["primary_rays"] + tm["shadow_rays"]  # (the counters are those of the launch's timing sample: one frame)
            steps = tm.get("primary_steps", 0) + tm.get("shadow_steps", 0)
            print(f"path {path:5s} shadow {int(shadow)}  azimuth {az:3d}: {msf * 1e3:7.2f} us/frame  {rays / msf / 1e3:9.1f} Mrays/s  samples/frame {steps / 1e6:.2f} M  hits {tm.get('hits', 0)}")
        r.Stop()
