"""GPU box: the full closest-hit kernel (lights, bounces, textures) against the lean kernel on the same scene, 1080p."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import volumetricraytracer_amd as v
from volumetricraytracer_amd import workloads, _abi
import copy
_c3 = copy.copy(workloads.bench_config3())
_c3.PointLights = [v.VPointLight(Position=(120.0, 60.0, 140.0), Color=(1.0, 0.9, 0.8), IlluminationStrength=40.0)]
for name, sc, bounces in (("config 3 volume, Cube mode (exact voxel walk, lean kernel)", workloads.bench_config3(), 0), ("config 3 volume + one point light (single instance, full kernel)", _c3, 0), ("full_closest_hit (lights+mirror bounce)", workloads.full_closest_hit_scene(7, 64), 2), ("textured", workloads.textured_scene(7, 64), 2), ("same scene, lean kernel (no lights/bounces)", None, 0)):
    if len(sys.argv) > 1 and sys.argv[1] not in name:
        continue
    if sc is None:
        sc = workloads.full_closest_hit_scene(7, 64); sc.PointLights=[]; sc.SpotLights=[]
    W,H=1920,1080
    mode=_abi.MODE_INTERP if name=="textured" else (_abi.MODE_CUBE_NOTEX if "Cube mode" in name else _abi.MODE_INTERP_NOTEX)
    p=v.default_params(W,H,workloads.min_cell(sc),255,shadow=True,mode=mode); p.max_bounces=bounces
    r=v.VHipRenderer(); assert r.Start(); r.SetSceneToRender(sc); r.ResizeRenderOutput(W,H); r.SyncWithScene()
    G=_abi.MAX_BLOCK_FRAMES  # ONE launch per block of 48 frames, one stream
    buf=torch.empty((G,H,W,4),dtype=torch.float32,device="cuda:0")
    def run(n, q):
        for i in range(n):
            r.render_block(q,G,buf.data_ptr(),H*W*16,0)
        torch.cuda.synchronize()
    for form, flag in (("three passes", 0), ("one kernel", _abi.FLAG_FULL_ONE_KERNEL))[:1 if len(sys.argv) > 2 else 2]:
        q=_abi.vrt_params.from_buffer_copy(p); q.flags |= flag
        run(6, q)  # the GPU's clocks take tens of milliseconds to ramp: untimed
        t0=time.perf_counter(); run(12, q); dt=(time.perf_counter()-t0)/(12*G)
        t=r.last_timing()
        rays=t["primary_rays"]+t["shadow_rays"]+t["bounce_rays"]
        print(f"{name} [{form}]: {dt*1e3:.4f} ms/frame, {rays/dt/1e9:.2f} Grays/s, rays/frame {rays}, samples/ray {(t['primary_steps']+t['shadow_steps'])/rays:.2f}, hits {t['hits']}")
    if "full kernel" in name or "lights" in name:  # a lone frame, waited for: which form should a single-frame launch take?
        for form, flag in (("passes", _abi.FLAG_FULL_THREE_PASS), ("one kernel", 0)):
            q=_abi.vrt_params.from_buffer_copy(p); q.flags |= flag
            def lone(n):
                for i in range(n):
                    r.render_block(q,1,buf.data_ptr(),H*W*16,0); torch.cuda.synchronize()
            lone(50)
            t0=time.perf_counter(); lone(200); dt=(time.perf_counter()-t0)/200
            print(f"    lone frame, host waits for each [{form}]: {dt*1e3:.4f} ms/frame")
    r.Stop()
