import sys, time, ctypes as C
sys.path.insert(0, '.')
import torch
import bench
import volumetricraytracer_amd as v
from volumetricraytracer_amd import _abi, workloads
sc, W, H, ms, sh, label = bench.build_workload('c3')
r = v.VHipRenderer(); assert r.Start(); r.SetSceneToRender(sc); r.ResizeRenderOutput(W, H); r.SyncWithScene()
p = v.default_params(W, H, workloads.min_cell(sc), ms, shadow=sh)
p8 = _abi.vrt_params.from_buffer_copy(p); p8.flags |= _abi.FLAG_OUTPUT_RGBA8
for n in (50, 50, 300):
    print(n, 'float', bench.end_to_end_leg(r, p, 2.4e6, n)['ms_per_frame'], 'rgba8', bench.end_to_end_leg(r, p8, 2.4e6, n)['ms_per_frame'], flush=True)
# after a 96-frame block on the null stream (what the bench's main line does)
cams = r.camera_array(workloads.orbit_cameras(sc, 96))
buf = torch.empty((96, H, W, 4), dtype=torch.float32, device='cuda')
for _ in range(3):
    r.render_block(p, 96, buf.data_ptr(), H * W * 16, 0, cameras=(cams, 0))
torch.cuda.synchronize()
for n in (50, 300):
    print('after blocks', n, 'float', bench.end_to_end_leg(r, p, 2.4e6, n)['ms_per_frame'], 'rgba8', bench.end_to_end_leg(r, p8, 2.4e6, n)['ms_per_frame'], flush=True)
del buf
torch.cuda.empty_cache()
for n in (50, 300):
    print('after freeing the 3.2 GB block buffer', n, 'float', bench.end_to_end_leg(r, p, 2.4e6, n)['ms_per_frame'], 'rgba8', bench.end_to_end_leg(r, p8, 2.4e6, n)['ms_per_frame'], flush=True)
r.Stop()
