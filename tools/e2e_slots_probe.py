"""How vrt_render_begin / _end pipelines by number of frame slots, before and after other streams exist in the process."""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import torch
import bench
import volumetricraytracer_amd as v
from volumetricraytracer_amd import _abi, workloads

sc, W, H, ms, sh, label = bench.build_workload('c3')
r = v.VHipRenderer(); assert r.Start(); r.SetSceneToRender(sc); r.ResizeRenderOutput(W, H); r.SyncWithScene()
lib = r._lib
def run(p, slots, n=300):
    for i in range(slots): _abi.check(lib.vrt_render_begin(r._ctx, C.byref(p), i), 'b')
    ptr = C.c_void_p()
    for i in range(slots): _abi.check(lib.vrt_render_end(r._ctx, i, C.byref(ptr)), 'e')
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        if i >= slots: _abi.check(lib.vrt_render_end(r._ctx, i % slots, C.byref(ptr)), 'e')
        _abi.check(lib.vrt_render_begin(r._ctx, C.byref(p), i % slots), 'b')
    for i in range(max(n - slots, 0), n): _abi.check(lib.vrt_render_end(r._ctx, i % slots, C.byref(ptr)), 'e')
    return (time.perf_counter() - t0) / n * 1e3
for tag in ("fresh process", "after 6 torch streams"):
    for fmt, flags in (("rgba8", _abi.FLAG_OUTPUT_RGBA8), ("float", 0)):
        p = v.default_params(W, H, workloads.min_cell(sc), ms, shadow=sh); p.flags |= flags
        print(tag, fmt, {s: round(run(p, s), 4) for s in (1, 2, 3)}, flush=True)
    keep = [torch.cuda.Stream() for _ in range(6)]
    x = torch.zeros(1 << 20, device='cuda')
    for st in keep:
        with torch.cuda.stream(st): x += 1
    torch.cuda.synchronize()
r.Stop()
