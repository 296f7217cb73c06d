import sys, time, ctypes as C
sys.path.insert(0,'/root/repo')
import torch, numpy as np
import bench
import volumetricraytracer_amd as v
from volumetricraytracer_amd import _abi, workloads
sc,W,H,ms,sh,label=bench.build_workload('c3')
p=v.default_params(W,H,workloads.min_cell(sc),ms,shadow=sh)
p.flags|=_abi.FLAG_OUTPUT_RGBA8
r=v.VHipRenderer(); assert r.Start(); r.SetSceneToRender(sc); r.ResizeRenderOutput(W,H); r.SyncWithScene()
lib=r._lib
def loop(fn,n=200):
    fn(); fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
print('SyncWithScene alone ms', loop(r.SyncWithScene))
state={'i':0}
def raw():
    i=state['i']; state['i']+=1
    _abi.check(lib.vrt_render_begin(r._ctx, C.byref(p), i%2),'b')
    if i>0:
        ptr=C.c_void_p(); _abi.check(lib.vrt_render_end(r._ctx,(i-1)%2,C.byref(ptr)),'e')
def raw_wrap():
    raw()
t=loop(raw_wrap,300)
ptr=C.c_void_p(); lib.vrt_render_end(r._ctx,(state['i']-1)%2,C.byref(ptr))
print('raw begin/end pipelined (no sync) ms/frame', t)
def full():
    i=state['i']; state['i']+=1
    r.render_begin(i%2,p)
    if i>0: r.render_end((i-1)%2,p,copy=False)
state['i']=0
t=loop(full,300); r.render_end((state['i']-1)%2,p,copy=False)
print('python render_begin/end (with SyncWithScene) ms/frame', t)
# copy alone
fb=torch.empty((H,W,4),dtype=torch.uint8,device='cuda:0'); host=torch.empty((H,W,4),dtype=torch.uint8).pin_memory()
def cp(): host.copy_(fb,non_blocking=True)
print('D2H 8.3MB pinned ms', loop(cp,200))
out=torch.empty((H,W,4),dtype=torch.uint8,device='cuda:0')
def k(): r.render_rows(p,0,H,out.data_ptr(),0)
print('kernel only (back to back, one stream) ms', loop(k,200))
def ks(): r.render_rows(p,0,H,out.data_ptr(),0); torch.cuda.synchronize()
print('kernel + sync ms', loop(ks,200))
r.Stop()
