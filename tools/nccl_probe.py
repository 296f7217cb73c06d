#!/usr/bin/env python3
"""One-rank RCCL smoke check on the GPU box: process group over backend "nccl", an asynchronous uint8 gather issued under a
side stream (what bench.py does with the RGBA8 tiles), a float64 all_reduce and a barrier."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.arange(16, dtype=torch.uint8, device=dev).reshape(4, 4)
out = [torch.zeros_like(t)]
s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s):
    w = dist.gather(t, out, dst=0, async_op=True)
    w.wait()
torch.cuda.synchronize()
te = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(te, op=dist.ReduceOp.MAX)
dist.barrier()
print("nccl ok:", torch.equal(out[0], t), float(te.item()), torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
dist.destroy_process_group()
