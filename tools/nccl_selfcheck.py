#!/usr/bin/env python3
"""One-GPU sanity check of the RCCL calls bench.py makes for N > 1 (gather of a band on a side
stream, all_reduce of the timing vector, barrier) with a process group of one rank: it proves the
backend, stream and tensor usage on the box; the N > 1 data path itself is covered by the gloo tests
and by `bench.py --backend gloo`."""
import os, torch, torch.distributed as dist
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29555", rank=0, world_size=1, device_id=dev)
band = torch.arange(3840 * 100, dtype=torch.int32, device=dev).reshape(100, 3840)
parts = [torch.empty_like(band)]
out = torch.zeros_like(band)
s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s):
    dist.gather(band, parts, dst=0)
    out.copy_(parts[0])
torch.cuda.synchronize()
assert torch.equal(out, band)
t = torch.tensor([1.0, 2.0], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.all_reduce(t, op=dist.ReduceOp.SUM); dist.barrier()
torch.cuda.synchronize()
print("nccl world-1 gather/all_reduce/barrier ok", t.tolist())
dist.destroy_process_group()
