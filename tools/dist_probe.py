"""Diagnostic: host-side cost of the c10d calls the data-parallel step makes, on ONE GPU (RCCL world of one rank): what does an async
all_reduce + wait cost the issuing thread, and does it block?   python tools/dist_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

os.environ.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": os.environ.get("MASTER_PORT", "29533")})
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.zeros(33_000_000, device="cuda")
busy = torch.zeros(64 << 20, device="cuda")
for n_mb in (32, 8):
    n = n_mb * (1 << 20) // 4
    for rep in range(3):
        torch.cuda.synchronize()
        # a long-running kernel sequence in front, as the backward is in front of a staged reduction
        for _ in range(20):
            busy.add_(1.0)
        t0 = time.perf_counter()
        works = [dist.all_reduce(x[i * n:(i + 1) * n], op=dist.ReduceOp.SUM, async_op=True) for i in range(4)]
        t1 = time.perf_counter()
        for w in works:
            w.wait()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print("4 x %d MB: issue %.3f ms, wait() %.3f ms, device drain %.3f ms" % (n_mb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
for rep in range(4):
    for _ in range(20):
        busy.add_(1.0)
    t0 = time.perf_counter()
    dist.barrier()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("dist.barrier(): %.3f ms (+ synchronize %.3f ms)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
one = torch.zeros(1, device="cuda")
for rep in range(4):
    for _ in range(20):
        busy.add_(1.0)
    t0 = time.perf_counter()
    dist.all_reduce(one)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("all_reduce(1 element) + synchronize: %.3f ms" % ((t1 - t0) * 1e3))
dist.destroy_process_group()
