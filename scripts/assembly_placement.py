"""The 1.05 GB assembly launch (65536 x 10 segments) on output buffers allocated at different points of a process: every pair is
re-measured after every allocation / release.  The rate belongs to the buffer (profiles/round6_assembly_placement.txt).
  python scripts/assembly_placement.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mrs_uav_trajectory_generation_amd import api
dev = torch.device("cuda:0")
ctx = api.Context(0)
ctx.use_torch_stream()
P, S = 65536, 10
so = (np.arange(P + 1, dtype=np.int64) * S).astype(np.int32)
plan = api.Plan(ctx, so)
t = torch.from_numpy(np.random.default_rng(0).uniform(0.5, 3.0, P * S)).to(dev)
nd = plan.block_doubles
bytes_big = 1608 * P * S
pairs = []
def frac(Hb, Ab):
    fn = lambda: plan.assemble(4, t, Hb, Ab)
    m, med, mn = bench.dispatch_stats(ctx, api.KERNEL_ASSEMBLE, fn, 20, torch)
    return bytes_big / (med * 1e-3) / 8e12
def show(tag):
    free, total = torch.cuda.mem_get_info()
    print("%-34s used %5.1f GB | " % (tag, (total - free) / 2**30) + "  ".join("%d:%.2f" % (k, frac(*p)) for k, p in enumerate(pairs)), flush=True)
for k in range(6):
    pairs.append((torch.empty(nd, dtype=torch.float64, device=dev), torch.empty(nd, dtype=torch.float64, device=dev)))
    show("after allocating pair %d" % k)
show("again")
x = torch.empty(64 * 1024 * 1024, dtype=torch.float64, device=dev)
show("after torch.empty(512 MB)")
x.zero_()
torch.cuda.synchronize()
show("after zeroing it")
del x
torch.cuda.empty_cache()
show("after freeing it")
pairs = pairs[:2]
torch.cuda.empty_cache()
show("after freeing pairs 2..5")
pairs.append((torch.empty(nd, dtype=torch.float64, device=dev), torch.empty(nd, dtype=torch.float64, device=dev)))
show("after allocating a new pair")
