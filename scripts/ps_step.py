"""Mellinger pipeline step time (inputs resident, one batch in flight) for a batch shape; MRS_TG_PS=0|1 selects the outer loop.
usage: ps_step.py n_paths n_seg|ragged [derivative_to_optimize] [box|walk]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mrs_uav_trajectory_generation_amd import api, problem as pr

P = int(sys.argv[1])
n_seg = sys.argv[2] if len(sys.argv) > 2 else "10"
n_seg = n_seg if n_seg == "ragged" else int(n_seg)
deriv = int(sys.argv[3]) if len(sys.argv) > 3 else 4
gen = sys.argv[4] if len(sys.argv) > 4 else "box"
ctx = api.Context(0)
ctx.use_torch_stream()
batch = pr.random_batch(P, n_seg, seed0=0, derivative_to_optimize=deriv, generator=gen)
plan = api.Plan(ctx, batch.seg_offsets)
db = api.DeviceBatch(batch, "cuda:0", sample_capacity=512)
est = api.default_options(estimate_times=1, derivative_to_optimize=deriv)
plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints, limits=db.limits)
torch.cuda.synchronize()
t0 = db.seg_times.clone()
opt = api.default_options(time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=512, derivative_to_optimize=deriv)


def step():
    db.seg_times.copy_(t0)
    plan.solve(opt, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, limits=db.limits,
               n_samples=db.n_samples, samples=db.samples)


for _ in range(5):
    step()
torch.cuda.synchronize()
reps = 30 if P <= 8192 else 8
s = time.perf_counter()
for _ in range(reps):
    step()
torch.cuda.synchronize()
el = (time.perf_counter() - s) / reps
ctx.set_profiling(True)
step()
ms = ctx.last_kernel_ms(api.KERNEL_NONLINEAR)
ctx.set_profiling(False)
st = db.status.cpu().numpy()
print("PS=%s  %d x %s: %.1f us per step, outer-loop kernel (first launch) %.1f us, status histogram %s, times checksum %.12e" %
      (os.environ.get("MRS_TG_PS", "default"), P, n_seg, el * 1e6, ms * 1e3, dict(zip(*np.unique(st, return_counts=True))),
       float(db.seg_times.sum())))
