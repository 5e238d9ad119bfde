"""Nonlinear pipeline step of 1024 x 10 rest-to-rest paths for the three objectives (d = 2 is the reference's shipping
default: the end vertices then keep jerk and snap free)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mrs_uav_trajectory_generation_amd import api, problem as pr

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = api.Context(0)
ctx.use_torch_stream()
for d in (4, 3, 2):
    for gen in ("box", "walk"):
        batch = pr.random_batch(P, 10, seed0=0, derivative_to_optimize=d, generator=gen)
        plan = api.Plan(ctx, batch.seg_offsets)
        db = api.DeviceBatch(batch, "cuda:0", sample_capacity=512)
        est = api.default_options(derivative_to_optimize=d, estimate_times=1)
        plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints, limits=db.limits)
        torch.cuda.synchronize()
        t0 = db.seg_times.clone()
        opt = api.default_options(derivative_to_optimize=d, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=512)

        def step():
            db.seg_times.copy_(t0)
            plan.solve(opt, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, limits=db.limits,
                       n_samples=db.n_samples, samples=db.samples)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(40):
            step()
        torch.cuda.synchronize()
        print("d=%d %s: %.1f us per %d-path nonlinear step" % (d, gen, (time.perf_counter() - t) / 40 * 1e6, P))
        plan.close()
