"""Device time of the Mellinger pipeline with and without MRS_TG_FLAG_CAREFUL_COST (inputs resident, one batch in flight)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mrs_uav_trajectory_generation_amd import api, problem as pr

ctx = api.Context(0)
ctx.use_torch_stream()
for P in (1024, 8192, 65536):
    batch = pr.random_batch(P, 10, seed0=0)
    plan = api.Plan(ctx, batch.seg_offsets)
    db = api.DeviceBatch(batch, "cuda:0", sample_capacity=512)
    est = api.default_options(estimate_times=1)
    plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints, limits=db.limits)
    torch.cuda.synchronize()
    t0 = db.seg_times.clone()
    line = "%6d paths:" % P
    for name, fl in (("fast", 0), ("careful", api.FLAG_CAREFUL_COST)):
        opt = api.default_options(time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=512, flags=fl)

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
        line += "  %s %.1f us" % (name, (time.perf_counter() - s) / reps * 1e6)
    print(line + "  (%d paths re-run)" % plan.careful_count(), flush=True)
    plan.close()
