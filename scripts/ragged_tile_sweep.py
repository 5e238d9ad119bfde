"""Ragged batches (3..30 segments): tile kernel vs lane kernels, linear (blocks) and nonlinear (fused) pipelines.
usage: ragged_tile_sweep.py P...   (sets MRS_TG_TILE_MAX_PATHS itself; the library reads it at every call)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mrs_uav_trajectory_generation_amd import api, problem as pr

ctx = api.Context(0)
ctx.use_torch_stream()
for P in [int(a) for a in sys.argv[1:]] or [2048, 4096, 8192, 16384]:
    batch = pr.random_batch(P, "ragged", seed0=0)
    plan = api.Plan(ctx, batch.seg_offsets)
    db = api.DeviceBatch(batch, "cuda:0", sample_capacity=512)
    est = api.default_options(derivative_to_optimize=4, estimate_times=1)
    plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints, limits=db.limits)
    torch.cuda.synchronize()
    t0 = db.seg_times.clone()
    lin = api.default_options(derivative_to_optimize=4)
    nl = api.default_options(derivative_to_optimize=4, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=512)
    for name, opt, reps in (("linear", lin, 50), ("nonlinear", nl, 20)):
        for tile_max in (0, 10 ** 9):
            os.environ["MRS_TG_TILE_MAX_PATHS"] = str(tile_max)

            def step():
                if name == "nonlinear":
                    db.seg_times.copy_(t0)
                    plan.solve(opt, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, limits=db.limits,
                               n_samples=db.n_samples, samples=db.samples)
                else:
                    plan.solve(opt, db.fixed_mask, db.fixed_values, t0, db.coeffs, db.status, db.cost)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                step()
            torch.cuda.synchronize()
            print("ragged P=%d %s %s: %.1f us" % (P, name, "lanes" if tile_max == 0 else "tile ", (time.perf_counter() - t) / reps * 1e6))
    plan.close()
