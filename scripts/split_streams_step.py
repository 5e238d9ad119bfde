"""One 8192-path Mellinger pipeline on one stream against the same paths as two 4096-path halves on two streams
(two contexts): does overlapping the halves' low-utilisation stages pay?  usage: split_streams_step.py [n_paths] [n_parts]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mrs_uav_trajectory_generation_amd import api, problem as pr

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
PARTS = int(sys.argv[2]) if len(sys.argv) > 2 else 2


def make(n_paths, seed0, stream):
    with torch.cuda.stream(stream):
        ctx = api.Context(0)
        ctx.use_torch_stream()
        batch = pr.random_batch(n_paths, 10, seed0=seed0)
        plan = api.Plan(ctx, batch.seg_offsets)
        db = api.DeviceBatch(batch, "cuda:0", sample_capacity=512)
        est = api.default_options(estimate_times=1)
        plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints, limits=db.limits)
        t0 = db.seg_times.clone()
        opt = api.default_options(time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=512)

        def step():
            with torch.cuda.stream(stream):
                db.seg_times.copy_(t0)
                plan.solve(opt, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, limits=db.limits,
                           n_samples=db.n_samples, samples=db.samples)
        return step, (ctx, plan, db)


def timeit(steps, reps):
    for _ in range(5):
        for s in steps:
            s()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        for s in steps:
            s()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6


whole, keep0 = make(P, 0, torch.cuda.current_stream())
print("%d paths, one stream: %.1f us per step" % (P, timeit([whole], 30)))
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(PARTS - 1)]
parts = [make(P // PARTS, i * (P // PARTS), streams[i]) for i in range(PARTS)]
print("%d x %d paths on %d streams: %.1f us per step" % (PARTS, P // PARTS, PARTS, timeit([p[0] for p in parts], 30)))
