#!/usr/bin/env python3
"""Where a 20-step timed region spends its time: empty synchronize, one launch + synchronize, 20 launches + synchronize."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrs_uav_trajectory_generation_amd import api, problem as pr  # noqa: E402

P, S, LANES = 1024, 10, 4
batch = pr.random_batch(P, S, seed0=0)
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(LANES - 1)]
plans, dbs, calls, ctxs = [], [], [], []
est = api.default_options(estimate_times=1)
opt = api.default_options(flags=api.FLAG_SHARED_DEVICE)
for st in streams:
    with torch.cuda.stream(st):
        c = api.Context(0)
        c.use_torch_stream()
        pl = api.Plan(c, batch.seg_offsets)
        db = api.DeviceBatch(batch, "cuda:0", sample_capacity=16)
        pl.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints, limits=db.limits)
        ctxs.append(c), plans.append(pl), dbs.append(db)
        calls.append(pl.bind_solve(opt, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost))
THREADS = int(os.environ.get("ISSUE_THREADS", "1"))
rr = api.RoundRobin(calls, threads=THREADS)
print("issuing threads:", THREADS)
rr(600)
torch.cuda.synchronize()


def timed(n, reps=30):
    best, tot = 1e9, 0.0
    for _ in range(reps):
        rr(300)   # keep the clocks up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if n:
            rr(n)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        best = min(best, t2 - t0)
        tot += t2 - t0
    return best * 1e6, tot / reps * 1e6, (t1 - t0) * 1e6


for n in (0, 1, 4, 20, 100):
    b, m, issue = timed(n)
    print("%3d launches + synchronize: best %.1f us, mean %.1f us (issue loop of the last repetition %.1f us)" % (n, b, m, issue))
