#!/usr/bin/env python3
"""MRS_TG_FLAG_SHARED_DEVICE on / off, 1024 x 10 linear steps on 1..4 streams, alternating in one process."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrs_uav_trajectory_generation_amd import api, problem as pr  # noqa: E402

P, S, LANES, STEPS = 1024, 10, 4, 400
batch = pr.random_batch(P, S, seed0=0)
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(LANES - 1)]
ctxs, plans, dbs = [], [], []
for st in streams:
    with torch.cuda.stream(st):
        c = api.Context(0)
        c.use_torch_stream()
        ctxs.append(c)
        plans.append(api.Plan(c, batch.seg_offsets))
        dbs.append(api.DeviceBatch(batch, "cuda:0", sample_capacity=16))
est = api.default_options(estimate_times=1)
for pl, db in zip(plans, dbs):
    pl.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints, limits=db.limits)
torch.cuda.synchronize()
calls = {}
for flag in (0, api.FLAG_SHARED_DEVICE):
    opt = api.default_options(flags=flag)
    calls[flag] = [pl.bind_solve(opt, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost)
                   for pl, db in zip(plans, dbs)]


def run(flag, lanes, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        calls[flag][k % lanes]()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


rr = {f: {n: api.RoundRobin(calls[f][:n]) for n in (1, 2, 4)} for f in calls}


def run_c(flag, lanes, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rr[flag][lanes](steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


run(0, 4, 600)
for rep in range(3):
    for lanes in (1, 2, 4):
        print("lanes %d: flag off %.2f us/step, flag on %.2f us/step; issued from C: off %.2f, on %.2f" %
              (lanes, run(0, lanes, STEPS), run(4, lanes, STEPS), run_c(0, lanes, STEPS), run_c(4, lanes, STEPS)))
