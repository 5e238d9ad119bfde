#!/usr/bin/env python3
"""The grouped dispatch (ten bound fixed-times solves of 1024 x 10 paths per launch, as bench.py's headline issues them) under
an objective order below snap: trajectories/s with solve_quad_group_kernel<., true> (free end slots eliminated in the kernel)
and, with MRS_TG_QUAD_ENDS=0 in the environment, with the general step.   python scripts/grouped_below_snap.py [deriv]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrs_uav_trajectory_generation_amd import api, problem as pr  # noqa: E402


def main():
    d = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ctx = api.Context(0)
    ctx.use_torch_stream()
    P, S, G = 1024, 10, 10
    batch = pr.random_batch(P, S, seed0=0, derivative_to_optimize=d)
    plan = api.Plan(ctx, batch.seg_offsets)
    dbs = [api.DeviceBatch(batch, "cuda:0") for _ in range(G)]
    est = api.default_options(derivative_to_optimize=d, estimate_times=1)
    for db in dbs:
        plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints,
                   limits=db.limits)
    torch.cuda.synchronize()
    opt = api.default_options(derivative_to_optimize=d)
    calls = [plan.bind_solve(opt, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost) for db in dbs]
    rr = api.RoundRobin(calls, grouped=True)
    api.kernel_trace_reset()
    rr(G)
    print("kernel:", api.kernel_trace()[-1])
    for _ in range(20):
        rr(G)
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 200
    for _ in range(n):
        rr(G)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print("d=%d: %.2f us per dispatch of %d x %d paths, %.1f M trajectories/s (one stream)" % (d, dt * 1e6, G, P, G * P / dt / 1e6))


if __name__ == "__main__":
    main()
