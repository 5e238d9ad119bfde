#!/usr/bin/env python3
"""Back-to-back time of the bound fixed-times solve at 65536 x 10 and 8192 x 10 (solve_quad_kernel from the value array and from
the waypoint array): the same-box A / B of a library against another one (MRS_TG_LIB_PATH=<other libmrs_tg.so>)."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from mrs_uav_trajectory_generation_amd import api, problem as pr
ctx = api.Context(0); ctx.use_torch_stream()
for n in (65536, 8192):
    b = pr.random_batch(n, 10, seed0=0)
    plan = api.Plan(ctx, b.seg_offsets); db = api.DeviceBatch(b, "cuda:0")
    est = api.default_options(derivative_to_optimize=4, estimate_times=1)
    plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints, limits=db.limits)
    torch.cuda.synchronize()
    for flags in (0, api.FLAG_POSITIONS_ARE_WAYPOINTS):
        opt = api.default_options(derivative_to_optimize=4, flags=flags)
        call = plan.bind_solve(opt, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints)
        for _ in range(5): call()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): call()
        torch.cuda.synchronize()
        print("linear %d x 10 flags %d: %.1f us" % (n, flags, (time.perf_counter() - t) / 50 * 1e6))
