#!/usr/bin/env python3
"""Per-dispatch duration of the saturated-device fixed-times solve (solve_quad_kernel) with vertex positions read from the
value array and, under MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS, from the compact waypoint array; results compared bit for bit.
    python scripts/quad_ab.py [paths ...]      (default 65536 8192)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrs_uav_trajectory_generation_amd import api, problem as pr  # noqa: E402

ctx = api.Context(0)
ctx.use_torch_stream()
for P in [int(a) for a in sys.argv[1:]] or [65536, 8192]:
    batch = pr.random_batch(P, 10, seed0=0)
    plan = api.Plan(ctx, batch.seg_offsets)
    db = api.DeviceBatch(batch, "cuda:0")
    est = api.default_options(derivative_to_optimize=4, estimate_times=1)
    plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints, limits=db.limits)
    torch.cuda.synchronize()
    res = {}
    for name, flags in (("values", 0), ("waypoints", api.FLAG_POSITIONS_ARE_WAYPOINTS)):
        opt = api.default_options(derivative_to_optimize=4, flags=flags)
        call = plan.bind_solve(opt, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints)
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        ctx.set_profiling(True)
        api.kernel_trace_reset()
        for _ in range(30):
            call()
        v = sorted(ctx.kernel_ms_history(api.KERNEL_SOLVE_LINEAR, 64))
        ctx.set_profiling(False)
        res[name] = (db.coeffs.cpu().numpy().copy(), db.cost.cpu().numpy().copy())
        comp = P * (40 * 10 + 288 + 328 * 10)
        print("%6d x 10  positions from %-9s  %s: mean %.1f us  median %.1f us  min %.1f us  -> %.2f TB/s of compulsory bytes (%.3f of 8 TB/s)"
              % (P, name, api.kernel_trace()[-1], 1e3 * sum(v) / len(v), 1e3 * v[len(v) // 2], 1e3 * v[0], comp / (sum(v) / len(v) * 1e-3) / 1e12,
                 comp / (sum(v) / len(v) * 1e-3) / 8e12))
    print("       bit-identical:", bool(np.array_equal(res["values"][0], res["waypoints"][0]) and np.array_equal(res["values"][1], res["waypoints"][1])))
    plan.close()
