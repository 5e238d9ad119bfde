#!/usr/bin/env python3
"""How many paths of a batch see the by-product cost fail its guard during the Mellinger outer loop
(mrs_tg_plan_careful_count), for the batches of the parity sweep."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrs_uav_trajectory_generation_amd import api, problem as pr  # noqa: E402

ctx = api.Context(0)
for n, seg, d, gen in ((1024, 10, 4, "box"), (8192, 10, 4, "box"), (65536, 10, 4, "box"), (8192, "ragged", 4, "box"),
                       (8192, 10, 2, "walk"), (4096, 10, 3, "box"), (16384, 10, 4, "mixed"), (16384, 10, 3, "mixed"),
                       (16384, 10, 2, "mixed")):
    batch = pr.random_mixed_batch(n, d, seed0=0) if gen == "mixed" else pr.random_batch(n, seg, seed0=0, derivative_to_optimize=d,
                                                                                       generator=gen)
    plan = api.Plan(ctx, batch.seg_offsets)
    db = api.DeviceBatch(batch, "cuda:0", sample_capacity=16)
    est = api.default_options(derivative_to_optimize=d, estimate_times=1)
    plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints, limits=db.limits)
    nl = api.default_options(derivative_to_optimize=d, time_alloc_method=api.TIME_ALLOC_MELLINGER, flags=api.FLAG_CAREFUL_COST)
    plan.solve(nl, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, limits=db.limits)
    torch.cuda.synchronize()
    print("%6d paths  segments %-6s d=%d  %-5s: %d guarded paths" % (n, seg, d, gen, plan.careful_count()), flush=True)
    plan.close()
