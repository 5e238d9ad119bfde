#!/usr/bin/env python3
"""A / B of a library environment knob on the large Mellinger batches: pipeline time per step and the outer-loop kernel's own
per-dispatch time (events on the launch) for 8192 x 10, 8192 ragged and 65536 x 10, one child process per setting.
usage: python scripts/lean_ab.py [ENV_NAME=MRS_TG_LEAN_SHARED] [values=0,1] [configs=config4,config5,config6]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(which):
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from mrs_uav_trajectory_generation_amd import api, problem as pr
    from scripts.measure_configs import measure
    batch = {"config3": lambda: pr.random_batch(1024, 10, seed0=0), "config4": lambda: pr.random_batch(8192, 10, seed0=0),
             "config5": lambda: pr.random_batch(8192, "ragged", seed0=0), "config6": lambda: pr.random_batch(65536, 10, seed0=0),
             "config2048": lambda: pr.random_batch(2048, 10, seed0=0), "config4096": lambda: pr.random_batch(4096, 10, seed0=0)}[which]()
    ctx = api.Context(0)
    ctx.use_torch_stream()
    r = measure(ctx, batch, True, 10)
    rl = measure(ctx, batch, False, 20)
    # the outer-loop kernel(s) alone
    plan = api.Plan(ctx, batch.seg_offsets)
    db = api.DeviceBatch(batch, "cuda:0", sample_capacity=512)
    est = api.default_options(derivative_to_optimize=4, estimate_times=1)
    plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints, limits=db.limits)
    torch.cuda.synchronize()
    t0 = db.seg_times.clone()
    opt = api.default_options(derivative_to_optimize=4, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=512)
    ctx.set_profiling(True)
    for _ in range(8):
        db.seg_times.copy_(t0)
        plan.solve(opt, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, limits=db.limits,
                   n_samples=db.n_samples, samples=db.samples)
    vals = ctx.kernel_ms_history(api.KERNEL_NONLINEAR, 8)
    ctx.set_profiling(False)
    print("%s: linear %.1f us per step; Mellinger pipeline %.1f us per step, outer-loop kernel %.1f us (median of %d), checksum times %.9f status %s"
          % (which, rl["ms_per_step"] * 1e3, r["ms_per_step"] * 1e3, float(np.median(vals)) * 1e3, len(vals), float(db.seg_times.sum().item()),
             r["status_histogram"]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    name = sys.argv[1] if len(sys.argv) > 1 else "MRS_TG_LEAN_SHARED"
    values = (sys.argv[2] if len(sys.argv) > 2 else "0,1").split(",")
    configs = (sys.argv[3] if len(sys.argv) > 3 else "config4,config5,config6").split(",")
    for which in configs:
        for v in values:
            env = dict(os.environ)
            env[name] = v
            print("%s=%s " % (name, v), end="", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", which], env=env, cwd=ROOT)
