#!/usr/bin/env python3
"""Measure every BASELINE.json config on one MI355X (device-resident inputs, kernel pipeline only) and
write profiles/<tag>_configs.json.  Not the headline benchmark (that is bench.py); a record of the other
configs' throughput.  Config 4 is measured as its per-GPU shard (8192 paths) and, for reference, as the whole
65536-path batch on one GPU."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrs_uav_trajectory_generation_amd import api, problem as pr  # noqa: E402


DERIV = int(os.environ.get("DERIV", "4"))   # objective order of the one-config mode (the nodelet's default config is 2)


MOVING = int(os.environ.get("MOVING", "0"))  # 1: every path starts from a moving state (velocity / acceleration / jerk of its
#                                               first vertex constrained to non-zero values: a replanning request in flight)


STOP = int(os.environ.get("STOP", "0"))      # 1: every interior waypoint is a stop_at vertex (velocity = acceleration = jerk = 0, snap free)


def with_moving_starts(batch):
    if STOP:
        parts = []
        for p in range(batch.n_paths):
            wp, _, _ = batch.path(p)
            parts.append(pr.build_vertices(wp, batch.derivative_to_optimize, stop_at=[True] * wp.shape[0]))
        batch = pr.assemble_batch(parts, batch.limits, batch.derivative_to_optimize)
    if not MOVING:
        return batch
    rng = np.random.default_rng(5)
    parts = []
    for p in range(batch.n_paths):
        wp, _, _ = batch.path(p)
        init = dict(heading=wp[0, 3], velocity=np.append(rng.uniform(-1, 1, 3), 0.1), acceleration=np.append(rng.uniform(-0.5, 0.5, 3), 0.0),
                    jerk=np.zeros(4))
        parts.append(pr.build_vertices(wp, batch.derivative_to_optimize, initial_state=init))
    return pr.assemble_batch(parts, batch.limits, batch.derivative_to_optimize)


def measure(ctx, batch, nonlinear, reps):
    batch = with_moving_starts(batch)
    plan = api.Plan(ctx, batch.seg_offsets)
    CAP = int(os.environ.get("CAP", "512"))
    db = api.DeviceBatch(batch, "cuda:0", sample_capacity=CAP)
    est = api.default_options(derivative_to_optimize=batch.derivative_to_optimize, estimate_times=1)
    plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints,
               limits=db.limits)
    torch.cuda.synchronize()
    t0 = db.seg_times.clone()
    if nonlinear:
        # (MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS: these batches' position constraints are their waypoints, as in bench.py)
        opt = api.default_options(derivative_to_optimize=batch.derivative_to_optimize, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2,
                                  sample_capacity=CAP, flags=api.FLAG_POSITIONS_ARE_WAYPOINTS | (api.FLAG_CONSTRAINED_SLOTS if (STOP or os.environ.get('HINT')) else 0))

        def step():
            db.seg_times.copy_(t0)
            plan.solve(opt, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints,
                       limits=db.limits, n_samples=db.n_samples, samples=db.samples)
    else:
        opt = api.default_options(derivative_to_optimize=batch.derivative_to_optimize, flags=api.FLAG_POSITIONS_ARE_WAYPOINTS)

        def step():
            plan.solve(opt, db.fixed_mask, db.fixed_values, t0, db.coeffs, db.status, db.cost, waypoints=db.waypoints)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    st = db.status.cpu().numpy()
    plan.close()
    return dict(paths=batch.n_paths, segments=int(batch.n_segments), ms_per_step=dt * 1e3,
                trajectories_per_s=batch.n_paths / dt, status_histogram={int(k): int(v) for k, v in zip(*np.unique(st, return_counts=True))})


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "round1"
    ctx = api.Context(0)
    ctx.use_torch_stream()
    if len(sys.argv) > 2:   # one config only, few repetitions: the command to put under rocprofv3
        which = sys.argv[2]
        if which.startswith("ragged"):   # ragged batch of any size: where the wide lane groups stop paying
            print(which, measure(ctx, pr.random_batch(int(which[6:]), "ragged", seed0=0, derivative_to_optimize=DERIV), True, 10))
            return
        if which.startswith("uniform"):  # uniform<paths>x<segments>
            n, seg = which[7:].split("x")
            print(which, measure(ctx, pr.random_batch(int(n), int(seg), seed0=0, derivative_to_optimize=DERIV), True, 10))
            return
        batch = {"config3": lambda: pr.random_batch(1024, 10, seed0=0), "config4": lambda: pr.random_batch(8192, 10, seed0=0),
                 "config5": lambda: pr.random_batch(8192, "ragged", seed0=0),
                 "config6": lambda: pr.random_batch(65536, 10, seed0=0)}[which]()
        print(which, measure(ctx, batch, True, 10))
        return
    out = {}
    out["config1_single_4wp_path_linear"] = measure(ctx, pr.config1_batch(), False, 200)
    out["config2_1024x10_linear"] = measure(ctx, pr.random_batch(1024, 10, seed0=0), False, 200)
    out["config3_1024x10_nonlinear_sampled"] = measure(ctx, pr.random_batch(1024, 10, seed0=0), True, 50)
    b8192 = pr.random_batch(8192, 10, seed0=0)
    out["config4_shard_8192x10_nonlinear_sampled"] = measure(ctx, b8192, True, 20)
    out["config4_shard_8192x10_linear"] = measure(ctx, b8192, False, 50)
    out["config5_8192_ragged_3to30_nonlinear_sampled"] = measure(ctx, pr.random_batch(8192, "ragged", seed0=0), True, 10)
    big = pr.random_batch(65536, 10, seed0=0)
    out["config4_whole_65536x10_linear_one_gpu"] = measure(ctx, big, False, 10)
    out["config4_whole_65536x10_nonlinear_sampled_one_gpu"] = measure(ctx, big, True, 5)
    for folder in ("profiles", "gpurun_out"):   # gpurun_out/ is what travels back from the GPU box
        if os.path.isdir(os.path.join(ROOT, folder)):
            with open(os.path.join(ROOT, folder, "%s_configs.json" % tag), "w") as f:
                json.dump(out, f, indent=1)
    for k, v in out.items():
        print("%-52s %10.3f ms/step %14.0f traj/s" % (k, v["ms_per_step"], v["trajectories_per_s"]))


if __name__ == "__main__":
    main()
