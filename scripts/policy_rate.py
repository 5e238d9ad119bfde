#!/usr/bin/env python3
"""Requests per second through the policy layer (mrs_tg_optimize_paths = the reference's optimize(): preprocessing, solve,
length check, spatial validation, mid-point subdivision rounds) for batches of requests: wall time of the call, host arrays in
and out.   python scripts/policy_rate.py [n_requests ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrs_uav_trajectory_generation_amd import api, problem as pr  # noqa: E402


def oracle_rate(n=48):
    """the CPU oracle's optimize() restatement (oracle/mto_policy.c) on the first n requests of each generator, one thread"""
    from oracle import pyoracle as po
    for gen, name in ((pr.random_walk_waypoints, "walk"), (pr.random_box_waypoints, "box")):
        paths = [gen(4 + (i % 8), 7000 + i) for i in range(n)]
        t = time.perf_counter()
        rounds = [po.optimize_path(p, limits=pr.DEFAULT_LIMITS, capacity=2048)["iterations"] for p in paths]
        dt = time.perf_counter() - t
        print("%-4s CPU oracle, one thread, %d requests: %.2f ms per request, %.0f requests/s (rounds mean %.2f)" % (
            name, n, dt / n * 1e3, n / dt, float(np.mean(rounds))))


def main():
    args = sys.argv[1:]
    with_oracle = "--oracle" in args
    sizes = [int(a) for a in args if a != "--oracle"] or [1, 16, 256, 1024, 4096]
    ctx = api.Context(0)
    for n in sizes:
        for gen, name in ((pr.random_walk_waypoints, "walk"), (pr.random_box_waypoints, "box")):
            paths = [gen(4 + (i % 8), 7000 + i) for i in range(n)]
            cap = 2048
            out = api.optimize_paths(ctx, paths, sample_capacity=cap)     # warm-up (plans, pinned arenas, clocks)
            reps = 3 if n >= 1024 else 10
            t = time.perf_counter()
            for _ in range(reps):
                out = api.optimize_paths(ctx, paths, sample_capacity=cap, out=out)   # (a server's loop: the response arrays stay)
            dt = (time.perf_counter() - t) / reps
            print("%-4s %5d requests: %9.3f ms per call, %8.1f us per request, %9.0f requests/s | success %d, rounds mean %.2f max %d, "
                  "waypoints in %.1f -> out %.1f, samples mean %.0f" % (
                      name, n, dt * 1e3, dt / n * 1e6, n / dt, int(out["success"].sum()), out["iterations"].mean(),
                      out["iterations"].max(), np.mean([len(p) for p in paths]), out["n_waypoints"].mean(), out["n_samples"].mean()))
    if with_oracle:
        oracle_rate()


if __name__ == "__main__":
    main()
