"""Latency of the one-request host interface (mrs_tg_find_trajectory / mrs_tg_optimize_paths: plan creation, allocation,
H2D, kernels, D2H) for the reference tests' four-waypoint path and for a batch of 64 such requests."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mrs_uav_trajectory_generation_amd import api, problem as pr

ctx = api.Context(0)
wp = pr.CONFIG1_WAYPOINTS
def timeit(f, n):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
print("find_trajectory (Mellinger, d=2, 4 waypoints): %.3f ms" % timeit(lambda: ctx.find_trajectory(wp), 50))
path = np.vstack([[0, 0, 3, 0.5], wp])
print("optimize_paths, 1 request (deviation loop):    %.3f ms" % timeit(lambda: api.optimize_paths(ctx, [path], sample_capacity=2048), 20))
print("optimize_paths, 64 requests in one call:       %.3f ms" % timeit(lambda: api.optimize_paths(ctx, [path] * 64, sample_capacity=2048), 10))
b = pr.random_batch(1024, 10, seed0=0)
print("solve_batch host buffers, 1024x10 linear:      %.3f ms" % timeit(lambda: ctx.solve_batch(b, None), 20))
print("solve_batch host buffers, 1024x10 nonlinear:   %.3f ms" % timeit(lambda: ctx.solve_batch(b, None, time_alloc_method=2, sampling_dt=0.2, sample_capacity=512), 10))
