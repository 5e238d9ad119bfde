"""Does a HIP graph of the linear step (assembly + solve) beat plain stream launches?  (experiment)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mrs_uav_trajectory_generation_amd import api, problem as pr
torch.zeros(1, device="cuda")
ctx = api.Context(0)
batch = pr.random_batch(1024, 10, seed0=0)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ctx.use_torch_stream()
    plan = api.Plan(ctx, batch.seg_offsets)
    db = api.DeviceBatch(batch, "cuda:0", sample_capacity=512)
    est = api.default_options(derivative_to_optimize=4, estimate_times=1)
    plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints, limits=db.limits)
    opt = api.default_options(derivative_to_optimize=4)
    t0 = db.seg_times.clone()
    def step():
        plan.solve(opt, db.fixed_mask, db.fixed_values, t0, db.coeffs, db.status, db.cost)
    for _ in range(20): step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(500): step()
    torch.cuda.synchronize()
    print("stream launches: %.2f us/step" % ((time.perf_counter() - t) / 500 * 1e6))
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            for _ in range(10): step()
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize()
        print("graph of 10 steps: %.2f us/step" % ((time.perf_counter() - t) / 500 * 1e6))
    except Exception as e:
        print("graph capture failed:", repr(e)[:300])
