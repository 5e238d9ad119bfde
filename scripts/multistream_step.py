"""Independent 1024-path linear steps issued round-robin on S HIP streams (one context + plan per stream): how much
of the GPU a single small batch leaves idle.  usage: multistream_step.py [paths] [workload]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mrs_uav_trajectory_generation_amd import api, problem as pr

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nonlinear = len(sys.argv) > 2 and sys.argv[2] == "nonlinear"
batch = pr.random_batch(P, 10, seed0=0)
dev = torch.device("cuda", 0)
for S in (1, 2, 3, 4, 6, 8):
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    ctxs, plans, dbs, t0s = [], [], [], []
    for s in streams:
        with torch.cuda.stream(s):
            c = api.Context(0)
            c.use_torch_stream()
            p = api.Plan(c, batch.seg_offsets)
            db = api.DeviceBatch(batch, dev, sample_capacity=512)
            est = api.default_options(derivative_to_optimize=4, estimate_times=1)
            p.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints,
                    limits=db.limits)
            ctxs.append(c); plans.append(p); dbs.append(db)
    torch.cuda.synchronize()
    t0s = [db.seg_times.clone() for db in dbs]
    opt = api.default_options(derivative_to_optimize=4, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2,
                              sample_capacity=512) if nonlinear else api.default_options(derivative_to_optimize=4)

    if nonlinear:
        calls = [plans[i].bind_solve(opt, dbs[i].fixed_mask, dbs[i].fixed_values, dbs[i].seg_times, dbs[i].coeffs, dbs[i].status,
                                     dbs[i].cost, limits=dbs[i].limits, n_samples=dbs[i].n_samples, samples=dbs[i].samples)
                 for i in range(S)]
    else:
        calls = [plans[i].bind_solve(opt, dbs[i].fixed_mask, dbs[i].fixed_values, t0s[i], dbs[i].coeffs, dbs[i].status, dbs[i].cost)
                 for i in range(S)]

    def step(k):
        i = k % S
        if nonlinear:
            with torch.cuda.stream(streams[i]):
                dbs[i].seg_times.copy_(t0s[i])   # the outer loop overwrites the times: restart from the same point
        calls[i]()   # the context is bound to streams[i]

    n = 2000 if not nonlinear else 400
    for k in range(50):
        step(k)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(n):
        step(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print("%d stream(s): %.2f us per %d-path step = %.1f M trajectories/s" % (S, dt * 1e6, P, P / dt / 1e6))
    for p in plans:
        p.close()
