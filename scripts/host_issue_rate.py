"""How fast can the host issue linear steps, and how fast does the GPU retire them?  (bench.py's four lanes)
python scripts/host_issue_rate.py [paths]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mrs_uav_trajectory_generation_amd import api, problem as pr

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
batch = pr.random_batch(P, 10, seed0=0)
dev = torch.device("cuda", 0)
lanes = []
for i in range(4):
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        ctx = api.Context(0)
        ctx.use_torch_stream()
        plan = api.Plan(ctx, batch.seg_offsets)
        db = api.DeviceBatch(batch, dev)
        est = api.default_options(estimate_times=1)
        plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints, limits=db.limits)
        call = plan.bind_solve(api.default_options(), db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost)
    lanes.append((st, ctx, plan, db, call))
torch.cuda.synchronize()
for n_lanes in (1, 2, 4):
    for _ in range(300):
        for i in range(n_lanes):
            lanes[i][4]()
    torch.cuda.synchronize()
    N = 2000
    t0 = time.perf_counter()
    for k in range(N):
        lanes[k % n_lanes][4]()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%d lane(s): host issues a step every %.2f us; %d steps retired in %.2f us per step" % (n_lanes, (t1 - t0) / N * 1e6, N, (t2 - t0) / N * 1e6))

# the driver's shape: 5 untimed steps, synchronise, 20 timed steps, synchronise
for n_lanes in (1, 2, 4):
    res = []
    for rep in range(6):
        for k in range(5):
            lanes[k % n_lanes][4]()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(20):
            lanes[k % n_lanes][4]()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6))
    print("%d lane(s), 20 steps: issue / total us per run:" % n_lanes, " ".join("%.0f/%.0f" % r for r in res))
