"""Which pairs of HIP streams overlap two grouped dispatches of the headline (10 x 1024 paths x 10 segments each)?  Wall time from
issue to synchronize for every pair of 1 + N streams (0 = torch's current stream); profiles/round6_stream_pairs.txt.
  python scripts/stream_pairs.py [N side streams, default 6]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrs_uav_trajectory_generation_amd import api, problem as pr

dev = torch.device("cuda:0")
P, S = 1024, 10
batch = pr.random_batch(P, S, seed0=0)
ctx0 = api.Context(0)
ctx0.use_torch_stream()
n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 6
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(n_side)]
ctxs = [ctx0]
for st in streams[1:]:
    with torch.cuda.stream(st):
        c = api.Context(0)
        c.use_torch_stream()
        ctxs.append(c)
plans = [api.Plan(c, batch.seg_offsets) for c in ctxs]
db = api.DeviceBatch(batch, dev, sample_capacity=0)
est = api.default_options(derivative_to_optimize=4, estimate_times=1)
plans[0].solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints, limits=db.limits)
torch.cuda.synchronize()
opt = api.default_options(derivative_to_optimize=4, flags=api.FLAG_POSITIONS_ARE_WAYPOINTS)
nS = batch.n_segments
outs = [(torch.zeros((nS, 4, 10), dtype=torch.float64, device=dev), torch.zeros(P, dtype=torch.int32, device=dev),
         torch.zeros(P, dtype=torch.float64, device=dev)) for _ in range(20)]
t_fixed = db.seg_times.clone()


def region_us(a, b, reps=15):
    calls = []
    for k in range(20):
        lane = a if k < 10 else b
        c, st, co = outs[k]
        calls.append(plans[lane].bind_solve(opt, db.fixed_mask, db.fixed_values, t_fixed, c, st, co, waypoints=db.waypoints))
    rr = api.RoundRobin(calls, grouped=True)
    for _ in range(5):
        rr(20)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rr(20)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    return float(np.median(ts)), float(np.min(ts))


# clock ramp
rr0 = None
for _ in range(3):
    region_us(0, 1, reps=30)
res = {}
for a in range(len(ctxs)):
    for b in range(a + 1, len(ctxs)):
        res[(a, b)] = region_us(a, b)
for (a, b), (med, mn) in sorted(res.items(), key=lambda kv: kv[1][0]):
    print("streams (%d, %d): region median %.1f us  min %.1f us   [stream ids %s %s]" % (a, b, med, mn, hex(streams[a].cuda_stream), hex(streams[b].cuda_stream)))
print("same stream twice (0, 0): %.1f us" % region_us(0, 0)[0])
