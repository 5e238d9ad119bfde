"""Time of one objective evaluation (mrs_tg_plan_cost_gradient) for a batch; MRS_TG_PS=0|1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrs_uav_trajectory_generation_amd import api, problem as pr
P = int(sys.argv[1]); n_seg = sys.argv[2]; n_seg = n_seg if n_seg == "ragged" else int(n_seg)
ctx = api.Context(0); ctx.use_torch_stream()
batch = pr.random_batch(P, n_seg, seed0=0)
plan = api.Plan(ctx, batch.seg_offsets)
db = api.DeviceBatch(batch, "cuda:0", sample_capacity=16)
est = api.default_options(estimate_times=1)
plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints, limits=db.limits)
cost = torch.zeros(P, dtype=torch.float64, device="cuda"); grad = torch.zeros(batch.n_segments, dtype=torch.float64, device="cuda")
for _ in range(5):
    plan.cost_gradient(4, db.fixed_mask, db.fixed_values, db.seg_times, cost, grad)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    plan.cost_gradient(4, db.fixed_mask, db.fixed_values, db.seg_times, cost, grad)
torch.cuda.synchronize()
print("PS=%s %d x %s: %.1f us per evaluation of the batch; checksum %.12e" % (os.environ.get("MRS_TG_PS"), P, n_seg, (time.perf_counter() - t0) / 50 * 1e6, float(grad.sum())))
