"""Trace one benchmark path (seed = argv[1]) through the outer loop: the oracle's trial point after every evaluation
budget next to the GPU pipeline's result, and J / gradient of both at those points (how the 65536-path runaway was found)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po
from tests import util
np.set_printoptions(linewidth=200, precision=6)
import torch; torch.zeros(1, device="cuda"); ctx = api.Context(0)
p = int(sys.argv[1])
batch = pr.random_batch(1, 10, seed0=p)
t0 = util.oracle_times(batch)
_, m, v = batch.path(0)
print("start times", t0)
for budget in range(1, 11):
    prm = po.default_nlopt(budget)
    rc, t, ne, fc = po.optimize_times(4, m, v, t0, prm)
    # GPU: outer loop only is not exposed; use solve with scaling disabled? compare cost/gradient at the oracle's point instead
    out = ctx.solve_batch(batch, t0.copy(), time_alloc_method=api.TIME_ALLOC_MELLINGER, max_iterations=budget,
                          )
    print("budget", budget, "oracle rc", rc, "ne", ne, "t", t)
    print("          gpu status", out["status"], "t(after scaling)", out["times"])

import torch
def gpu_cg(t):
    plan = api.Plan(ctx, batch.seg_offsets)
    cost = torch.zeros(1, dtype=torch.float64, device="cuda"); grad = torch.zeros(10, dtype=torch.float64, device="cuda")
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    plan.cost_gradient(4, dv(batch.fixed_mask), dv(batch.fixed_values), dv(t), cost, grad)
    torch.cuda.synchronize(); plan.close()
    return cost.cpu().numpy()[0], grad.cpu().numpy()
for budget in (1, 2, 3, 4, 5):
    rc, t, ne, fc = po.optimize_times(4, m, v, t0, po.default_nlopt(budget))
    Jo, go = po.cost_and_gradient(4, m, v, t)
    Jg, gg = gpu_cg(t)
    print("point of evaluation", budget, "J oracle %.10e gpu %.10e rel %.2e" % (Jo, Jg, abs(Jo-Jg)/abs(Jo)))
    print("   grad oracle", go)
    print("   grad gpu   ", gg)
