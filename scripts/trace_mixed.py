"""Trace paths of the mixed-constraint sweep batch (parity_sweep.py, generator 'mixed') through the outer loop: for every
evaluation budget the oracle's and the GPU's result of the whole pipeline, and J / gradient of both at the start point.
usage: trace_mixed.py deriv path [path ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po
from tests import util

np.set_printoptions(linewidth=220, precision=8)
deriv = int(sys.argv[1])
paths = [int(a) for a in sys.argv[2:]]
full = pr.random_mixed_batch(max(paths) + 1, deriv)
so = full.seg_offsets
torch.zeros(1, device="cuda")
ctx = api.Context(0)


def sub(p):
    a, b = so[p], so[p + 1]
    va, vb = a + p, b + p + 1
    return pr.Batch(np.array([0, b - a], dtype=np.int32), full.waypoints[va:vb].copy(), full.fixed_mask[va:vb].copy(),
                    full.fixed_values[va:vb].copy(), full.limits[p:p + 1].copy(), deriv)


def gpu_cg(bb, t):
    plan = api.Plan(ctx, bb.seg_offsets)
    n = len(t)
    cost = torch.zeros(1, dtype=torch.float64, device="cuda")
    grad = torch.zeros(n, dtype=torch.float64, device="cuda")
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    plan.cost_gradient(deriv, dv(bb.fixed_mask), dv(bb.fixed_values), dv(t), cost, grad)
    torch.cuda.synchronize()
    plan.close()
    return cost.cpu().numpy()[0], grad.cpu().numpy()


for p in paths:
    bb = sub(p)
    t0 = util.oracle_times(bb)
    _, m, v = bb.path(0)
    print("=== path", p, "S", len(t0), "mask rows", m.tolist())
    print("start times", t0)
    Jo, go = po.cost_and_gradient(deriv, m, v, t0)
    Jg, gg = gpu_cg(bb, t0)
    print("J oracle %.12e gpu %.12e rel %.2e; grad max rel diff %.2e" % (Jo, Jg, abs(Jo - Jg) / abs(Jo), np.max(np.abs(go - gg)) / np.max(np.abs(go))))
    if os.environ.get("TRACE_NO_SCALING"):
        bb.limits[:] = 1e12  # no feasibility scaling: the returned times are the outer loop's last evaluated point
    for budget in range(1, 11):
        ref = po.solve_batch(bb.seg_offsets, bb.waypoints, bb.fixed_mask, bb.fixed_values, bb.limits, t0.copy(), deriv=deriv,
                             time_alloc_method=2, runaway_rule=True, estimate_times=False, sampling_dt=0.0, sample_capacity=0, n_threads=1,
                             max_iterations=budget)
        out = ctx.solve_batch(bb, t0.copy(), time_alloc_method=2, max_iterations=budget)
        d = np.max(np.abs(out["times"] - ref["times"]) / ref["times"])
        print("budget %2d status oracle %d gpu %d  max rel dt %.2e" % (budget, ref["status"][0], out["status"][0], d))
        if os.environ.get("TRACE_NO_SCALING"):
            Jo, go = po.cost_and_gradient(deriv, m, v, ref["times"])
            Jg, gg = po.cost_and_gradient(deriv, m, v, out["times"])
            print("   oracle point", ref["times"], "J %.10e" % Jo)
            print("   gpu point   ", out["times"], "J %.10e (oracle's J at the GPU's point)" % Jg)
            Jg2, gg2 = gpu_cg(bb, ref["times"])
            print("   at the oracle's point: gpu J %.10e; grad oracle" % Jg2, go)
            print("                                              grad gpu   ", gg2)
        if d > 1e-6:
            print("   oracle", ref["times"])
            print("   gpu   ", out["times"])
            rc, t, ne, fc = po.optimize_times(deriv, m, v, t0, po.default_nlopt(budget))
            print("   oracle outer-loop point (before scaling)", t, "rc", rc, "ne", ne)
            Jo, go = po.cost_and_gradient(deriv, m, v, t)
            Jg, gg = gpu_cg(bb, t)
            print("   at that point: J oracle %.12e gpu %.12e; grad rel diff %.2e" % (Jo, Jg, np.max(np.abs(go - gg)) / np.max(np.abs(go))))
            break
