"""one path of a mixed batch through the library (current kernel selection) against the oracle: status, times, evaluations.
usage: python scripts/debug_path.py <index> [deriv] [max_segments] [n_paths]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po

idx = int(sys.argv[1])
deriv = int(sys.argv[2]) if len(sys.argv) > 2 else 4
max_seg = int(sys.argv[3]) if len(sys.argv) > 3 else 12
P = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
batch = pr.random_mixed_batch(P, deriv, seed0=0, max_segments=max_seg).select([idx])
ctx = api.Context(0)
far = pr.Batch(batch.seg_offsets, batch.waypoints, batch.fixed_mask, batch.fixed_values, np.full((1, 9), 1e9), deriv)
for name, b in (("with limits", batch), ("limits far away (the search's own end point)", far)):
    out = ctx.solve_batch(b, None, time_alloc_method=2)
    ref = po.solve_batch(b.seg_offsets, b.waypoints, b.fixed_mask, b.fixed_values, b.limits, np.zeros(b.n_segments), deriv=deriv,
                         time_alloc_method=2, runaway_rule=True, estimate_times=True)
    print(name, "S", b.n_segments, "status gpu", out["status"], "oracle", ref["status"])
    print("  gpu   ", out["times"])
    print("  oracle", ref["times"])
wp, m, v = batch.path(0)
t0 = po.estimate_times(wp, batch.limits[0])
print("mask", m.reshape(-1, 5).tolist())
print("start", t0)
for budget in range(1, 11):
    prm = po.default_nlopt(budget)
    rc, t, ne, fl = po.optimize_times(deriv, m, v, t0, prm)
    o = ctx.solve_batch(far, t0, time_alloc_method=2, max_iterations=budget)
    print("budget %2d oracle rc %d evals %d %s | gpu st %d %s" % (budget, rc, ne, np.array2string(t, precision=6), o["status"][0],
                                                              np.array2string(o["times"], precision=6)))
