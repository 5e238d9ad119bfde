import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po
ctx = api.Context(0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
batch = pr.random_batch(P, 10, seed0=0)
out = ctx.solve_batch(batch, None, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=64)
bad = np.where(~np.isfinite(out["coeffs"]).reshape(P, -1).all(axis=1))[0]
print("non-finite paths:", len(bad), bad[:20])
print("bad times:", np.where(~np.isfinite(out["times"]))[0][:10], "times<0.01:", np.sum(out["times"] < 0.01))
for p in bad[:6]:
    a, b = batch.seg_offsets[p], batch.seg_offsets[p + 1]
    print("path", p, "status", out["status"][p], "times", out["times"][a:b])
    sub = batch.select([p])
    ref = po.solve_batch(sub.seg_offsets, sub.waypoints, sub.fixed_mask, sub.fixed_values, sub.limits, np.zeros(10), deriv=4,
                         time_alloc_method=2, estimate_times=True)
    print("   oracle status", ref["status"], "times", ref["times"])
    one = ctx.solve_batch(sub, None, time_alloc_method=api.TIME_ALLOC_MELLINGER)
    print("   alone  status", one["status"], "times", one["times"], "finite", np.isfinite(one["coeffs"]).all())
    lin = ctx.solve_batch(sub, out["times"][a:b])
    print("   linear solve at those times finite:", np.isfinite(lin["coeffs"]).all())
