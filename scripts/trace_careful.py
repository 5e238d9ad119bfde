"""One benchmark path (seed = argv[1]) with growing evaluation budgets: oracle, fast GPU kernel, GPU with the careful re-run."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po
from tests import util

np.set_printoptions(linewidth=220, precision=5)
ctx = api.Context(0)
p = int(sys.argv[1])
batch = pr.random_batch(1, 10, seed0=p)
t0 = util.oracle_times(batch)
_, m, v = batch.path(0)
print("start", t0)
for budget in range(1, 11):
    rc, t, ne, fc = po.optimize_times(4, m, v, t0, po.default_nlopt(budget))
    res = {}
    for name, fl in (("fast", 0), ("careful", api.FLAG_CAREFUL_COST)):
        # (the pipeline's feasibility scaling is switched off by huge limits so that the times are the outer loop's own)
        b2 = pr.random_batch(1, 10, seed0=p, limits=np.full(9, 1e9))
        out = ctx.solve_batch(b2, t0.copy(), time_alloc_method=api.TIME_ALLOC_MELLINGER, max_iterations=budget, flags=fl)
        res[name] = (int(out["status"][0]), out["times"])
    print("budget %2d  oracle rc %d ne %d" % (budget, rc, ne), t)
    for name in res:
        print("           %-8s st %d   " % (name, res[name][0]), res[name][1], " max rel diff %.2e" % np.max(np.abs(res[name][1] - t) / t))
