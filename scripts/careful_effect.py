"""Effect of MRS_TG_FLAG_CAREFUL_COST on a batch: which paths change, and how far fast / careful are from the oracle's outer
loop on them (limits huge: no feasibility scaling, the times are the outer loop's own).  usage: careful_effect.py [n_paths]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po
from tests import util

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = api.Context(0)
batch = pr.random_batch(P, 10, seed0=0, limits=np.full(9, 1e9))
t0 = util.oracle_times(pr.random_batch(P, 10, seed0=0))   # Euclidean estimate with the default limits
res = {}
for name, fl in (("fast", 0), ("careful", api.FLAG_CAREFUL_COST)):
    for rep in range(2):
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        res[name] = ctx.solve_batch(batch, t0.copy(), time_alloc_method=api.TIME_ALLOC_MELLINGER, flags=fl)
        el = time.perf_counter() - t_start
    print("%-8s %.2f ms per host-buffer call" % (name, el * 1e3))
so = batch.seg_offsets
changed = [p for p in range(P) if not np.array_equal(res["fast"]["times"][so[p]:so[p + 1]], res["careful"]["times"][so[p]:so[p + 1]])
           or res["fast"]["status"][p] != res["careful"]["status"][p]]
print("paths whose result changed:", len(changed))
nf = nc = 0
for p in changed:
    _, m, v = batch.path(p)
    rc, t, ne, fc = po.optimize_times(4, m, v, t0[so[p]:so[p + 1]], po.default_nlopt(10))
    df = np.max(np.abs(res["fast"]["times"][so[p]:so[p + 1]] - t) / t)
    dc = np.max(np.abs(res["careful"]["times"][so[p]:so[p + 1]] - t) / t)
    nf += df < 1e-6
    nc += dc < 1e-6
    print("  path %5d oracle rc %d: fast st %d dt %.2e   careful st %d dt %.2e" % (p, rc, res["fast"]["status"][p], df, res["careful"]["status"][p], dc))
print("within 1e-6 of the oracle: fast %d, careful %d of %d" % (nf, nc, len(changed)))
