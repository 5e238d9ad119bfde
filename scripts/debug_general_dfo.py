"""the position-free batch of tests/test_gpu_general_patterns.py in a gradient-free mode: the paths on which library and oracle
keep different times, with the oracle's objective at both.   usage: python scripts/debug_general_dfo.py <mode> <n_seg|ragged>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po
from tests.test_gpu_general_patterns import _free_some_positions, _oracle

mode = int(sys.argv[1])
n_seg = sys.argv[2] if sys.argv[2] == "ragged" else int(sys.argv[2])
rng = np.random.default_rng(40 + mode)
base = pr.random_batch(48, n_seg, seed0=1400)
batch, touched = _free_some_positions(base, rng, share=0.2)
ctx = api.Context(0)
out = ctx.solve_batch(batch, None, time_alloc_method=mode, max_iterations=10)
ref = _oracle(batch, mode, max_iterations=10)
so = batch.seg_offsets
for p in range(batch.n_paths):
    a, b = so[p], so[p + 1]
    if np.max(np.abs(out["times"][a:b] - ref["times"][a:b]) / ref["times"][a:b]) < 1e-13:
        continue
    wp, m, v = batch.path(p)
    t0 = po.estimate_times(wp, batch.limits[p])
    print("path", p, "touched", p in touched, "S", b - a, "mask", m.reshape(-1, 5)[:, 0].tolist())
    for name, t in (("start", t0), ("gpu", out["times"][a:b]), ("oracle", ref["times"][a:b])):
        f, parts = po.objective_time(4, m, v, t, batch.limits[p], mode=mode)
        print("  %-6s f %.9g parts %s  t %s" % (name, f, parts, np.array2string(t, precision=5)))
    print("  cost gpu %.9g oracle %.9g" % (out["cost"][p], ref["cost"][p]))
    sub = batch.select([p])
    for budget in range(1, 11):
        o = ctx.solve_batch(sub, None, time_alloc_method=mode, max_iterations=budget)
        r = _oracle(sub, mode, max_iterations=budget)
        fo = po.objective_time(4, m, v, o["times"], batch.limits[p], mode=mode)[0]
        fr = po.objective_time(4, m, v, r["times"], batch.limits[p], mode=mode)[0]
        print("  budget %2d gpu f %.9g %s | oracle f %.9g %s" % (budget, fo, np.array2string(o["times"], precision=5), fr,
                                                                 np.array2string(r["times"], precision=5)))
