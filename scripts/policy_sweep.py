"""Whole-batch agreement of the policy layer (mrs_tg_optimize_paths: preprocessing, deviation loop, length check,
sampling) with the oracle's optimize() restatement over many random requests with initial states, stop flags, relaxed
heading and both waypoint generators.  ORACLE_ARITH=2: the oracle's linear solve in 113-bit arithmetic (oracle/mto_linear.c) -- 14 s per request on one core, so
only for a few dozen requests."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
deriv = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx = api.Context(0)
arith = int(os.environ.get("ORACLE_ARITH", "0"))
po.lib().mto_set_arithmetic(arith)
rng = np.random.default_rng(7)
paths, stops, inits, relax = [], [], [], []
for i in range(N):
    gen = pr.random_walk_waypoints if i % 2 else pr.random_box_waypoints
    wp = gen(3 + i % 9, 3000 + i)
    paths.append(wp)
    stops.append([bool(rng.random() < 0.2) for _ in range(len(wp))] if i % 3 == 0 else None)
    if i % 4 == 0:
        v = rng.normal(size=4) * 0.5
        inits.append(dict(heading=float(wp[0, 3]), velocity=v.tolist(), acceleration=(rng.normal(size=4) * 0.3).tolist(),
                          jerk=[0, 0, 0, 0]))
    else:
        inits.append(None)
    relax.append(i % 5 == 0)
pol = api.default_policy_options(solver=dict(derivative_to_optimize=deriv))
t0 = time.time()
out = api.optimize_paths(ctx, paths, stop_flags=stops, initial_states=inits, relax_heading=np.array(relax, dtype=np.uint8),
                         policy=pol, sample_capacity=4096)
t_gpu = time.time() - t0
same_succ = same_struct = close = 0
worst = 0.0
t0 = time.time()
for p in range(N):
    ref = po.optimize_path(paths[p], stop_at=stops[p], initial_state=inits[p], limits=pr.DEFAULT_LIMITS, relax_heading=relax[p],
                           deriv=deriv, capacity=4096)
    if out["success"][p] == ref["success"]:
        same_succ += 1
    if out["success"][p] == ref["success"] and out["n_waypoints"][p] == ref["n_waypoints"] and \
            out["iterations"][p] == ref["iterations"] and out["n_samples"][p] == ref["n_samples"]:
        same_struct += 1
        n = ref["n_samples"]
        if n:
            e = float(np.max(np.abs(out["samples"][p, :n, :3] - ref["samples"][:, :3])))
            worst = max(worst, e)
            if e < 1e-6:
                close += 1
        else:
            close += 1
print("requests %d (d=%d)%s: gpu %.2f s, oracle %.1f s" % (N, deriv, ("", "  ORACLE: exact unit-time constants",
                                                                   "  ORACLE: linear solve in 113-bit arithmetic")[arith],
                                                          t_gpu, time.time() - t0))
print("success flag equal: %.2f %%; same waypoint count / iterations / sample count: %.2f %%; of all, samples within 1e-6 m: %.2f %%"
      % (100 * same_succ / N, 100 * same_struct / N, 100 * close / N))
print("successes gpu: %d; worst sample difference among structurally equal: %.3g m" % (int(out["success"].sum()), worst))
