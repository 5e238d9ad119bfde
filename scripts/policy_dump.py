"""GPU side of the policy sweep only: write the results to gpurun_out/policy_dump_d<deriv>.npz (compare on the CPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mrs_uav_trajectory_generation_amd import api, problem as pr
N = int(sys.argv[1]); deriv = int(sys.argv[2])
ctx = api.Context(0)
rng = np.random.default_rng(7)
paths, stops, inits, relax = [], [], [], []
for i in range(N):
    gen = pr.random_walk_waypoints if i % 2 else pr.random_box_waypoints
    wp = gen(3 + i % 9, 3000 + i)
    paths.append(wp)
    stops.append([bool(rng.random() < 0.2) for _ in range(len(wp))] if i % 3 == 0 else None)
    if i % 4 == 0:
        v = rng.normal(size=4) * 0.5
        inits.append(dict(heading=float(wp[0, 3]), velocity=v.tolist(), acceleration=(rng.normal(size=4) * 0.3).tolist(), jerk=[0, 0, 0, 0]))
    else:
        inits.append(None)
    relax.append(i % 5 == 0)
pol = api.default_policy_options(solver=dict(derivative_to_optimize=deriv))
out = api.optimize_paths(ctx, paths, stop_flags=stops, initial_states=inits, relax_heading=np.array(relax, dtype=np.uint8), policy=pol, sample_capacity=4096)
np.savez(os.path.join("gpurun_out", "policy_dump_d%d.npz" % deriv), success=out["success"], n_waypoints=out["n_waypoints"],
         iterations=out["iterations"], n_samples=out["n_samples"], max_deviation=out["max_deviation"])
print("done", int(out["success"].sum()))
