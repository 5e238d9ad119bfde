#!/usr/bin/env python3
"""How long the paths of the policy layer get: percentiles of the final waypoint count of 1024 box-generator requests after the
deviation-subdivision rounds (a third end above 60 waypoints -- why the outer loop of 61 .. 121 segments matters)."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from mrs_uav_trajectory_generation_amd import api, problem as pr
ctx = api.Context(0)
paths = [pr.random_box_waypoints(4 + (i % 8), 7000 + i) for i in range(1024)]
out = api.optimize_paths(ctx, paths, sample_capacity=2048)
nw = out["n_waypoints"]
print("final waypoints percentiles 10/50/90/99/max:", np.percentile(nw, [10, 50, 90, 99]), nw.max(), " share > 61 waypoints:", (nw > 61).mean())
