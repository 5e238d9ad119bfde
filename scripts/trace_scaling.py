"""A benchmark path (seed = argv[1]) whose outer loop ends with a segment on the 0.01 s bound: per-segment maxima and scale
factors of the GPU and of the oracle on the oracle's own outer-loop result."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po
from tests import util

np.set_printoptions(linewidth=220, precision=6)
ctx = api.Context(0)
p = int(sys.argv[1])
batch = pr.random_batch(1, 10, seed0=p)
t0 = util.oracle_times(batch)
_, m, v = batch.path(0)
rc, t, ne, fc = po.optimize_times(4, m, v, t0, po.default_nlopt(10))
print("outer loop result", t)
c = po.solve_linear(4, m, v, t)
lim = batch.limits[0]
plan = api.Plan(ctx, batch.seg_offsets)
mx = torch.zeros(10 * 9, dtype=torch.float64, device="cuda")
plan.segment_maxima(torch.from_numpy(np.ascontiguousarray(c)).cuda(), torch.from_numpy(t).cuda(), mx)
torch.cuda.synchronize()
mx = mx.cpu().numpy().reshape(10, 3, 3)
for i in range(10):
    ref = np.zeros((3, 3))
    for k in (1, 2, 3):
        for gi, dims in enumerate(((0, 1), (2,), (3,))):
            ref[k - 1, gi] = po.segment_max_magnitude(c[i], t[i], k, dims)
    rel = np.abs(mx[i] - ref) / np.maximum(ref, 1e-300)
    print("segment %d T=%.4g  worst rel diff of the 9 maxima %.2e" % (i, t[i], rel.max()))
    if rel.max() > 1e-6:
        print("   gpu", mx[i].ravel())
        print("   ref", ref.ravel())
ok, c2, t2, sweeps = po.scale_segment_times(c, t, lim)
print("oracle scaled times", t2, "sweeps", sweeps)
