"""Per-segment maxima of the GPU's grid + polish search against the oracle's Jenkins-Traub maxima on the solved
trajectories of a random batch: worst and percentile relative differences.  usage: maxima_accuracy.py [paths]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = api.Context(0)
worst = []
for gen, d in (("box", 4), ("walk", 2), ("box", 3)):
    batch = pr.random_batch(P, 10, seed0=77000, derivative_to_optimize=d, generator=gen)
    out = ctx.solve_batch(batch, None, time_alloc_method=api.TIME_ALLOC_MELLINGER)
    plan = api.Plan(ctx, batch.seg_offsets)
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    mx = torch.zeros((batch.n_segments, 3, 3), dtype=torch.float64, device="cuda")
    plan.segment_maxima(dv(out["coeffs"]), dv(out["times"]), mx)
    torch.cuda.synchronize()
    mx = mx.cpu().numpy()
    plan.close()
    groups = [[0, 1], [2], [3]]
    rel, where = [], []
    for s in range(0, batch.n_segments, 3):
        for k in (1, 2, 3):
            for gi, grp in enumerate(groups):
                ref = po.segment_max_magnitude(out["coeffs"][s], out["times"][s], k, grp)
                rel.append(abs(mx[s, k - 1, gi] - ref) / max(ref, 1e-9))
                where.append((s, k, gi, ref, mx[s, k - 1, gi]))
    rel = np.array(rel)
    w = where[int(np.argmax(rel))]
    print("   worst: segment %d derivative %d group %d: oracle %.12g gpu %.12g (T = %.6g)" % (w + (out["times"][w[0]],)))
    print("%s d=%d: %d maxima, relative difference median %.1e  99.9 %% %.1e  worst %.1e" % (gen, d, rel.size, np.median(rel), np.percentile(rel, 99.9), rel.max()))
