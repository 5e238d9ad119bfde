"""Per-path agreement of the Mellinger pipeline with the oracle on long paths (129 .. 256 segments): which criterion differs."""
import sys
import numpy as np
sys.path.insert(0, ".")
from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po
from tests import util
from tests.test_gpu_large_batches import _moving

ctx = api.Context(0)
for deriv, n_seg, moving in [(4, 129, False), (2, 200, True), (2, 256, False), (4, 256, True)]:
    batch = pr.random_batch(6, n_seg, seed0=8600 + n_seg, derivative_to_optimize=deriv)
    if moving:
        batch = _moving(batch, seed=11)
    cap = 8192
    api.kernel_trace_reset()
    out = ctx.solve_batch(batch, None, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=cap)
    print(deriv, n_seg, moving, api.kernel_trace())
    ref = po.solve_batch(batch.seg_offsets, batch.waypoints, batch.fixed_mask, batch.fixed_values, batch.limits,
                         np.zeros(batch.n_segments), deriv=deriv, time_alloc_method=2, estimate_times=True, sampling_dt=0.2,
                         sample_capacity=cap, n_threads=6)
    so = batch.seg_offsets
    for p in range(batch.n_paths):
        a, b = so[p], so[p + 1]
        dt = np.max(np.abs(out["times"][a:b] - ref["times"][a:b]) / ref["times"][a:b])
        print("  path %d: status %d / %d, times %.2e, coeffs %.2e, samples %d / %d, cost %.6e / %.6e, sumT %.3f / %.3f" % (
            p, out["status"][p], ref["status"][p], dt, util.coeff_error(out["coeffs"][a:b], ref["coeffs"][a:b]),
            out["n_samples"][p], ref["n_samples"][p], out["cost"][p], ref["cost"][p], out["times"][a:b].sum(), ref["times"][a:b].sum()))
