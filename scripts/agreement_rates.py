"""Agreement rates of the nonlinear pipeline with the oracle on the batches the GPU tests use (to set their thresholds):
python scripts/agreement_rates.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrs_uav_trajectory_generation_amd import api, problem as pr  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from tests import util  # noqa: E402

ctx = api.Context(0)


def rates(name, batch, deriv=4, cap=1024):
    out = ctx.solve_batch(batch, None, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=cap)
    ref = po.solve_batch(batch.seg_offsets, batch.waypoints, batch.fixed_mask, batch.fixed_values, batch.limits,
                         np.zeros(batch.n_segments), deriv=deriv, time_alloc_method=2, runaway_rule=True, estimate_times=True,
                         sampling_dt=0.2, sample_capacity=cap, n_threads=os.cpu_count())
    so = batch.seg_offsets
    dt = np.array([np.max(np.abs(out["times"][a:b] - ref["times"][a:b]) / ref["times"][a:b]) for a, b in zip(so[:-1], so[1:])])
    dc = np.array([util.coeff_error(out["coeffs"][a:b], ref["coeffs"][a:b]) for a, b in zip(so[:-1], so[1:])])
    same = (out["status"] == ref["status"]) & (out["n_samples"] == np.minimum(ref["n_samples"], cap + 1))
    good = same & (dt < 1e-6) & (dc < 1e-6)
    print("%-34s paths %5d  status+samples equal %.4f  times<1e-6 %.4f  all (status, times, coeffs 1e-6) %.4f  times<1e-3 %.4f"
          % (name, batch.n_paths, same.mean(), (dt < 1e-6).mean(), good.mean(), (dt < 1e-3).mean()))


for n_seg, n_paths in [(10, 256), (3, 64), ("ragged", 96), (20, 24), (15, 32), (4, 16)]:
    rates("end_to_end %s x %d" % (n_seg, n_paths), pr.random_batch(n_paths, n_seg, seed0=4242))
for d in (2, 3, 4):
    rates("mixed d=%d" % d, pr.random_mixed_batch(768, d), deriv=d, cap=512)
rates("uniform 10 x 2048", pr.random_batch(2048, 10, seed0=0))
rates("ragged 1024", pr.random_batch(1024, "ragged", seed0=0))
