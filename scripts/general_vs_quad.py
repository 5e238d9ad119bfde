"""Paths with position-free vertices (MRS_TG_FLAG_GENERAL_PATTERNS) against the oracle's reference-style and 113-bit routes:
the linear solve at fixed times, and the Mellinger pipeline.  usage (GPU box): python scripts/general_vs_quad.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po
from tests import util
from tests.test_gpu_general_patterns import _free_some_positions

ctx = api.Context(0)
for n_seg, d in (("ragged", 4), (8, 2), (6, 4), ("ragged", 3)):
    rng = np.random.default_rng(7 + d)
    base = pr.random_batch(96, n_seg, seed0=1300, derivative_to_optimize=d)
    batch, touched = _free_some_positions(base, rng, share=0.15)
    tm = np.zeros(batch.n_paths, bool)
    tm[touched] = True
    so = batch.seg_offsets
    t = util.oracle_times(base)
    lin = ctx.solve_batch(batch, t)
    for mode in (0, 2):
        po.lib().mto_set_arithmetic(mode)
        ref = po.solve_batch(batch.seg_offsets, batch.waypoints, batch.fixed_mask, batch.fixed_values, batch.limits, t, deriv=d,
                             n_threads=16)
        po.lib().mto_set_arithmetic(0)
        e = np.array([util.coeff_error(lin["coeffs"][a:b], ref["coeffs"][a:b]) for a, b in zip(so[:-1], so[1:])])
        ej = np.abs(lin["cost"] - ref["cost"]) / np.abs(ref["cost"])
        print("%s d=%d LINEAR vs oracle route %d: touched coeff err max %.1e median %.1e, cost err max %.1e | others coeff max %.1e"
              % (n_seg, d, mode, e[tm].max(), np.median(e[tm]), ej[tm].max(), e[~tm].max()))
    out = ctx.solve_batch(batch, None, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=1024)
    for mode in (0, 2):
        po.lib().mto_set_arithmetic(mode)
        ref = po.solve_batch(batch.seg_offsets, batch.waypoints, batch.fixed_mask, batch.fixed_values, batch.limits,
                             np.zeros(batch.n_segments), deriv=d, time_alloc_method=2, runaway_rule=True, estimate_times=True, sampling_dt=0.2,
                             sample_capacity=1024, n_threads=16)
        po.lib().mto_set_arithmetic(0)
        dt = np.array([np.max(np.abs(out["times"][a:b] - ref["times"][a:b]) / ref["times"][a:b]) for a, b in zip(so[:-1], so[1:])])
        same = out["status"] == ref["status"]
        print("%s d=%d MELLINGER vs oracle route %d: touched: status equal %d/%d, dt<1e-6 %d/%d, worst %.1e | others dt<1e-6 %d/%d"
              % (n_seg, d, mode, same[tm].sum(), tm.sum(), (dt[tm] < 1e-6).sum(), tm.sum(), dt[tm].max(), (dt[~tm] < 1e-6).sum(), (~tm).sum()))
        if mode == 2:
            for p in np.nonzero(tm & ((dt > 1e-6) | ~same))[0]:
                a, b = so[p], so[p + 1]
                print("    path %d S=%d: status gpu %d oracle %d, dt %.1e, gpu max T %.3g, oracle max T %.3g, min T %.3g"
                      % (p, b - a, out["status"][p], ref["status"][p], dt[p], out["times"][a:b].max(), ref["times"][a:b].max(),
                         ref["times"][a:b].min()))
