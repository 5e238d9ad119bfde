"""Shared helpers of the test-suite (tests may use the oracle; the product package may not)."""
import numpy as np

from mrs_uav_trajectory_generation_amd import problem as pr
from oracle import pyoracle as po


def coeff_error(c, ref, seg_offsets=None):
    """SURVEY.md 8d error metric: per path max|c - ref| / max|ref|, maximised over the batch."""
    c = np.asarray(c)
    ref = np.asarray(ref)
    if seg_offsets is None:
        return float(np.max(np.abs(c - ref)) / np.max(np.abs(ref)))
    worst = 0.0
    for p in range(len(seg_offsets) - 1):
        a, b = seg_offsets[p], seg_offsets[p + 1]
        worst = max(worst, float(np.max(np.abs(c[a:b] - ref[a:b])) / np.max(np.abs(ref[a:b]))))
    return worst


def case_arrays(case):
    return (case["derivative_to_optimize"], np.array(case["fixed_mask"], dtype=np.uint8),
            np.array(case["fixed_values"], dtype=np.float64), np.array(case["seg_times"], dtype=np.float64),
            np.array(case["waypoints"], dtype=np.float64))


def case_batch(case):
    d, m, v, t, wp = case_arrays(case)
    return pr.assemble_batch([(wp, m, v)], pr.DEFAULT_LIMITS[None], d), t


def oracle_times(batch):
    """Euclidean initial times for every path (oracle estimator)."""
    out = np.zeros(batch.n_segments)
    for p in range(batch.n_paths):
        wp, _, _ = batch.path(p)
        out[batch.seg_offsets[p]:batch.seg_offsets[p + 1]] = po.estimate_times(wp, batch.limits[p])
    return out


def oracle_linear(batch, times):
    return po.solve_batch(batch.seg_offsets, batch.waypoints, batch.fixed_mask, batch.fixed_values, batch.limits,
                          times, deriv=batch.derivative_to_optimize)


def eval_poly(c, t, deriv=0):
    """c: [..., 10] ascending; derivative `deriv` at t."""
    from math import factorial
    n = c.shape[-1]
    acc = np.zeros(c.shape[:-1])
    for k in range(n - 1, deriv - 1, -1):
        acc = acc * t + factorial(k) / factorial(k - deriv) * c[..., k]
    return acc


def _endpoint_scale(c, T):
    """|| [T^r p^(r)(0), T^r p^(r)(T)]_{r=0..4} ||_inf per dimension: the size of the segment's end-point
    derivative vector in normalised time.  Recovering end-point derivatives from monomial coefficients is
    a round trip through the mapping matrix A (cond ~ 4e6 at unit time), so defects
    are meaningful relative to this scale (bound ~ cond * eps ~ 1e-9), not relative to the single value."""
    sc = np.zeros(c.shape[:-1])
    for r in range(5):
        sc = np.maximum(sc, np.abs(eval_poly(c, 0.0, r)) * T ** r)
        sc = np.maximum(sc, np.abs(eval_poly(c, T, r)) * T ** r)
    return np.maximum(sc, 1e-300)


def continuity_defect(batch, coeffs, times, paths=None):
    """max over interior vertices / derivatives 0..4 of T^k |p_i^(k)(T_i) - p_{i+1}^(k)(0)| / end-point scale"""
    worst = 0.0
    for p in (range(batch.n_paths) if paths is None else paths):
        a, b = int(batch.seg_offsets[p]), int(batch.seg_offsets[p + 1])
        for s in range(a, b - 1):
            T = times[s]
            scale = _endpoint_scale(coeffs[s], T)
            for k in range(5):
                end = eval_poly(coeffs[s], T, k)
                start = eval_poly(coeffs[s + 1], 0.0, k)
                worst = max(worst, float(np.max(np.abs(end - start) * T ** k / scale)))
    return worst


def constraint_defect(batch, coeffs, times, paths=None):
    """max violation of the fixed vertex constraints by the polynomials (same normalisation)."""
    worst = 0.0
    for p in (range(batch.n_paths) if paths is None else paths):
        a, b = int(batch.seg_offsets[p]), int(batch.seg_offsets[p + 1])
        wp, m, v = batch.path(p)
        S = b - a
        for vert in range(S + 1):
            s = a + min(vert, S - 1)
            T = times[s]
            scale = _endpoint_scale(coeffs[s], T)
            for k in range(5):
                if not m[vert, k]:
                    continue
                val = eval_poly(coeffs[s], 0.0 if vert < S else T, k)
                worst = max(worst, float(np.max(np.abs(val - v[vert, k]) * T ** k / scale)))
    return worst
