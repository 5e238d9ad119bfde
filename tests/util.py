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


ROUNDOFF_LIMITED = -4   # MRS_TG_STATUS_ROUNDOFF_LIMITED, the product's own status word (include/mrs_tg.h)


def status_matches(out_status, ref_status):
    """Statuses of the HIP path against the oracle run on the REFERENCE's rule (pyoracle.solve_batch's default): equal, except
    on the paths the product hands back as ROUNDOFF_LIMITED (-4: its feasibility scaling ran away) -- there the reference
    reports the outer loop's own code (>= 1) and its nodelet discards the trajectory one step later by the length check
    (src/mrs_trajectory_generation.cpp:1178-1199).  The -4 set itself is asserted separately (runaway_sets_agree).
    Works elementwise on arrays and on scalars."""
    out_status, ref_status = np.asarray(out_status), np.asarray(ref_status)
    return (out_status == ref_status) | ((out_status == ROUNDOFF_LIMITED) & (ref_status >= 1))


def runaway_sets_agree(batch, out, ref, start_times=None, factor=25.0):
    """The product's deviation made visible: the paths it flags -4 are exactly the paths on which the ORACLE's own result
    (reference rule: a success code) has a total time of more than `factor` times the total the search started from.
    start_times: the start point (None = the Euclidean estimate, estimate_times).  Paths whose ratio lies within 1e-6 of the
    threshold could fall either way and are skipped.  Returns the number of flagged paths."""
    so = batch.seg_offsets
    t0 = oracle_times(batch) if start_times is None else np.asarray(start_times)
    ratio = np.add.reduceat(np.asarray(ref["times"]), so[:-1]) / np.add.reduceat(t0, so[:-1])
    flagged = np.asarray(out["status"]) == ROUNDOFF_LIMITED
    clear = np.abs(ratio / factor - 1.0) > 1e-6
    assert np.array_equal(flagged[clear], (ratio > factor)[clear] & (np.asarray(ref["status"])[clear] >= 1)), \
        (np.nonzero(flagged)[0][:8], np.nonzero(ratio > factor)[0][:8])
    return int(flagged.sum())
