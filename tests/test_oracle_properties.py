"""Known-answer and invariant tests of the C oracle (no GPU): closed forms, structural invariants the
reference's construction guarantees (SURVEY.md 8c iii), the root finder, feasibility scaling, sampling,
estimators and the outer loop."""
import math

import numpy as np
import pytest

from mrs_uav_trajectory_generation_amd import problem as pr
from oracle import pyoracle as po
from tests import util


def test_closed_form_single_segment():
    wp = np.array([[0.0, 1.0, 2.0, 0.1], [3.0, -1.0, 4.0, 0.9]])
    T = 1.7
    _, m, v = pr.build_vertices(wp, pr.SNAP)
    c = po.solve_linear(4, m, v, [T])
    for dim in range(4):
        exp = np.zeros(10)
        exp[0] = wp[0, dim]
        for k, a in zip(range(5, 10), (126, -420, 540, -315, 70)):
            exp[k] = a * (wp[1, dim] - wp[0, dim]) / T ** k
        assert np.max(np.abs(c[0, dim] - exp)) <= 1e-11 * np.max(np.abs(exp))


def test_base_coefficients_table():
    L = po.lib()
    for r in range(6):
        for k in range(12):
            exp = math.factorial(k) // math.factorial(k - r) if k >= r else 0
            assert L.mto_base_coeff(r, k) == float(exp)


@pytest.mark.parametrize("n_seg", [1, 2, 3, 10, 30])
def test_continuity_and_constraints(n_seg):
    batch = pr.random_batch(6, n_seg, seed0=50 + n_seg)
    t = util.oracle_times(batch)
    out = util.oracle_linear(batch, t)
    assert util.continuity_defect(batch, out["coeffs"], t) < 1e-7
    assert util.constraint_defect(batch, out["coeffs"], t) < 1e-7
    assert np.all(out["status"] == 1)


def test_cost_equals_numeric_integral():
    batch = pr.random_batch(3, 5, seed0=7)
    t = util.oracle_times(batch)
    out = util.oracle_linear(batch, t)
    xs, ws = np.polynomial.legendre.leggauss(12)
    for p in range(3):
        a, b = batch.seg_offsets[p], batch.seg_offsets[p + 1]
        J = 0.0
        for s in range(a, b):
            tt = 0.5 * t[s] * (xs + 1.0)
            snap = np.stack([util.eval_poly(out["coeffs"][s], x, 4) for x in tt])
            J += 0.5 * t[s] * np.sum(ws[:, None] * snap ** 2)
        assert abs(J - out["cost"][p]) < 1e-8 * abs(J)


def test_jenkins_traub_against_numpy_roots():
    rng = np.random.default_rng(3)
    for deg in (2, 3, 5, 7, 11, 13, 15):
        for _ in range(20):
            c = rng.normal(size=deg + 1)
            r = po.find_roots(c)
            assert r.size == deg
            ref = np.roots(c[::-1])
            for z in r:
                assert np.min(np.abs(ref - z)) < 1e-6 * max(1.0, abs(z))


def test_jenkins_traub_edge_cases():
    assert po.find_roots([0.0, 0.0, 0.0]).size == 0          # all zero: no roots
    assert po.find_roots([3.0]).size == 0                     # constant
    r = po.find_roots([0.0, 0.0, 1.0, 1.0])                   # t^2 (t + 1): two zeros at the origin
    assert np.allclose(np.sort(r.real), [-1.0, 0.0, 0.0]) and np.allclose(r.imag, 0)
    r = po.find_roots([1.0, -2.0, 1.0, 0.0, 0.0])             # trailing (highest-power) zeros are stripped
    assert r.size == 2 and np.allclose(r.real, 1.0, atol=1e-6)
    r = po.find_roots([-6.0, 11.0, -6.0, 1.0])
    assert np.allclose(np.sort(r.real), [1.0, 2.0, 3.0])


def test_maxima_bound_sampled_maxima():
    # analytic maxima >= densely sampled maxima (test_utils.h:40-50 idea), and close to them
    batch = pr.random_batch(2, 6, seed0=91)
    t = util.oracle_times(batch)
    out = util.oracle_linear(batch, t)
    for s in range(batch.n_segments):
        tt = np.linspace(0.0, t[s], 2001)
        for k in (1, 2, 3):
            for grp in ([0, 1], [2], [3]):
                vals = np.stack([util.eval_poly(out["coeffs"][s, grp], x, k) for x in tt])
                sampled = np.max(np.linalg.norm(vals, axis=1))
                m = po.segment_max_magnitude(out["coeffs"][s], t[s], k, grp)
                assert m >= sampled * (1 - 1e-12) and m <= sampled * (1 + 1e-4) + 1e-12


def test_scaling_meets_limits_and_is_one_sweep():
    batch = pr.random_batch(8, 10, seed0=17)
    t = 0.5 * util.oracle_times(batch)          # too fast on purpose
    out = util.oracle_linear(batch, t)
    for p in range(batch.n_paths):
        a, b = batch.seg_offsets[p], batch.seg_offsets[p + 1]
        ok, c2, t2, sweeps = po.scale_segment_times(out["coeffs"][a:b], t[a:b], batch.limits[p])
        assert ok == 1 and sweeps == 1          # exact arithmetic needs one sweep
        assert np.all(t2 >= t[a:b] * (1 - 1e-15))
        for s in range(b - a):
            for k in (1, 2, 3):
                for gi, grp in enumerate(([0, 1], [2], [3])):
                    m = po.segment_max_magnitude(c2[s], t2[s], k, grp)
                    assert m <= batch.limits[p][3 * (k - 1) + gi] * 1.001
        # time scaling leaves the geometric path unchanged: same end points
        assert np.allclose(c2[:, :, 0], out["coeffs"][a:b, :, 0])


def test_sampling_semantics():
    batch = pr.config1_batch()
    t = np.array([5.0, 7.0710678118654755, 5.0])
    out = util.oracle_linear(batch, t)
    s, n = po.sample_trajectory(out["coeffs"], t, 0.2)
    # accumulate-and-carry: about ceil(sum T / dt) samples, first sample = first waypoint
    assert abs(n - math.ceil(t.sum() / 0.2)) <= 1
    assert np.allclose(s[0], pr.CONFIG1_WAYPOINTS[0], atol=1e-12)
    # samples lie on the polynomials: the k-th sample of segment 0 is p_0(0.2 k) while inside it
    for k in range(20):
        assert np.allclose(s[k], util.eval_poly(out["coeffs"][0], 0.2 * k), atol=1e-9)
    assert abs(po.wrap_yaw(4.0) - (4.0 - 2 * math.pi)) < 1e-12 and abs(po.wrap_yaw(-3.0) + 3.0) < 1e-12


def test_segment_time_estimators():
    lim = pr.DEFAULT_LIMITS
    t = po.estimate_times(pr.CONFIG1_WAYPOINTS, lim)
    assert np.allclose(t, [5.0, math.hypot(10, 10) / 2.0, 5.0])
    # vertical move uses the vertical speed limit; pure heading change is bounded below by the heading time
    wp = np.array([[0, 0, 0, 0.0], [0, 0, 4.0, 0.0], [0, 0, 4.0, 3.0]])
    t = po.estimate_times(wp, lim)
    assert abs(t[0] - 2.0) < 1e-12
    assert abs(t[1] - 1.5 * ((3.0 - 0.5) / 1.0 + 1.0)) < 1e-12
    relaxed = lim.copy()
    relaxed[[2, 5, 8]] = np.finfo(np.float32).max
    assert abs(po.estimate_times(wp, relaxed)[1] - 0.01) < 1e-15   # relax_heading: only the 0.01 floor remains
    tb = po.estimate_times(pr.CONFIG1_WAYPOINTS, lim, baca=True)
    assert np.all(tb >= t.min()) and np.all(np.isfinite(tb))
    assert abs(po.unwrap_heading(3.5, -3.0) - (3.5 - 2 * math.pi)) < 1e-12


def test_outer_loop_reduces_cost_and_honours_budget():
    batch = pr.random_batch(6, 10, seed0=23)
    for p in range(batch.n_paths):
        wp, m, v = batch.path(p)
        t0 = po.estimate_times(wp, batch.limits[p])
        J0, _ = po.cost_and_gradient(4, m, v, t0)
        rc, t1, ne, f_last = po.optimize_times(4, m, v, t0)
        assert rc in (1, 3, 4, 5) and 1 <= ne <= 10
        assert np.all(t1 >= 0.01)
        if rc in (3, 4):                      # stopped on an accepted step: monotone decrease
            assert f_last <= J0 * (1 + 1e-12)
    # a start below the lower bound is rejected like NLopt does
    rc, _, ne, _ = po.optimize_times(4, m, v, np.full(10, 0.001))
    assert rc == -2 and ne == 0
    # single segment: zero gradient, immediate success (nonlinear_impl.h:264-271)
    wp1, m1, v1 = pr.build_vertices(pr.random_box_waypoints(1, 3), 4)
    rc, t1, ne, _ = po.optimize_times(4, m1, v1, [2.0])
    assert rc == 1 and ne == 1 and t1[0] == 2.0


def test_batch_driver_threads_agree():
    batch = pr.random_batch(16, "ragged", seed0=3)
    a = po.solve_batch(batch.seg_offsets, batch.waypoints, batch.fixed_mask, batch.fixed_values, batch.limits,
                       np.zeros(batch.n_segments), estimate_times=True, time_alloc_method=2, sampling_dt=0.2,
                       sample_capacity=64, n_threads=1)
    b = po.solve_batch(batch.seg_offsets, batch.waypoints, batch.fixed_mask, batch.fixed_values, batch.limits,
                       np.zeros(batch.n_segments), estimate_times=True, time_alloc_method=2, sampling_dt=0.2,
                       sample_capacity=64, n_threads=4)
    for k in ("times", "coeffs", "status", "cost", "n_samples", "samples"):
        assert np.array_equal(a[k], b[k]), k


def _textbook_min_derivative_qp(waypoints, mask, vals, times, deriv):
    """Independent route to the same minimiser: the equality-constrained QP in COEFFICIENT space,
        min  sum_seg 1/2 c^T Q_seg c   s.t.  p^(k)(vertex) = value for every constrained (vertex, k),
                                            p_i^(k)(T_i) = p_{i+1}^(k)(0), k = 0..4, at every interior vertex,
    solved by the null-space method with numpy / scipy (no mapping matrices, no reordering, no block elimination)."""
    from math import factorial
    from scipy.linalg import null_space
    S, n = len(times), 10
    def row(t, k):   # d^k/dt^k of [1, t, ..., t^9]
        r = np.zeros(n)
        for j in range(k, n):
            r[j] = factorial(j) / factorial(j - k) * t ** (j - k)
        return r
    Q = np.zeros((S * n, S * n))
    for s, T in enumerate(times):
        for i in range(deriv, n):
            for j in range(deriv, n):
                e = i + j - 2 * deriv + 1
                Q[s * n + i, s * n + j] = factorial(i) / factorial(i - deriv) * factorial(j) / factorial(j - deriv) * T ** e / e
    sol = np.zeros((S, 4, n))
    for dim in range(4):
        rows, rhs = [], []
        for v in range(S + 1):
            for k in range(5):
                seg, t = (v, 0.0) if v < S else (S - 1, times[S - 1])
                if mask[v, k]:
                    r = np.zeros(S * n)
                    r[seg * n:(seg + 1) * n] = row(t, k)
                    rows.append(r)
                    rhs.append(vals[v, k, dim])
                if 0 < v < S:   # continuity between segment v-1 (end) and segment v (start)
                    r = np.zeros(S * n)
                    r[(v - 1) * n:v * n] = row(times[v - 1], k)
                    r[v * n:(v + 1) * n] = -row(0.0, k)
                    rows.append(r)
                    rhs.append(0.0)
        # null-space method (constraints met to rounding, then an SPD reduced system): c = c_p + N z
        E = np.array(rows)
        r = np.array(rhs)
        c_p = np.linalg.lstsq(E, r, rcond=None)[0]
        N = null_space(E)
        z = np.linalg.solve(N.T @ Q @ N, -N.T @ (Q @ c_p))
        x = c_p + N @ z
        sol[:, dim, :] = x[:S * n].reshape(S, n)
    return sol


@pytest.mark.parametrize("n_seg,deriv,seed", [(3, 4, 1), (4, 4, 2), (3, 2, 3), (5, 3, 4), (2, 4, 5)])
def test_linear_solution_equals_the_textbook_kkt_solution(n_seg, deriv, seed):
    """the reference's route (end-point derivative variables, A^-1, reordering, reduced system) and the textbook
    coefficient-space QP describe the same minimiser; this pins the oracle's algebra to the problem statement
    (Richter et al., ISRR 2013) rather than to its own formulas"""
    rng = np.random.default_rng(seed)
    wp = pr.random_box_waypoints(n_seg, 600 + seed)
    stop = np.zeros(n_seg + 1, dtype=bool)
    if n_seg >= 4:
        stop[2] = True
    _, m, v = pr.build_vertices(wp, deriv, stop_at=stop)
    t = rng.uniform(0.8, 2.5, n_seg)
    c = po.solve_linear(deriv, m, v, t)
    ref = _textbook_min_derivative_qp(wp, m.reshape(-1, 5), v.reshape(-1, 5, 4), t, deriv)
    # minimum acceleration with ten coefficients is nearly degenerate (smallest eigenvalue of R_pp ~ 6e-6): the
    # two routes agree on the cost to 1e-9 but on the coefficients only to cond * eps (minimum jerk sits in between)
    assert util.coeff_error(c, ref) < {2: 1e-5, 3: 1e-7, 4: 1e-9}[deriv]
    J = po.compute_cost(deriv, t, c)
    Jref = po.compute_cost(deriv, t, ref)
    assert abs(J - Jref) <= 1e-8 * abs(Jref)


def test_sample_count_is_a_property_of_dt_and_the_total_time_only():
    """The premise of the HIP sampler's accumulated-time table (mrs_tg_sampling.hpp, sample_acc_table): the reference's
    `accumulated_time += dt` runs from 0 whatever the segments are, so the number of samples of a trajectory is
    #{k : A[k] < t_end} with A[k] = k sequential additions of dt and t_end the sequential sum of the segment times -- checked
    against the oracle's walk (Trajectory::evaluateRange restated) on trajectories cut into segments in different ways,
    including totals that are multiples of dt, where the count hangs on the rounding of the accumulation."""
    rng = np.random.default_rng(5)
    for dt in (0.2, 0.1, 0.05, 0.3):
        acc = [0.0]
        for _ in range(6000):
            acc.append(acc[-1] + dt)
        acc = np.array(acc)
        for trial in range(40):
            n_seg = int(rng.integers(1, 9))
            if trial % 4 == 0:    # a total that is a multiple of dt (in exact arithmetic)
                times = np.full(n_seg, dt * int(rng.integers(3, 40)))
            else:
                times = rng.uniform(0.05, 12.0, n_seg)
            coeffs = rng.standard_normal((n_seg, 4, 10))
            t_end = 0.0
            for t in times:
                t_end += float(t)
            expect = int(np.searchsorted(acc, t_end, side="left"))   # the first k with A[k] >= t_end
            _, n = po.sample_trajectory(coeffs, times, dt, capacity=8192)
            # (the carry into the next segment can run off the last segment one sample early: the reference's own break)
            assert n in (expect, expect - 1), (dt, times, n, expect)
            if n == expect - 1:
                assert abs(acc[expect - 1] - t_end) < 1e-9 * max(1.0, t_end)


def test_band_limited_qr_is_the_dense_qr_bit_for_bit():
    """oracle/mto_linear.c qr_solve: R_pp is block-tridiagonal and holds exact zeros outside its band; the loops limited to the
    window in which operands can be non-zero leave out only operations on exact zeros.  Same coefficients, costs, times and
    samples to the last bit as the full dense loops, in the reference's arithmetic, with the exact tables and in 113 bits --
    on min-snap / min-acceleration paths, stop_at vertices, moving starts (variable block sizes) and through the Mellinger loop."""
    L = po.lib()
    rng = np.random.default_rng(3)
    parts = []
    for seed, (n_seg, deriv) in enumerate([(3, 4), (10, 4), (17, 2), (30, 3), (45, 4), (64, 2)]):
        wp = pr.random_box_waypoints(n_seg, 900 + seed)
        stop = [(0 < i < n_seg) and (i % 3 == 0) and seed % 2 == 1 for i in range(n_seg + 1)]
        init = None
        if seed % 3 == 2:
            init = dict(heading=wp[0, 3], velocity=np.append(rng.uniform(-1, 1, 3), 0.1), acceleration=np.zeros(4), jerk=np.zeros(4))
        parts.append((deriv, pr.build_vertices(wp, deriv, stop_at=stop, initial_state=init)))
    try:
        for mode in (po.REFERENCE_ARITHMETIC, po.EXACT_CONSTANTS, po.QUAD_PRECISION):
            L.mto_set_arithmetic(mode)
            for deriv, part in parts:
                batch = pr.assemble_batch([part], pr.DEFAULT_LIMITS[None, :], deriv)
                res = []
                for dense in (1, 0):
                    L.mto_set_dense_qr(dense)
                    res.append(po.solve_batch(batch.seg_offsets, batch.waypoints, batch.fixed_mask, batch.fixed_values, batch.limits,
                                              np.zeros(batch.n_segments), deriv=deriv, estimate_times=True,
                                              time_alloc_method=(2 if mode != po.QUAD_PRECISION else -1), sampling_dt=0.2,
                                              sample_capacity=1024))
                for k in ("coeffs", "times", "status", "cost", "n_samples", "samples"):
                    assert np.array_equal(res[0][k], res[1][k]), (mode, deriv, batch.n_segments, k)
    finally:
        L.mto_set_arithmetic(po.REFERENCE_ARITHMETIC)
        L.mto_set_dense_qr(0)


def test_oracle_follows_the_product_to_256_segments():
    """MTO_MAX_SEG = MRS_TG_MAX_SEGMENTS: a 256-segment path solves, continuously and on its constraints, in both arithmetic
    routes, which agree with each other as far as the double route's own rounding goes"""
    batch = pr.random_batch(1, 256, seed0=8200)
    t = util.oracle_times(batch)
    ref = util.oracle_linear(batch, t)
    assert ref["status"][0] == 1
    assert util.continuity_defect(batch, ref["coeffs"], t) < 1e-6 and util.constraint_defect(batch, ref["coeffs"], t) < 1e-6
    po.lib().mto_set_arithmetic(po.QUAD_PRECISION)
    try:
        refq = util.oracle_linear(batch, t)
    finally:
        po.lib().mto_set_arithmetic(po.REFERENCE_ARITHMETIC)
    assert util.continuity_defect(batch, refq["coeffs"], t) < 1e-9
    assert util.coeff_error(ref["coeffs"], refq["coeffs"], batch.seg_offsets) < 1e-6
    assert po.solve_batch(pr.random_batch(1, 257, seed0=1).seg_offsets, *[getattr(pr.random_batch(1, 257, seed0=1), k) for k in
                          ("waypoints", "fixed_mask", "fixed_values", "limits")], np.ones(257))["status"][0] < 0
