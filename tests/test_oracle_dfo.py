"""Oracle-side tests of the gradient-free time-allocation modes 0 / 1 (kSquaredTime / kRichterTime,
nonlinear_impl.h:121-157, 568-614, 725-762): the objective's three parts against independent numpy
arithmetic, and the invariants of the derivative-free search that stands in for NLopt's BOBYQA
(DESIGN.md 5b).  No GPU."""
import math

import numpy as np
import pytest

from mrs_uav_trajectory_generation_amd import problem as pr
from oracle import pyoracle as po
from tests import util


def _path(n_seg, seed):
    batch = pr.random_batch(1, n_seg, seed0=seed)
    t = util.oracle_times(batch)
    _, m, v = batch.path(0)
    return batch, m, v, t


def _dense_max(coeffs, times, k, n=4001):
    """max over the trajectory of the 4-D norm of derivative k, by dense sampling"""
    best = 0.0
    for s in range(len(times)):
        for x in np.linspace(0.0, times[s], n):
            best = max(best, float(np.linalg.norm(util.eval_poly(coeffs[s], x, k))))
    return best


@pytest.mark.parametrize("mode", [0, 1])
def test_objective_parts(mode):
    batch, m, v, t = _path(4, 11)
    lim = pr.DEFAULT_LIMITS
    f, parts = po.objective_time(4, m, v, t, lim, mode=mode, time_penalty=100.0, soft=1, weight=1.5)
    c = po.solve_linear(4, m, v, t)
    assert parts[0] == pytest.approx(po.compute_cost(4, t, c), rel=1e-13)
    total = float(np.sum(t))
    assert parts[1] == pytest.approx(100.0 * (total if mode == 1 else total * total), rel=1e-14)
    # soft constraints: 12 terms, every derivative's 4-D maximum against horizontal (x2), vertical, heading limits
    soft = 0.0
    for k in (1, 2, 3):
        mk = po.max_of_magnitude(c, t, k)
        dense = _dense_max(c, t, k, 801)
        assert mk >= dense * (1 - 1e-12) and mk <= dense * (1 + 1e-3)
        for grp, mult in ((0, 2), (1, 1), (2, 1)):
            value = lim[(k - 1) * 3 + grp]
            soft += mult * min(1e12, math.exp((mk - value) / value * 1.5))
    assert parts[2] == pytest.approx(soft, rel=1e-12)
    assert f == pytest.approx(parts.sum(), rel=1e-15)
    f0, parts0 = po.objective_time(4, m, v, t, lim, mode=mode, soft=0)
    assert parts0[2] == 0.0 and f0 == pytest.approx(parts[0] + parts[1], rel=1e-15)


def test_soft_constraint_cost_is_capped():
    _, m, v, t = _path(3, 5)
    lim = pr.DEFAULT_LIMITS * 1e-3  # 1000x violation: exp() overflows the cap
    _, parts = po.objective_time(4, m, v, t, lim, soft=1, weight=1.5)
    assert parts[2] == 12e12


def _python_search(fun, x0, lb, ub, budget, f_rel=0.05, x_rel=0.1):
    """DESIGN.md 5b restated in Python (an independent transcription of the specification, not of the C code): greedy
    coordinate search, h_i doubles on success and all h halve after a sweep without one, the last evaluation is the best
    point.  Returns (kept point, its value, evaluations, code)."""
    n = len(x0)
    h = 0.1 * np.abs(x0)
    h[np.abs(x0) <= np.finfo(float).eps] = 1e-13
    evals = [0]

    def relstop(a, b, tol):
        return abs(b - a) < tol * (abs(a) + abs(b)) * 0.5 or (tol > 0 and a == b)

    def f(x):
        evals[0] += 1
        return fun(x)
    best, fbest = np.array(x0, dtype=float), f(x0)
    last_is_best = True
    if budget == 1:
        return best, fbest, 1, 5

    def finish(code):   # the search ends on its best point
        if last_is_best and code != 5:
            return best, fbest, evals[0], code
        return best, f(best), evals[0], code
    while True:
        f_sweep, improved = fbest, False
        for i in range(n):
            acc_any = False
            for sg in (+1.0, -1.0):
                while True:
                    if evals[0] >= budget - 1:
                        return finish(5)
                    t = min(max(best[i] + sg * h[i], lb[i]), ub[i])
                    if t == best[i]:
                        break
                    x = best.copy()
                    x[i] = t
                    v = f(x)
                    if v < fbest - 1e-6 * abs(fbest):
                        best, fbest, improved, acc_any, last_is_best = x, v, True, True, True
                        h[i] *= 2.0
                        continue
                    last_is_best = False
                    break
                if acc_any:
                    break
        if improved:
            if relstop(f_sweep, fbest, f_rel):
                return finish(3)
        else:
            h *= 0.5
            if np.all(h < x_rel * np.abs(best)):
                return finish(4)


@pytest.mark.parametrize("n_seg,mode", [(1, 0), (3, 0), (3, 1), (10, 0), (10, 1)])
def test_dfo_follows_its_specification_evaluation_by_evaluation(n_seg, mode):
    """every budget from 1 to 40: the kept point, its value, the evaluation count and the code of the C search equal those of
    the Python transcription of DESIGN.md 5b; the kept point is the BEST evaluated point (the search's last evaluation
    revisits it), so its value never goes up with the budget"""
    _, m, v, t = _path(n_seg, 21 + n_seg)
    lim = pr.DEFAULT_LIMITS
    fun = lambda x: po.objective_time(4, m, v, x, lim, mode=mode)[0]  # noqa: E731
    f_prev = np.inf
    for budget in range(1, 41):
        rc, x, ne, fl = po.optimize_times_dfo(4, m, v, t, lim, mode=mode, max_iterations=budget)
        ex, ef, en, ec = _python_search(fun, t, np.full(n_seg, 0.01), np.full(n_seg, np.inf), budget)
        assert (rc, ne) == (ec, en), (budget, rc, ne, ec, en)
        assert np.array_equal(x, ex) and fl == ef
        assert fl == pytest.approx(fun(x), rel=1e-15)
        assert fl <= f_prev
        f_prev = fl
        if rc != 5:
            break
    # the first trial is x0 + 0.1 x0 e_0 (initial_stepsize_rel, nonlinear_impl.h:127-130)
    assert f_prev <= fun(t)


def test_dfo_converges_and_never_leaves_the_bounds():
    _, m, v, t = _path(3, 8)
    lim = pr.DEFAULT_LIMITS
    f_start = po.objective_time(4, m, v, t, lim, mode=1)[0]
    rc, x, ne, fl = po.optimize_times_dfo(4, m, v, t, lim, mode=1, max_iterations=400)
    assert rc in (3, 4) and ne < 400
    assert np.all(x >= 0.01)
    # a budget that is exactly what the search used ends on the same point (by the budget's rule then: code 5)
    rc2, x2, ne2, fl2 = po.optimize_times_dfo(4, m, v, t, lim, mode=1, max_iterations=ne)
    assert np.array_equal(x, x2) and fl == fl2
    assert fl < f_start


def test_dfo_rejects_start_below_lower_bound():
    _, m, v, t = _path(3, 8)
    t[1] = 0.005
    rc, x, ne, _ = po.optimize_times_dfo(4, m, v, t, pr.DEFAULT_LIMITS)
    assert rc == -2 and ne == 0 and np.array_equal(x, t)


def test_batch_driver_modes_0_1():
    batch = pr.random_batch(6, "ragged", seed0=3)
    for mode in (0, 1):
        out = po.solve_batch(batch.seg_offsets, batch.waypoints, batch.fixed_mask, batch.fixed_values, batch.limits,
                             np.zeros(batch.n_segments), deriv=4, time_alloc_method=mode, estimate_times=True)
        assert np.all(out["status"] == 5)
        assert util.continuity_defect(batch, out["coeffs"], out["times"]) < 1e-7
        assert util.constraint_defect(batch, out["coeffs"], out["times"]) < 1e-7


# ---- modes 3 / 4: segment times + free end-point derivatives (nonlinear_impl.h:429-536, 651-722, 765-804) ----

def test_free_constraints_round_trip():
    for n_seg, seed in ((1, 3), (4, 9), (10, 2)):
        _, m, v, t = _path(n_seg, seed)
        c, free = po.solve_linear_free(4, m, v, t)
        assert free.shape == (4, 4 * (n_seg - 1))
        assert np.array_equal(c, po.solve_linear(4, m, v, t))
        c2 = po.coeffs_from_free(4, m, v, t, free)
        assert util.coeff_error(c2, c) < 1e-13
        # the free constraints are the derivatives of the solution at the interior vertices
        for vert in range(1, n_seg):
            for k in range(1, 5):
                got = free[:, (vert - 1) * 4 + (k - 1)]
                assert np.allclose(got, math.factorial(k) * c[vert, :, k], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("mode", [3, 4])
def test_objective_with_free_constraints_equals_time_only_objective_at_the_linear_solution(mode):
    _, m, v, t = _path(5, 17)
    lim = pr.DEFAULT_LIMITS
    _, free = po.solve_linear_free(4, m, v, t)
    f, parts = po.objective_time_and_constraints(4, m, v, np.concatenate([t, free.ravel()]), lim, mode=mode)
    f01, parts01 = po.objective_time(4, m, v, t, lim, mode=mode - 3)
    assert np.allclose(parts, parts01, rtol=1e-9)
    # moving a free constraint away from the minimiser of J_d raises J_d
    x = np.concatenate([t, free.ravel()])
    x[len(t) + 3] += 0.05
    _, parts_moved = po.objective_time_and_constraints(4, m, v, x, lim, mode=mode)
    assert parts_moved[0] > parts[0] and parts_moved[1] == parts[1]


def test_free_derivative_bounds_snap_and_the_acceleration_quirk():
    lim = pr.DEFAULT_LIMITS  # v 2/2/1, a 2/2/2, j 20/20/20 (horizontal, vertical, heading)
    wp = pr.random_box_waypoints(3, 4)
    # derivative_to_optimize = 4: every free constraint (v, a, j, s per interior vertex) is walked, bounds aligned
    _, m, v = pr.build_vertices(wp, pr.SNAP)
    lo, hi = po.free_derivative_bounds(4, m, v, lim)
    big = np.finfo(float).max
    for dim, grp in enumerate((0, 0, 1, 2)):
        exp = [lim[grp], lim[3 + grp], lim[6 + grp], big] * 2
        assert np.array_equal(hi[dim], exp) and np.array_equal(lo[dim], -np.array(exp))
    # derivative_to_optimize = 2: ends fix p, v, a (jerk and snap free), the walk counts derivatives 0..2 only, so the
    # velocity / acceleration bounds of the two interior vertices land on free constraints 0..3 = (v0 jerk, v0 snap,
    # v1 velocity, v1 acceleration) instead of (v1 v, v1 a, v2 v, v2 a)
    _, m2, v2 = pr.build_vertices(wp, 2)
    assert m2.reshape(-1, 5).sum(axis=1).tolist() == [3, 1, 1, 3]
    lo2, hi2 = po.free_derivative_bounds(2, m2, v2, lim)
    assert hi2.shape == (4, 12)
    for dim, grp in enumerate((0, 0, 1, 2)):
        exp = [lim[grp], lim[3 + grp], lim[grp], lim[3 + grp]] + [big] * 8
        assert np.array_equal(hi2[dim], exp) and np.array_equal(lo2[dim], -np.array(exp))


@pytest.mark.parametrize("mode", [3, 4])
def test_dfo_time_and_constraints_follows_its_specification(mode):
    _, m, v, t = _path(4, 6)
    lim = pr.DEFAULT_LIMITS
    c0, free = po.solve_linear_free(4, m, v, t)
    rc, x, c, ne, fl = po.optimize_time_and_constraints_dfo(4, m, v, t, lim, mode=mode, max_iterations=1)
    assert rc == 5 and ne == 1 and np.array_equal(x, t) and util.coeff_error(c, c0) < 1e-13
    # variables [T, free derivatives of dimension 0, 1, 2, 3], bounds widened to the start (:496-501)
    x0 = np.concatenate([t, free.ravel()])
    lo, hi = po.free_derivative_bounds(4, m, v, lim)
    lb = np.concatenate([np.full(len(t), 0.01), np.minimum(lo.ravel(), free.ravel())])
    ub = np.concatenate([np.full(len(t), np.inf), np.maximum(hi.ravel(), free.ravel())])
    fun = lambda z: po.objective_time_and_constraints(4, m, v, z, lim, mode=mode)[0]  # noqa: E731
    # the first trial is T_0 + 10 % with the free constraints HELD (setFreeConstraints, not a re-solve)
    t1 = t.copy()
    t1[0] *= 1.1
    x1 = np.concatenate([t1, free.ravel()])
    rc, x, c, ne, fl = po.optimize_time_and_constraints_dfo(4, m, v, t, lim, mode=mode, max_iterations=3)
    assert rc == 5 and ne == 3
    if fun(x1) < fun(x0):
        assert np.allclose(x, t1, rtol=1e-15)
        assert util.coeff_error(c, po.coeffs_from_free(4, m, v, t1, free)) < 1e-13
        assert util.coeff_error(c, po.solve_linear(4, m, v, t1)) > 1e-6
    else:
        assert np.array_equal(x, t) and util.coeff_error(c, c0) < 1e-13
    # every budget up to 30 (inside the interpolation sweep: n = 52 variables): evaluation k >= 2 is x0 + h e_(k-2), the last
    # one goes back to the best of them
    h = 0.1 * np.abs(x0)
    h[np.abs(x0) <= np.finfo(float).eps] = 1e-13
    trials = [x0]
    for i in range(30):
        z = x0.copy()
        z[i] = min(max(x0[i] + h[i] if x0[i] + h[i] <= ub[i] else x0[i] - h[i], lb[i]), ub[i])
        trials.append(z)
    values = [fun(z) for z in trials]
    for budget in range(2, 31):
        rc, x, c, ne, fl = po.optimize_time_and_constraints_dfo(4, m, v, t, lim, mode=mode, max_iterations=budget)
        k = int(np.argmin(values[:budget - 1]))     # budget - 1 evaluations of the sweep (x0 included), then the revisit
        assert rc == 5 and ne == budget and np.array_equal(x, trials[k][:len(t)]) and fl == pytest.approx(values[k], rel=1e-13), budget
        assert util.coeff_error(c, po.coeffs_from_free(4, m, v, trials[k][:len(t)], trials[k][len(t):].reshape(4, -1))) < 1e-12


def test_dfo_time_and_constraints_long_run_improves_and_keeps_continuity():
    batch = pr.random_batch(1, 3, seed0=12)
    t = util.oracle_times(batch)
    _, m, v = batch.path(0)
    lim = pr.DEFAULT_LIMITS
    _, free = po.solve_linear_free(4, m, v, t)
    f0 = po.objective_time_and_constraints(4, m, v, np.concatenate([t, free.ravel()]), lim, mode=4)[0]
    fbest = min(po.optimize_time_and_constraints_dfo(4, m, v, t, lim, mode=4, max_iterations=k)[4] for k in (60, 90, 120))
    assert fbest < f0
    rc, x, c, ne, fl = po.optimize_time_and_constraints_dfo(4, m, v, t, lim, mode=4, max_iterations=120)
    assert np.all(x >= 0.01)
    assert util.continuity_defect(batch, c, x) < 1e-7 and util.constraint_defect(batch, c, x) < 1e-7


def test_batch_driver_modes_3_4():
    batch = pr.random_batch(5, "ragged", seed0=4)
    for mode in (3, 4):
        out = po.solve_batch(batch.seg_offsets, batch.waypoints, batch.fixed_mask, batch.fixed_values, batch.limits,
                             np.zeros(batch.n_segments), deriv=4, time_alloc_method=mode, estimate_times=True)
        assert np.all(out["status"] == 5)
        assert util.continuity_defect(batch, out["coeffs"], out["times"]) < 1e-7
        assert util.constraint_defect(batch, out["coeffs"], out["times"]) < 1e-7


def test_gradient_free_modes_run_on_every_arithmetic_route():
    """the oracle's other arithmetic routes (exact unit-time tables, 113-bit linear solve) under the gradient-free searches:
    same stopping reasons, times within the noise the reference-style route carries"""
    import numpy as np
    from mrs_uav_trajectory_generation_amd import problem as pr
    from tests import util
    batch = pr.random_batch(6, 4, seed0=321)
    t = util.oracle_times(batch)
    so = batch.seg_offsets
    for p in range(batch.n_paths):
        _, m, v = batch.path(p)
        tp = t[so[p]:so[p + 1]]
        base = {}
        for mode in (0, 1, 3):
            with po.arithmetic(po.REFERENCE_ARITHMETIC):
                base[mode] = (po.optimize_times_dfo(4, m, v, tp, batch.limits[p], mode=mode, max_iterations=12) if mode < 3 else
                              po.optimize_time_and_constraints_dfo(4, m, v, tp, batch.limits[p], mode=mode, max_iterations=12))
        for route in (po.EXACT_CONSTANTS, po.QUAD_PRECISION):
            with po.arithmetic(route):
                for mode in (0, 1, 3):
                    r = (po.optimize_times_dfo(4, m, v, tp, batch.limits[p], mode=mode, max_iterations=12) if mode < 3 else
                         po.optimize_time_and_constraints_dfo(4, m, v, tp, batch.limits[p], mode=mode, max_iterations=12))
                    assert r[0] == base[mode][0]   # stopping reason
                    assert np.max(np.abs(r[1] - base[mode][1]) / base[mode][1]) < 1e-6, (p, mode, route)   # times
    assert po.lib().mto_get_arithmetic() == 0
