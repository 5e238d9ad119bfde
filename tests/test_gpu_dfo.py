"""GPU parity of the gradient-free time-allocation modes 0 / 1 (kSquaredTime / kRichterTime) and 3 / 4
(k*TimeAndConstraints) through the C ABI,
against oracle/mto_dfo.c.

Tolerances: the trial points of the search are exact functions of the start point and of the outcomes of the
comparisons f < f_best, so with the shipping budget (10 evaluations: only the initial interpolation sweep)
segment times agree to 1e-14 relative; coefficients to 1e-6 (SURVEY.md 8d metric; the reference-style
oracle's own error is <= 3e-8).  With a long budget a comparison can flip on the 1e-9 difference of the two
arithmetic routes; >= 90 % of the paths must agree to 1e-9 on the times, all must satisfy the invariants."""
import numpy as np
import pytest

from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po
from tests import util

pytestmark = pytest.mark.gpu


def _oracle(batch, mode, max_iterations, **kw):
    return po.solve_batch(batch.seg_offsets, batch.waypoints, batch.fixed_mask, batch.fixed_values, batch.limits,
                          np.zeros(batch.n_segments), deriv=batch.derivative_to_optimize, time_alloc_method=mode,
                          estimate_times=True, max_iterations=max_iterations, n_threads=8, **kw)


@pytest.mark.parametrize("mode", [api.TIME_ALLOC_SQUARED_TIME, api.TIME_ALLOC_RICHTER_TIME])
@pytest.mark.parametrize("n_seg,n_paths", [(10, 128), (3, 40), ("ragged", 64), (1, 5)])
def test_shipping_budget_matches_oracle(gpu_ctx, mode, n_seg, n_paths):
    batch = pr.random_batch(n_paths, n_seg, seed0=808)
    out = gpu_ctx.solve_batch(batch, None, time_alloc_method=mode)
    ref = _oracle(batch, mode, 10)
    assert np.array_equal(out["status"], ref["status"])
    if n_seg == 10:
        assert np.all(out["status"] == 5)  # 10 evaluations end inside the 2 S + 1 initial sweep
    assert np.max(np.abs(out["times"] - ref["times"]) / ref["times"]) < 1e-14
    assert util.coeff_error(out["coeffs"], ref["coeffs"], batch.seg_offsets) < 1e-6
    assert np.max(np.abs(out["cost"] - ref["cost"]) / np.abs(ref["cost"])) < 1e-6
    assert util.continuity_defect(batch, out["coeffs"], out["times"]) < 1e-9
    assert util.constraint_defect(batch, out["coeffs"], out["times"]) < 1e-9


@pytest.mark.parametrize("mode,soft", [(0, 1), (1, 1), (1, 0)])
def test_long_budget_matches_oracle(gpu_ctx, mode, soft):
    batch = pr.random_batch(48, 4, seed0=99)
    out = gpu_ctx.solve_batch(batch, None, time_alloc_method=mode, max_iterations=120, use_soft_constraints=soft,
                              time_penalty=20.0)
    ref = _oracle(batch, mode, 120, use_soft_constraints=soft, time_penalty=20.0)
    assert np.all(np.isin(out["status"], (3, 4, 5)))
    same = 0
    for p in range(batch.n_paths):
        a, b = batch.seg_offsets[p], batch.seg_offsets[p + 1]
        if out["status"][p] == ref["status"][p] and \
                np.max(np.abs(out["times"][a:b] - ref["times"][a:b]) / ref["times"][a:b]) < 1e-9:
            same += 1
    print("RATE dfo_long mode %d soft %d: %d / %d" % (mode, soft, same, batch.n_paths))
    assert same >= batch.n_paths - 1, same   # measured 48 / 48 in every mode: one comparison may flip on a 1e-9 difference of the solves
    assert np.all(out["times"] >= 0.01)
    assert util.continuity_defect(batch, out["coeffs"], out["times"]) < 1e-9


def test_start_below_lower_bound_reports_failure(gpu_ctx):
    batch = pr.random_batch(4, 5, seed0=1)
    t = util.oracle_times(batch)
    t[7] = 0.001
    out = gpu_ctx.solve_batch(batch, t, time_alloc_method=api.TIME_ALLOC_SQUARED_TIME)
    assert out["status"][1] == -1 and np.all(out["status"][[0, 2, 3]] == 5)
    assert np.array_equal(out["times"][5:10], t[5:10])


def test_unknown_mode_is_rejected(gpu_ctx):
    batch = pr.random_batch(2, 3, seed0=1)
    for mode in (-2, 5, 7):
        with pytest.raises(api.MrsTgError, match="time_alloc_method"):
            gpu_ctx.solve_batch(batch, None, time_alloc_method=mode)


# ---- modes 3 / 4: segment times and free end-point derivatives as variables ----
# The start point's free constraints come from two different linear solvers (oracle: reference-style dense QR,
# product: block Cholesky on exact constants), which agree to ~1e-9 of the end-point scale; the trial points
# inherit that difference, so times / coefficients are compared to 1e-6 and statuses exactly.

@pytest.mark.parametrize("mode", [api.TIME_ALLOC_SQUARED_TIME_AND_CONSTRAINTS, api.TIME_ALLOC_RICHTER_TIME_AND_CONSTRAINTS])
@pytest.mark.parametrize("n_seg,n_paths,budget", [(10, 64, 10), (3, 40, 10), ("ragged", 48, 10), (1, 5, 10), (4, 32, 30),
                                                  (4, 32, 45)])
def test_time_and_constraints_matches_oracle(gpu_ctx, mode, n_seg, n_paths, budget):
    batch = pr.random_batch(n_paths, n_seg, seed0=515)
    out = gpu_ctx.solve_batch(batch, None, time_alloc_method=mode, max_iterations=budget)
    ref = _oracle(batch, mode, budget)
    assert np.array_equal(out["status"], ref["status"])
    assert np.max(np.abs(out["times"] - ref["times"]) / ref["times"]) < 1e-9
    assert util.coeff_error(out["coeffs"], ref["coeffs"], batch.seg_offsets) < 1e-6
    assert np.max(np.abs(out["cost"] - ref["cost"]) / np.abs(ref["cost"])) < 1e-6
    assert util.continuity_defect(batch, out["coeffs"], out["times"]) < 1e-9
    assert util.constraint_defect(batch, out["coeffs"], out["times"]) < 1e-9


def test_time_and_constraints_with_acceleration_objective_and_stop_vertices(gpu_ctx):
    # derivative_to_optimize = 2 (the shipping default): jerk and snap are free at the ends, and the reference's
    # bound walk lands on other free constraints than intended (oracle/mto_dfo.c::mto_free_derivative_bounds)
    parts = []
    for s in range(24):
        wp = pr.random_box_waypoints(5, 900 + s)
        stop = np.zeros(6, dtype=bool)
        stop[2 + s % 2] = True
        parts.append(pr.build_vertices(wp, 2, stop_at=stop))
    batch = pr.assemble_batch(parts, np.tile(pr.DEFAULT_LIMITS, (len(parts), 1)), 2)
    for mode, budget in ((3, 10), (4, 40)):
        out = gpu_ctx.solve_batch(batch, None, time_alloc_method=mode, max_iterations=budget)
        ref = _oracle(batch, mode, budget)
        assert np.array_equal(out["status"], ref["status"])
        assert np.max(np.abs(out["times"] - ref["times"]) / ref["times"]) < 1e-9
        assert util.coeff_error(out["coeffs"], ref["coeffs"], batch.seg_offsets) < 1e-6
        assert util.continuity_defect(batch, out["coeffs"], out["times"]) < 1e-9


def test_time_and_constraints_long_budget(gpu_ctx):
    batch = pr.random_batch(24, 2, seed0=77)
    out = gpu_ctx.solve_batch(batch, None, time_alloc_method=4, max_iterations=200, time_penalty=20.0)
    ref = _oracle(batch, 4, 200, time_penalty=20.0)
    assert np.all(np.isin(out["status"], (3, 4, 5)))
    same = 0
    for p in range(batch.n_paths):
        a, b = batch.seg_offsets[p], batch.seg_offsets[p + 1]
        if out["status"][p] == ref["status"][p] and \
                np.max(np.abs(out["times"][a:b] - ref["times"][a:b]) / ref["times"][a:b]) < 1e-6 and \
                util.coeff_error(out["coeffs"][a:b], ref["coeffs"][a:b]) < 1e-5:
            same += 1
    print("RATE dfo_tc_long: %d / %d" % (same, batch.n_paths))
    assert same >= batch.n_paths - 1, same   # measured 24 / 24
    assert util.continuity_defect(batch, out["coeffs"], out["times"]) < 1e-9
