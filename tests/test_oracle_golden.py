"""The C oracle against the 60-digit ground-truth fixtures (tests/golden, made by oracle/gen_golden.py).

The oracle follows the reference's arithmetic route, whose own error is what these bounds record
(SURVEY.md 8c measured 5e-12 .. 2e-10 for T in [0.05, 20] s with snap; acceleration cost and very short
segments are worse).  The bounds are per-case ceilings with head-room, not targets.
"""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import util


def _tol_coeff(case):
    if "short" in case["name"]:
        return 2e-7
    return {2: 5e-8, 3: 1e-9, 4: 1e-9}[case["derivative_to_optimize"]]


def test_oracle_coefficients_and_cost(golden):
    assert len(golden["cases"]) >= 10
    for case in golden["cases"]:
        d, m, v, t, _ = util.case_arrays(case)
        c = po.solve_linear(d, m, v, t)
        exact = np.array(case["coeffs"])
        assert util.coeff_error(c, exact) < _tol_coeff(case), case["name"]
        J = po.compute_cost(d, t, c)
        assert abs(J - case["cost"]) <= 1e-9 * abs(case["cost"]), case["name"]


def test_oracle_segment_blocks(golden):
    for case in golden["cases"]:
        if "H" not in case:
            continue
        d, _, _, t, _ = util.case_arrays(case)
        for s, T in enumerate(t):
            H, Ai = po.segment_hessian(d, T)
            He, Ae = np.array(case["H"][s]), np.array(case["Ainv"][s])
            assert np.max(np.abs(H - He)) <= 2e-7 * np.max(np.abs(He)), (case["name"], s)
            assert np.max(np.abs(Ai - Ae)) <= 2e-7 * np.max(np.abs(Ae)), (case["name"], s)


def test_oracle_mellinger_gradient(golden):
    n = 0
    for case in golden["cases"]:
        if "gradient" not in case:
            continue
        n += 1
        d, m, v, t, _ = util.case_arrays(case)
        J, g = po.cost_and_gradient(d, m, v, t)
        ge = np.array(case["gradient"])
        assert abs(J - case["cost"]) <= 1e-9 * abs(case["cost"])
        # forward difference of two ~1e-10-accurate costs over h = 0.1
        assert np.max(np.abs(g - ge)) <= 1e-6 * np.max(np.abs(ge)), case["name"]
    assert n >= 4


def test_survey_probe_values_config1(golden):
    # config 1 with Euclidean times [5, 7.0711, 5]: J_d = 20.4602, grad = [-2.2572, 5.6827, -2.2572]
    case = next(c for c in golden["cases"] if c["name"] == "config1_snap")
    assert abs(case["cost"] - 20.4602) < 1e-4
    assert np.allclose(case["gradient"], [-2.2572, 5.6827, -2.2572], atol=1e-4)


def test_oracle_maxima(golden):
    groups = [[0, 1], [2], [3]]
    n = 0
    for case in golden["cases"]:
        if "maxima" not in case:
            continue
        n += 1
        d, m, v, t, _ = util.case_arrays(case)
        exact_c = np.array(case["coeffs"])
        mx = np.array(case["maxima"])
        for s in range(len(t)):
            for k in (1, 2, 3):
                for gi, grp in enumerate(groups):
                    val = po.segment_max_magnitude(exact_c[s], t[s], k, grp)
                    assert abs(val - mx[s, k - 1, gi]) <= 1e-10 * max(mx[s, k - 1, gi], 1e-6), (case["name"], s, k, gi)
    assert n >= 2
