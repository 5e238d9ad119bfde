"""Row a20 (optimizeTimeMellingerOuterLoop): how GOOD is the project's own projected L-BFGS?

NLopt's LD_LBFGS is not vendored and its iterates are not reproduced; the kernel and the oracle run the
same own search, so "GPU == oracle" says nothing about its quality.  tests/golden/optimizer_quality.json holds what an
optimiser that shares no code with either -- scipy's L-BFGS-B, same objective, same h = 0.1 forward-difference "gradient",
same bounds, best point of its first 10 objective evaluations -- reaches on 2 x 200 seeded 10-segment paths
(tests/golden/gen_optimizer_quality.py).

The quantity compared is the scale-invariant cost  J_d * (sum T)^(2d-1):  J_d(a T) = a^(1-2d) J_d(T), and the Mellinger
"gradient" is taken along directions that keep the total time constant (nonlinear_impl.h:281-322), so J_d on its own can be
made arbitrarily small by stretching the trajectory -- which the feasibility scaling that follows undoes.  What the outer
loop is there for is the DISTRIBUTION of the time over the segments, and J_d (sum T)^7 measures exactly that.

Stated factors (measured in the build container, asserted with ~10 % slack):
  * without the early-stop tolerances (f_rel = x_rel = 0: both searches use their 10 evaluations) the own search is on par:
    median own / scipy 0.99 (box paths) and 1.12 (random-walk paths);
  * with the reference's tolerances (ftol_rel 0.05, xtol_rel 0.1, src/mrs_trajectory_generation.cpp:884-885) it stops after
    3.6 / 5.5 evaluations on average and ends at a median 1.26 / 1.75 of scipy's 10-evaluation result, while still taking
    the scale-invariant cost down to 0.39 / 0.17 of the Euclidean start;
  * LIKE FOR LIKE (round 3): scipy's L-BFGS-B under the SAME stopping rule (NLopt's relstop with ftol_rel 0.05 / xtol_rel 0.1
    on successive accepted iterates, 10 evaluations at most) stops after 3.7 / 5.6 evaluations -- the early stop belongs to the
    reference's tolerances, not to the search -- and the own search ends at a median 0.87 (box) / 1.05 (random walk) of what
    scipy reaches under that rule (p90 1.01 / 2.03): the gap of the previous item is the stopping rule's.
"""
import json
import os

import numpy as np
import pytest

from mrs_uav_trajectory_generation_amd import problem as pr
from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def quality():
    with open(os.path.join(ROOT, "tests", "golden", "optimizer_quality.json")) as f:
        return json.load(f)


def _normalised(J, times, d=4):
    return J * np.sum(times) ** (2 * d - 1)


def _reference(quality, gen):
    s = [x for x in quality["sets"] if x["generator"] == gen][0]
    batch = pr.random_batch(quality["n_paths"], quality["n_segments"], seed0=quality["seed0"], generator=gen)
    sc = np.array([r["J_scipy"] * r["sum_T_scipy"] ** 7 for r in s["paths"]])
    st = np.array([r["J_start"] * r["sum_T_start"] ** 7 for r in s["paths"]])
    return batch, sc, st


# (generator, tolerances) -> bounds on: median own/scipy, 90th percentile own/scipy, median own/start, share of the paths
# that end below their start (the point kept is the LAST EVALUATED one, accepted or not -- the reference's semantics,
# nonlinear_impl.h:212-213 -- so a search that spends its whole budget can end on a rejected trial)
BOUNDS = {("box", "reference"): (1.40, 2.8, 0.43, 0.93), ("box", "none"): (1.03, 1.05, 0.33, 0.93),
          ("walk", "reference"): (1.95, 10.5, 0.19, 0.93), ("walk", "none"): (1.25, None, 0.14, 0.85)}


@pytest.mark.parametrize("gen,tol", sorted(BOUNDS))
def test_own_lbfgs_against_scipy_lbfgsb_on_the_oracle(quality, gen, tol):
    batch, sc, st = _reference(quality, gen)
    own = np.zeros(batch.n_paths)
    for p in range(batch.n_paths):
        wp, m, v = batch.path(p)
        prm = po.default_nlopt()
        if tol == "none":
            prm.f_rel = 0.0
            prm.x_rel = 0.0
        _, t, _, _ = po.optimize_times(4, m, v, po.estimate_times(wp, batch.limits[p]), prm)
        J, _ = po.cost_and_gradient(4, m, v, t)
        own[p] = _normalised(J, t)
    med, p90, start, helped = BOUNDS[(gen, tol)]
    r = own / sc
    assert np.median(r) <= med, np.median(r)
    if p90 is not None:
        assert np.percentile(r, 90) <= p90, np.percentile(r, 90)
    assert np.median(own / st) <= start, np.median(own / st)
    assert np.mean(own < st) >= helped, np.mean(own < st)


# generator -> bounds on median and 90th percentile of own / scipy-under-the-same-stopping-rule (measured 0.865, 1.013 and
# 1.054, 2.029), and on the mean number of evaluations either search spends (measured 3.63 vs 3.69 and 5.49 vs 5.55)
SAME_RULE = {"box": (0.95, 1.10, 4.0), "walk": (1.15, 2.30, 6.0)}


def _same_rule_reference(quality, gen):
    s = [x for x in quality["sets"] if x["generator"] == gen][0]
    return (np.array([r["J_scipy_same_rule"] * r["sum_T_scipy_same_rule"] ** 7 for r in s["paths"]]),
            np.mean([r["evaluations_same_rule"] for r in s["paths"]]))


@pytest.mark.parametrize("gen", ["box", "walk"])
def test_own_lbfgs_against_scipy_under_the_same_stopping_rule(quality, gen):
    batch, _, _ = _reference(quality, gen)
    sr, scipy_evals = _same_rule_reference(quality, gen)
    own, evals = np.zeros(batch.n_paths), []
    for p in range(batch.n_paths):
        wp, m, v = batch.path(p)
        _, t, ne, _ = po.optimize_times(4, m, v, po.estimate_times(wp, batch.limits[p]), po.default_nlopt())
        own[p] = _normalised(po.cost_and_gradient(4, m, v, t)[0], t)
        evals.append(ne)
    med, p90, ev = SAME_RULE[gen]
    r = own / sr
    assert np.median(r) <= med and np.percentile(r, 90) <= p90, (np.median(r), np.percentile(r, 90))
    assert np.mean(evals) <= ev and scipy_evals <= ev       # both stop early: the rule, not the search


@pytest.mark.gpu
@pytest.mark.parametrize("gen", ["box", "walk"])
def test_gpu_outer_loop_against_scipy_lbfgsb(gpu_ctx, quality, gen):
    """The same check through the C ABI (shipping tolerances): limits far away, so that the feasibility scaling that follows
    the outer loop leaves the optimiser's segment times alone."""
    from mrs_uav_trajectory_generation_amd import api
    batch, sc, st = _reference(quality, gen)
    far = pr.random_batch(quality["n_paths"], quality["n_segments"], seed0=quality["seed0"], generator=gen,
                          limits=np.full(9, 1e9))
    t0 = np.concatenate([po.estimate_times(batch.path(p)[0], batch.limits[p]) for p in range(batch.n_paths)])
    out = gpu_ctx.solve_batch(far, t0, time_alloc_method=api.TIME_ALLOC_MELLINGER)
    own = np.array([_normalised(out["cost"][p], out["times"][batch.seg_offsets[p]:batch.seg_offsets[p + 1]])
                    for p in range(batch.n_paths)])
    med, p90, start, helped = BOUNDS[(gen, "reference")]
    r = own / sc
    assert np.median(r) <= med and np.percentile(r, 90) <= p90, (np.median(r), np.percentile(r, 90))
    assert np.median(own / st) <= start and np.mean(own < st) >= helped
    sr, _ = _same_rule_reference(quality, gen)
    med, p90, _ = SAME_RULE[gen]
    assert np.median(own / sr) <= med and np.percentile(own / sr, 90) <= p90, (np.median(own / sr), np.percentile(own / sr, 90))
