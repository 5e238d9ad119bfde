"""BASELINE.json's configurations at their FULL sizes on one MI355X, checked through size-independent properties
(the oracle needs ~2 ms per nonlinear path, so it only sees a strided subset):

  * configs[3] whole: 65536 random 10-segment paths, Mellinger outer loop + feasibility scaling + sampling;
  * configs[4]: 8192 paths with 3..30 segments (ragged), same pipeline;
  * configs[1] at 65536 paths (linear).

Properties: every path ends with an accepted nlopt code and finite output; paths are independent, so solving the
batch in reversed order gives bit-identical per-path results (every path lands on another lane / workgroup / bin
position); sample counts follow from the segment times (count = number of multiples of dt below the total time,
+-1 for the accumulated rounding); the first sample is the first waypoint; C0..C4 continuity and the vertex
constraints hold on a strided subset; that subset agrees with the oracle as in the small-batch tests."""
import numpy as np
import pytest

from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po
from tests import util

pytestmark = pytest.mark.gpu
CAP = 192


def _reverse(batch):
    return batch.select(list(range(batch.n_paths - 1, -1, -1)))


def _per_path(arr, offsets, p):
    return arr[offsets[p]:offsets[p + 1]]


def _check_nonlinear(gpu_ctx, batch, n_oracle, max_bad=1):
    out = gpu_ctx.solve_batch(batch, None, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=CAP)
    P, so = batch.n_paths, batch.seg_offsets
    assert np.all(np.isin(out["status"], (1, 3, 4, 5, api.STATUS_ROUNDOFF_LIMITED)))
    assert np.all(np.isfinite(out["coeffs"])) and np.all(np.isfinite(out["times"])) and np.all(out["times"] >= 0.01)
    # no runaway leaves the pipeline as a success: a path whose feasibility scaling multiplied its total time by more than
    # MRS_TG_RUNAWAY_TIME_FACTOR carries ROUNDOFF_LIMITED (which the nodelet's gate rejects), and only such paths do
    total = np.add.reduceat(out["times"], so[:-1])
    total0 = np.add.reduceat(gpu_ctx.solve_batch(batch, None)["times"], so[:-1])      # the Euclidean start
    runaway = total > api.RUNAWAY_TIME_FACTOR * total0
    assert np.array_equal(runaway, out["status"] == api.STATUS_ROUNDOFF_LIMITED), (runaway.sum(), (out["status"] == -4).sum())
    assert runaway.mean() < 2e-3, runaway.sum()
    print("runaway paths flagged: %d of %d (largest healthy ratio %.2f)" % (runaway.sum(), P, (total / total0)[~runaway].max()))
    # sample counts from the times
    expect = np.minimum(np.ceil(total / 0.2 - 1e-9), CAP + 1)
    assert np.all(np.abs(out["n_samples"] - expect) <= 1), np.max(np.abs(out["n_samples"] - expect))
    # first sample = first waypoint (positions exactly; heading wrapped)
    first_wp = batch.waypoints[so[:-1] + np.arange(P)]
    assert np.max(np.abs(out["samples"][:, 0, :3] - first_wp[:, :3])) < 1e-9
    # independence of the paths: the reversed batch gives the same answers bit for bit
    rev = gpu_ctx.solve_batch(_reverse(batch), None, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2,
                              sample_capacity=CAP)
    assert np.array_equal(rev["status"][::-1], out["status"])
    assert np.array_equal(rev["n_samples"][::-1], out["n_samples"])
    rso = np.concatenate([[0], np.cumsum(np.diff(so)[::-1])])
    for p in range(0, P, max(1, P // 512)):
        q = P - 1 - p
        assert np.array_equal(_per_path(out["times"], so, p), _per_path(rev["times"], rso, q)), p
        assert np.array_equal(_per_path(out["coeffs"], so, p), _per_path(rev["coeffs"], rso, q)), p
        n = min(int(out["n_samples"][p]), CAP)
        assert np.array_equal(out["samples"][p, :n], rev["samples"][q, :n]), p
    # structural invariants and the oracle on a strided subset
    idx = list(range(0, P, max(1, P // n_oracle)))[:n_oracle]
    sub = batch.select(idx)
    st = np.concatenate([_per_path(out["times"], so, p) for p in idx])
    sc = np.concatenate([_per_path(out["coeffs"], so, p) for p in idx])
    assert util.continuity_defect(sub, sc, st) < 1e-9 and util.constraint_defect(sub, sc, st) < 1e-9
    ref = po.solve_batch(sub.seg_offsets, sub.waypoints, sub.fixed_mask, sub.fixed_values, sub.limits,
                         np.zeros(sub.n_segments), deriv=4, time_alloc_method=2, estimate_times=True, sampling_dt=0.2,
                         sample_capacity=CAP, n_threads=8)
    good = 0
    for k, p in enumerate(idx):
        a, b = sub.seg_offsets[k], sub.seg_offsets[k + 1]
        if util.status_matches(out["status"][p], ref["status"][k]) and np.max(np.abs(st[a:b] - ref["times"][a:b]) / ref["times"][a:b]) < 1e-6 \
                and util.coeff_error(sc[a:b], ref["coeffs"][a:b]) < 1e-6:
            good += 1
    print("RATE baseline P=%d: %d / %d" % (P, good, len(idx)))
    # the gate is what is measured (every path of the subset on all three batches), less ONE path: a regression from 100 % to
    # 99 % fails (round 4's gate was 0.99 of the subset)
    assert good >= len(idx) - max_bad, (good, len(idx))
    # statuses on the REFERENCE's rule (pyoracle's default): the product's -4 paths are the oracle's own runaways, which the
    # reference hands back with a success code
    sub_out = dict(status=out["status"][idx], times=st)
    flagged = util.runaway_sets_agree(sub, sub_out, ref)
    print("subset paths flagged ROUNDOFF_LIMITED where the reference's rule reports a success code: %d" % flagged)
    return good / len(idx)


def test_config2_1024_paths_nonlinear(gpu_ctx):
    """BASELINE configs[2] at its own size: every one of the 1024 paths through the invariants, a strided 256 of them
    against the oracle (measured 256 / 256)."""
    rate = _check_nonlinear(gpu_ctx, pr.random_batch(1024, 10, seed0=0), 256)
    print("configs[2] agreement with the oracle on the strided subset: %.4f" % rate)


def test_config3_whole_65536_paths_nonlinear(gpu_ctx):
    batch = pr.random_batch(65536, 10, seed0=0)
    _check_nonlinear(gpu_ctx, batch, 512, max_bad=2)   # measured 511 / 512 (round 5; the gate of round 4 was 507)
    # path 8615 (a 2.7 s segment next to one scaled to 9e11 s) and its siblings, which round 2 returned as successes
    one = batch.select([8615])
    out = gpu_ctx.solve_batch(one, None, time_alloc_method=api.TIME_ALLOC_MELLINGER)
    assert out["status"][0] == api.STATUS_ROUNDOFF_LIMITED and out["times"].sum() > 1e6
    # the deviation itself, on the GPU: the REFERENCE's rule hands the same runaway back with the outer loop's code (a
    # "success" its nodelet then discards by the length check), the product's rule names it
    ref = po.solve_batch(one.seg_offsets, one.waypoints, one.fixed_mask, one.fixed_values, one.limits, np.zeros(10), deriv=4,
                         time_alloc_method=2, estimate_times=True)
    assert ref["status"][0] >= 1 and ref["times"].sum() > 1e6
    assert util.status_matches(out["status"], ref["status"]).all() and out["status"][0] != ref["status"][0]
    assert util.runaway_sets_agree(one, out, ref) == 1


def test_config4_8192_ragged_paths_nonlinear(gpu_ctx):
    batch = pr.random_batch(8192, "ragged", seed0=0)
    counts = np.diff(batch.seg_offsets)
    assert counts.min() == 3 and counts.max() == 30
    _check_nonlinear(gpu_ctx, batch, 512)   # measured 512 / 512 (round 2 ran 96 paths against a gate of 0.95)


def test_config1_at_65536_paths_linear_is_linear_in_the_waypoints(gpu_ctx):
    """fixed times: the coefficients are a linear function of the constrained values; c(a X + b Y) = a c(X) + b c(Y)
    for every one of the 65536 paths (a checksum of the whole batch without any reference solve)"""
    X = pr.random_batch(65536, 10, seed0=0)
    Y = pr.random_batch(65536, 10, seed0=70000)
    times = gpu_ctx.solve_batch(X, None)["times"]
    a, b = 0.75, -1.5
    Z = pr.random_batch(65536, 10, seed0=0)
    Z.fixed_values[:] = a * X.fixed_values + b * Y.fixed_values
    cx = gpu_ctx.solve_batch(X, times)["coeffs"]
    cy = gpu_ctx.solve_batch(Y, times)["coeffs"]
    cz = gpu_ctx.solve_batch(Z, times)["coeffs"]
    lin = a * cx + b * cy
    scale = np.maximum(np.abs(cx).reshape(65536, -1).max(axis=1), np.abs(cy).reshape(65536, -1).max(axis=1))
    err = np.abs(cz - lin).reshape(65536, -1).max(axis=1) / scale
    assert np.percentile(err, 99.9) < 1e-9 and err.max() < 1e-6, (np.percentile(err, 99.9), err.max())
