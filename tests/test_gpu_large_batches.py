"""Kernel variants that only engage on larger batches (the library picks the lane mapping from the batch
size, or that are forced): the lane solve kernels -- one lane per (path, dimension) up to 32768 paths, one lane per path
beyond -- run when a path is too long for an LDS tile or when MRS_TG_TILE_MAX_PATHS forces them; compact outer-loop
mapping (P > 3072).  Parity is checked against the oracle on a strided subset (the oracle needs
~30 us per linear path and ~2 ms per nonlinear path), and on every path through size-independent
properties: continuity, constraints, linearity, agreement between the materialised-block and the fused
pipelines, and agreement with the same paths solved in a small batch (different kernels, same answer)."""
import numpy as np
import pytest

from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po
from tests import util

pytestmark = pytest.mark.gpu


def _subset_vs_oracle(batch, out, idx, tol):
    sub = batch.select(idx)
    t = np.concatenate([out["times"][batch.seg_offsets[p]:batch.seg_offsets[p + 1]] for p in idx])
    ref = util.oracle_linear(sub, t)
    got = np.concatenate([out["coeffs"][batch.seg_offsets[p]:batch.seg_offsets[p + 1]] for p in idx])
    assert util.coeff_error(got, ref["coeffs"], sub.seg_offsets) < tol


@pytest.mark.parametrize("n_paths,tile_max", [(2304, None), (6400, None), (6400, 0), (33024, None), (33024, 0)])
def test_linear_large_batches(gpu_ctx, monkeypatch, n_paths, tile_max):
    if tile_max is not None:
        monkeypatch.setenv("MRS_TG_TILE_MAX_PATHS", str(tile_max))  # read at every call: both pipelines on the lane kernels
        monkeypatch.setenv("MRS_TG_ROWS_KERNEL", "0")                # (and not on the rows kernel, the default fused solve)
    batch = pr.random_batch(n_paths, 10, seed0=7000)
    out = gpu_ctx.solve_batch(batch, None, flags=api.FLAG_MATERIALIZED_BLOCKS)
    fused = gpu_ctx.solve_batch(batch, out["times"])
    assert np.all(out["status"] == 1) and np.all(fused["status"] == 1)
    # The two pipelines round the block entries differently (1 ulp); the solution moves by cond(R_pp) * eps,
    # and among tens of thousands of random paths a few have a 0.05 s segment next to 10 s ones (cond ~1e9).
    err = np.array([util.coeff_error(out["coeffs"][a:b], fused["coeffs"][a:b])
                    for a, b in zip(batch.seg_offsets[:-1], batch.seg_offsets[1:])])
    assert np.percentile(err, 99) < 1e-10 and err.max() < 1e-5, (np.percentile(err, 99), err.max())
    assert np.percentile(np.abs(out["cost"] - fused["cost"]) / np.abs(fused["cost"]), 99) < 1e-10
    idx = list(range(0, n_paths, n_paths // 97))
    _subset_vs_oracle(batch, out, idx, 1e-7)
    # the DEFAULT solve's result as well (from 6144 paths on that is solve_quad_kernel, the kernel of bench.py's grouped
    # dispatches): held to the oracle itself, not only to the materialised-block result above
    _subset_vs_oracle(batch, fused, idx, 1e-7)
    # the same paths in a small batch go through the tile kernel: same answer
    monkeypatch.delenv("MRS_TG_TILE_MAX_PATHS", raising=False)
    small = batch.select(idx)
    ts = np.concatenate([out["times"][batch.seg_offsets[p]:batch.seg_offsets[p + 1]] for p in idx])
    sout = gpu_ctx.solve_batch(small, ts)
    got = np.concatenate([out["coeffs"][batch.seg_offsets[p]:batch.seg_offsets[p + 1]] for p in idx])
    assert util.coeff_error(got, sout["coeffs"], small.seg_offsets) < 1e-9
    chk = batch.select(range(0, n_paths, n_paths // 31))
    tc = np.concatenate([out["times"][batch.seg_offsets[p]:batch.seg_offsets[p + 1]] for p in range(0, n_paths, n_paths // 31)])
    cc = np.concatenate([out["coeffs"][batch.seg_offsets[p]:batch.seg_offsets[p + 1]] for p in range(0, n_paths, n_paths // 31)])
    assert util.continuity_defect(chk, cc, tc) < 1e-9 and util.constraint_defect(chk, cc, tc) < 1e-9


def test_linear_ragged_large_batch(gpu_ctx):
    batch = pr.random_batch(7000, "ragged", seed0=8000)     # BASELINE config 5 shape, per-lane kernels
    out = gpu_ctx.solve_batch(batch, None)
    assert np.all(out["status"] == 1)
    _subset_vs_oracle(batch, out, list(range(0, 7000, 61)), 1e-7)
    small = pr.random_batch(4500, "ragged", seed0=8000)     # same shape through the tile kernel
    outs = gpu_ctx.solve_batch(small, None)
    assert np.all(outs["status"] == 1)
    _subset_vs_oracle(small, outs, list(range(0, 4500, 41)), 1e-7)


def test_long_paths_one_per_tile(gpu_ctx):
    batch = pr.random_batch(6, 120, seed0=8100)             # 120 segments: one path per LDS tile
    out = gpu_ctx.solve_batch(batch, None)
    assert np.all(out["status"] == 1)
    _subset_vs_oracle(batch, out, list(range(6)), 1e-6)
    nl = gpu_ctx.solve_batch(batch.select([0, 1]), None, time_alloc_method=api.TIME_ALLOC_MELLINGER)
    assert np.all(np.isin(nl["status"], (1, 3, 4, 5))) and np.all(np.isfinite(nl["coeffs"]))


@pytest.mark.parametrize("n_seg", [129, 200, 218, 219, 256])
def test_long_paths_tile_and_lane_kernels_agree(gpu_ctx, monkeypatch, n_seg):
    """129 .. 256 segments (MRS_TG_MAX_SEGMENTS; what the reference's subdivision loop can grow a request to): an LDS tile holds
    a path of up to 218 segments (one path per workgroup); longer ones go to the lane kernels.  Both must give the same
    trajectory, continuous and on its constraints -- and it must be the ORACLE's (round 6: the oracle follows the product to
    256 segments; its QR solve skips the exact zeros outside R_pp's band, bit-identical to the dense loops), in the
    reference's arithmetic and in the 113-bit route."""
    batch = pr.random_batch(3, n_seg, seed0=8300 + n_seg)
    out = gpu_ctx.solve_batch(batch, None, flags=api.FLAG_MATERIALIZED_BLOCKS)
    fused = gpu_ctx.solve_batch(batch, out["times"])
    monkeypatch.setenv("MRS_TG_TILE_MAX_PATHS", "0")
    monkeypatch.setenv("MRS_TG_ROWS_KERNEL", "0")
    lane = gpu_ctx.solve_batch(batch, out["times"], flags=api.FLAG_MATERIALIZED_BLOCKS)
    assert np.all(out["status"] == 1) and np.all(lane["status"] == 1)
    assert util.coeff_error(out["coeffs"], lane["coeffs"], batch.seg_offsets) < 1e-8
    assert util.coeff_error(fused["coeffs"], lane["coeffs"], batch.seg_offsets) < 1e-8
    assert util.continuity_defect(batch, out["coeffs"], out["times"]) < 1e-8
    assert util.constraint_defect(batch, out["coeffs"], out["times"]) < 1e-8
    # the oracle at the same times: the reference-style double route (its own error grows with the path: 1e-6), and the
    # 113-bit route, where what is left is the HIP path's error
    assert np.allclose(out["times"], util.oracle_times(batch), rtol=1e-12, atol=0)
    ref = util.oracle_linear(batch, out["times"])
    po.lib().mto_set_arithmetic(po.QUAD_PRECISION)
    try:
        refq = util.oracle_linear(batch, out["times"])
    finally:
        po.lib().mto_set_arithmetic(po.REFERENCE_ARITHMETIC)
    for name, got in (("blocks", out), ("fused", fused), ("lane", lane)):
        e, eq = util.coeff_error(got["coeffs"], ref["coeffs"], batch.seg_offsets), util.coeff_error(got["coeffs"], refq["coeffs"], batch.seg_offsets)
        print("ERR long linear S=%d %s: vs oracle %.2e, vs 113-bit %.2e" % (n_seg, name, e, eq))
        assert e < 1e-6 and eq < 1e-9, (name, e, eq)
        assert np.max(np.abs(got["cost"] - refq["cost"]) / np.abs(refq["cost"])) < 1e-9


@pytest.mark.parametrize("deriv,n_seg,moving", [(4, 129, False), (2, 200, True), (2, 256, False), (4, 256, True)])
def test_long_paths_mellinger_vs_oracle(gpu_ctx, deriv, n_seg, moving):
    """The whole Mellinger pipeline (outer loop, feasibility scaling, sampling) on paths of 129 .. 256 segments against the
    oracle: status on the reference's rule, sample counts equal, times 1e-5, coefficients 1e-6, samples 1e-4 m.  The two
    looser figures are the length of these trajectories, not the kernels: 1000 .. 2100 s and 5000 .. 10500 samples each, a
    relative difference of 2e-6 in one scaled segment time (measured: 2e-9 .. 2e-6; the scale factors are roots of degree-15
    polynomials found by two different methods) shifts every later sample by microseconds at up to 2 m/s.
    (The oracle needs ~1-3 s per path of this length.)"""
    batch = pr.random_batch(6, n_seg, seed0=8600 + n_seg, derivative_to_optimize=deriv)
    if moving:
        batch = _moving(batch, seed=11)
    cap = 12288
    out = gpu_ctx.solve_batch(batch, None, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=cap)
    ref = po.solve_batch(batch.seg_offsets, batch.waypoints, batch.fixed_mask, batch.fixed_values, batch.limits,
                         np.zeros(batch.n_segments), deriv=deriv, time_alloc_method=2, estimate_times=True, sampling_dt=0.2,
                         sample_capacity=cap, n_threads=6)
    so = batch.seg_offsets
    good = 0
    assert np.all(ref["n_samples"] <= cap) and np.all(ref["n_samples"] > 4000)
    for p in range(batch.n_paths):
        a, b = so[p], so[p + 1]
        ok = util.status_matches(out["status"][p], ref["status"][p])
        ok &= np.max(np.abs(out["times"][a:b] - ref["times"][a:b]) / ref["times"][a:b]) < 1e-5
        ok &= util.coeff_error(out["coeffs"][a:b], ref["coeffs"][a:b]) < 1e-6
        ok &= int(out["n_samples"][p]) == int(ref["n_samples"][p])
        if ok:
            n = min(int(ref["n_samples"][p]), cap)
            ok &= np.max(np.abs(out["samples"][p, :n, :3] - ref["samples"][p, :n, :3])) < 1e-4
        good += bool(ok)
    print("RATE long mellinger d=%d S=%d moving %s: %d / %d" % (deriv, n_seg, moving, good, batch.n_paths))
    assert good >= batch.n_paths - 1, good
    assert util.continuity_defect(batch, out["coeffs"], out["times"]) < 1e-8
    assert util.constraint_defect(batch, out["coeffs"], out["times"]) < 1e-8


def test_longest_accepted_path_and_one_beyond(gpu_ctx):
    batch = pr.random_batch(2, 256, seed0=8200)              # MRS_TG_MAX_SEGMENTS
    out = gpu_ctx.solve_batch(batch, None, time_alloc_method=api.TIME_ALLOC_MELLINGER, max_iterations=3, sampling_dt=0.2,
                              sample_capacity=64)
    assert np.all(np.isin(out["status"], (1, 3, 4, 5))) and np.all(np.isfinite(out["coeffs"]))
    assert util.continuity_defect(batch, out["coeffs"], out["times"]) < 1e-8
    ref = po.solve_batch(batch.seg_offsets, batch.waypoints, batch.fixed_mask, batch.fixed_values, batch.limits,
                         np.zeros(batch.n_segments), deriv=4, time_alloc_method=2, estimate_times=True, max_iterations=3, n_threads=2)
    assert all(util.status_matches(o, r) for o, r in zip(out["status"], ref["status"]))
    assert np.max(np.abs(out["times"] - ref["times"]) / ref["times"]) < 1e-6
    with pytest.raises(api.MrsTgError, match="at most 256"):
        gpu_ctx.solve_batch(pr.random_batch(1, 257, seed0=1), None)


def test_nonlinear_compact_mapping_matches_split_mapping(gpu_ctx):
    # P > 3072 switches the outer loop to one lane per time vector; the same paths in a small batch use four
    batch = pr.random_batch(4200, 10, seed0=9000)
    out = gpu_ctx.solve_batch(batch, None, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=64)
    assert np.all(np.isin(out["status"], (1, 3, 4, 5)))
    idx = list(range(0, 4200, 35))
    small = gpu_ctx.solve_batch(batch.select(idx), None, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2,
                                sample_capacity=64)
    agree = 0
    for k, p in enumerate(idx):
        a, b = batch.seg_offsets[p], batch.seg_offsets[p + 1]
        if out["status"][p] == small["status"][k] and \
                np.max(np.abs(out["times"][a:b] - small["times"][10 * k:10 * k + 10]) / small["times"][10 * k:10 * k + 10]) < 1e-7:
            agree += 1
    print("RATE compact_vs_split: %d / %d" % (agree, len(idx)))
    assert agree >= len(idx) - 1, agree   # measured 120 / 120
    # and against the oracle on a handful
    few = idx[:12]
    ref = po.solve_batch(batch.select(few).seg_offsets, batch.select(few).waypoints, batch.select(few).fixed_mask,
                         batch.select(few).fixed_values, batch.select(few).limits, np.zeros(120), deriv=4,
                         time_alloc_method=2, estimate_times=True, n_threads=4)
    good = sum(1 for k, p in enumerate(few)
               if np.max(np.abs(out["times"][10 * p:10 * p + 10] - ref["times"][10 * k:10 * k + 10]) / ref["times"][10 * k:10 * k + 10]) < 1e-6)
    assert good >= len(few) - 1


def test_separate_sampler_equals_the_sampler_in_the_solve_kernels_tail(gpu_ctx):
    """Above 2048 paths the samples come from sample_kernel (its own launch), below from the tail of solve_rows_kernel: the
    same walk -- the reference's additions in the reference's order -- so sample counts and every sample agree bit for bit, on
    uniform and ragged paths, with a capacity some paths overflow, and with a path that ends exactly on a sample time.
    (Round 3 also built a one-lane-per-path walk for large batches -- 64 paths per wavefront, the pair (time in segment,
    segment) parked in the sample's own output slot, a second kernel evaluating it; bit-identical by this test and three times
    SLOWER at 8192 paths, 147 + 41 us against 52: the lane-private 16-byte stores of the walk cost a memory round trip per
    sample.  Not kept)"""
    for batch, cap in ((pr.random_batch(4608, 10, seed0=40), 192), (pr.random_batch(4100, "ragged", seed0=41), 160)):
        times = gpu_ctx.solve_batch(batch, None)["times"]
        times[:10] = 1.0    # 10 s at dt 0.2: the last sample falls on the end time (the carry test decides)
        big = gpu_ctx.solve_batch(batch, times, sampling_dt=0.2, sample_capacity=cap)
        assert (big["n_samples"] == cap + 1).any() and (big["n_samples"] <= cap).any()
        so = batch.seg_offsets
        for a in range(0, batch.n_paths, 1024):
            idx = list(range(a, min(a + 1024, batch.n_paths)))
            sub = batch.select(idx)
            small = gpu_ctx.solve_batch(sub, times[so[idx[0]]:so[idx[-1] + 1]], sampling_dt=0.2, sample_capacity=cap)
            assert np.array_equal(small["n_samples"], big["n_samples"][idx])
            for k, p in enumerate(idx):
                n = min(int(small["n_samples"][k]), cap)
                assert np.array_equal(small["samples"][k, :n], big["samples"][p, :n]), p


@pytest.mark.parametrize("deriv,ragged", [(2, False), (3, False), (2, True), (2, 2), (3, 3)])
def test_objective_orders_below_snap_on_the_saturated_device_kernels(gpu_ctx, deriv, ragged):
    """Rest-to-rest paths under min-acceleration / min-jerk leave jerk and / or snap FREE at their end vertices
    (makeStartOrEnd(., derivative_to_optimize): the nodelet's default config is min-acceleration).  From 6144 paths per launch
    the fixed-times solve runs solve_quad_kernel<., true>, which eliminates such an end vertex like an interior one (round 5;
    the general step before), and the Mellinger outer loop of large batches runs the shared half sweeps with free ends
    (optimize_lean_shared_ends_kernel).  Both against the oracle on a strided subset, every path through the invariants, and
    the same paths in a small batch (other kernels) to 1e-9."""
    n = 6400
    batch = pr.random_batch(n, "ragged" if ragged is True else (ragged or 10), seed0=9100, derivative_to_optimize=deriv)   # (2, 3: that many segments)
    api.kernel_trace_reset()
    lin = gpu_ctx.solve_batch(batch, None)
    trace = api.kernel_trace()
    if ragged is not True:   # (a ragged 3..30 batch does not fit the quad kernel's LDS records: rows kernel)
        assert "solve_quad_kernel<false, true>" in trace, trace
    assert np.all(lin["status"] == 1)
    so = batch.seg_offsets
    idx = list(range(0, n, n // 127))
    _subset_vs_oracle(batch, lin, idx, 1e-7)
    small = batch.select(idx)
    ts = np.concatenate([lin["times"][so[p]:so[p + 1]] for p in idx])
    sout = gpu_ctx.solve_batch(small, ts)
    got = np.concatenate([lin["coeffs"][so[p]:so[p + 1]] for p in idx])
    assert util.coeff_error(got, sout["coeffs"], small.seg_offsets) < 1e-9
    assert np.max(np.abs(lin["cost"][idx] - sout["cost"]) / np.abs(sout["cost"])) < 1e-9
    chk = list(range(0, n, 13))
    sub = batch.select(chk)
    tc = np.concatenate([lin["times"][so[p]:so[p + 1]] for p in chk])
    cc = np.concatenate([lin["coeffs"][so[p]:so[p + 1]] for p in chk])
    assert util.continuity_defect(sub, cc, tc) < 1e-9 and util.constraint_defect(sub, cc, tc) < 1e-9
    # ---- the Mellinger pipeline of the same batch
    api.kernel_trace_reset()
    out = gpu_ctx.solve_batch(batch, None, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=192)
    trace = api.kernel_trace()
    assert "optimize_lean_shared_ends_kernel" in trace, trace
    oidx = list(range(0, n, n // 255))
    osub = batch.select(oidx)
    ref = po.solve_batch(osub.seg_offsets, osub.waypoints, osub.fixed_mask, osub.fixed_values, osub.limits,
                         np.zeros(osub.n_segments), deriv=deriv, time_alloc_method=2, estimate_times=True, sampling_dt=0.2,
                         sample_capacity=192, n_threads=8)
    good = 0
    for k, p in enumerate(oidx):
        a, b = osub.seg_offsets[k], osub.seg_offsets[k + 1]
        t = out["times"][so[p]:so[p + 1]]
        if util.status_matches(out["status"][p], ref["status"][k]) and np.max(np.abs(t - ref["times"][a:b]) / ref["times"][a:b]) < 1e-6 \
                and util.coeff_error(out["coeffs"][so[p]:so[p + 1]], ref["coeffs"][a:b]) < 1e-6:
            good += 1
    print("RATE below-snap d=%d %s: %d / %d" % (deriv, {True: "ragged", False: "10 segments"}.get(ragged, "%s segments" % ragged), good, len(oidx)))
    assert good >= len(oidx) - 3, (good, len(oidx))


def _moving(batch, seed=5):
    """every path starts from a moving state: velocity / acceleration / jerk of its first vertex constrained to non-zero values
    (a replanning request of a vehicle in flight, /root/reference/src/mrs_trajectory_generation.cpp:946-957)"""
    rng = np.random.default_rng(seed)
    parts = []
    for p in range(batch.n_paths):
        wp, _, _ = batch.path(p)
        init = dict(heading=wp[0, 3], velocity=np.append(rng.uniform(-1, 1, 3), 0.1),
                    acceleration=np.append(rng.uniform(-0.5, 0.5, 3), 0.0), jerk=np.append(rng.uniform(-0.2, 0.2, 3), 0.0))
        parts.append(pr.build_vertices(wp, batch.derivative_to_optimize, initial_state=init))
    return pr.assemble_batch(parts, batch.limits, batch.derivative_to_optimize)


@pytest.mark.parametrize("deriv,n_seg", [(4, 10), (2, 10), (3, 10), (2, "ragged"), (4, 20)])
def test_moving_starts_on_the_saturated_device_kernels(gpu_ctx, deriv, n_seg):
    """Paths that start from a moving state -- what a nodelet sends for every replanning request in flight -- in the large
    batches' kernels (round 5: the shared half sweeps take the start vertex's constrained derivative values as right-hand-side
    terms of their first step, with or without free slots beside them; the sweeping kernel behind the launch took such paths
    before).  Mellinger pipeline against the oracle on a strided subset, the kernel trace says which outer loop ran."""
    n = 6400 if n_seg != 20 else 3000
    batch = _moving(pr.random_batch(n, n_seg, seed0=9300, derivative_to_optimize=deriv))
    so = batch.seg_offsets
    if n_seg == 10:
        # the fixed-times solve: solve_quad_kernel keeps such paths on its fast road too (the start vertex's values are
        # right-hand-side terms of its first step and the first segment's boundary values; the general step before)
        api.kernel_trace_reset()
        lin = gpu_ctx.solve_batch(batch, None)
        # (min-snap: the two-sided kernel, whose side 0 takes the moving start; below snap the quad kernel with free end slots)
        assert ("solve_quad_kernel<false, true>" if deriv < 4 else "solve_duo_kernel<false>") in api.kernel_trace(), api.kernel_trace()
        assert np.all(lin["status"] == 1)
        idx = list(range(0, n, n // 127))
        _subset_vs_oracle(batch, lin, idx, 1e-7)
        small = batch.select(idx)
        ts = np.concatenate([lin["times"][so[p]:so[p + 1]] for p in idx])
        sout = gpu_ctx.solve_batch(small, ts)          # (127 paths: the rows kernel)
        got = np.concatenate([lin["coeffs"][so[p]:so[p + 1]] for p in idx])
        assert util.coeff_error(got, sout["coeffs"], small.seg_offsets) < 1e-9
        assert np.max(np.abs(lin["cost"][idx] - sout["cost"]) / np.abs(sout["cost"])) < 1e-9
        c13 = list(range(0, n, 13))
        s13 = batch.select(c13)
        assert util.continuity_defect(s13, np.concatenate([lin["coeffs"][so[p]:so[p + 1]] for p in c13]),
                                      np.concatenate([lin["times"][so[p]:so[p + 1]] for p in c13])) < 1e-9
        assert util.constraint_defect(s13, np.concatenate([lin["coeffs"][so[p]:so[p + 1]] for p in c13]),
                                      np.concatenate([lin["times"][so[p]:so[p + 1]] for p in c13])) < 1e-9
    api.kernel_trace_reset()
    out = gpu_ctx.solve_batch(batch, None, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=192)
    trace = api.kernel_trace()
    if n_seg != "ragged" or deriv < 4:   # (a ragged min-snap batch runs the mixed kernel, whose one-sided sweeps do not take them)
        assert ("optimize_lean_shared_ends_kernel" if deriv < 4 else "optimize_lean_shared_kernel") in trace, trace
    oidx = list(range(0, n, n // 255))
    osub = batch.select(oidx)
    ref = po.solve_batch(osub.seg_offsets, osub.waypoints, osub.fixed_mask, osub.fixed_values, osub.limits,
                         np.zeros(osub.n_segments), deriv=deriv, time_alloc_method=2, estimate_times=True, sampling_dt=0.2,
                         sample_capacity=192, n_threads=8)
    good = 0
    for k, p in enumerate(oidx):
        a, b = osub.seg_offsets[k], osub.seg_offsets[k + 1]
        t = out["times"][so[p]:so[p + 1]]
        if util.status_matches(out["status"][p], ref["status"][k]) and np.max(np.abs(t - ref["times"][a:b]) / ref["times"][a:b]) < 1e-6 \
                and util.coeff_error(out["coeffs"][so[p]:so[p + 1]], ref["coeffs"][a:b]) < 1e-6:
            good += 1
    print("RATE moving starts d=%d %s: %d / %d" % (deriv, n_seg, good, len(oidx)))
    assert good >= len(oidx) - 3, (good, len(oidx))
    chk = list(range(0, n, 17))
    sub = batch.select(chk)
    tc = np.concatenate([out["times"][so[p]:so[p + 1]] for p in chk])
    cc = np.concatenate([out["coeffs"][so[p]:so[p + 1]] for p in chk])
    assert util.continuity_defect(sub, cc, tc) < 1e-9 and util.constraint_defect(sub, cc, tc) < 1e-9


@pytest.mark.parametrize("deriv,n_seg,every,moving", [(2, 10, 1, False), (3, 10, 2, False), (2, "ragged", 3, False), (2, 10, 2, True),
                                                      (4, 10, 2, False), (4, "ragged", 1, True)])
def test_stop_at_vertices_below_snap_on_the_saturated_device_kernels(gpu_ctx, deriv, n_seg, every, moving):
    """stop_at waypoints (velocity = acceleration = jerk = 0 at an interior vertex, snap free;
    /root/reference/src/mrs_trajectory_generation.cpp:969-973) under the objective orders below snap: the free-end
    instantiations of the large batches' kernels eliminate such a vertex with the identity in its constrained slots (round 5;
    the general steps before).  Every `every`-th interior vertex is one; optionally from a moving state.  Fixed-times solve
    and Mellinger pipeline against the oracle on a strided subset, invariants on more paths.  Under min-snap the same
    instantiations run when the caller says so (MRS_TG_FLAG_CONSTRAINED_SLOTS) -- mrs_tg_solve_batch, which holds the masks in
    host memory, says it by itself: the d = 4 cases."""
    n = 6400
    base = pr.random_batch(n, n_seg, seed0=9700, derivative_to_optimize=deriv)
    rng = np.random.default_rng(3)
    parts = []
    for p in range(n):
        wp, _, _ = base.path(p)
        V = wp.shape[0]
        stop = [(0 < i < V - 1) and (i % every == 0) for i in range(V)]
        init = None
        if moving:
            init = dict(heading=wp[0, 3], velocity=np.append(rng.uniform(-1, 1, 3), 0.1), acceleration=np.append(rng.uniform(-0.5, 0.5, 3), 0.0),
                        jerk=np.append(rng.uniform(-0.2, 0.2, 3), 0.0))
        parts.append(pr.build_vertices(wp, deriv, stop_at=stop, initial_state=init))
    batch = pr.assemble_batch(parts, base.limits, deriv)
    so = batch.seg_offsets
    if n_seg == 10:
        api.kernel_trace_reset()
        lin = gpu_ctx.solve_batch(batch, None)
        assert "solve_quad_kernel<false, true>" in api.kernel_trace(), api.kernel_trace()
        assert np.all(lin["status"] == 1)
        idx = list(range(0, n, n // 127))
        _subset_vs_oracle(batch, lin, idx, 1e-7)
        small = batch.select(idx)
        ts = np.concatenate([lin["times"][so[p]:so[p + 1]] for p in idx])
        sout = gpu_ctx.solve_batch(small, ts)          # (127 paths: the rows kernel)
        got = np.concatenate([lin["coeffs"][so[p]:so[p + 1]] for p in idx])
        assert util.coeff_error(got, sout["coeffs"], small.seg_offsets) < 1e-9
        assert np.max(np.abs(lin["cost"][idx] - sout["cost"]) / np.abs(sout["cost"])) < 1e-9
    api.kernel_trace_reset()
    out = gpu_ctx.solve_batch(batch, None, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=192)
    assert "optimize_lean_shared_ends_kernel" in api.kernel_trace(), api.kernel_trace()
    oidx = list(range(0, n, n // 255))
    osub = batch.select(oidx)
    ref = po.solve_batch(osub.seg_offsets, osub.waypoints, osub.fixed_mask, osub.fixed_values, osub.limits,
                         np.zeros(osub.n_segments), deriv=deriv, time_alloc_method=2, estimate_times=True, sampling_dt=0.2,
                         sample_capacity=192, n_threads=8)
    good = 0
    for k, p in enumerate(oidx):
        a, b = osub.seg_offsets[k], osub.seg_offsets[k + 1]
        t = out["times"][so[p]:so[p + 1]]
        if util.status_matches(out["status"][p], ref["status"][k]) and np.max(np.abs(t - ref["times"][a:b]) / ref["times"][a:b]) < 1e-6 \
                and util.coeff_error(out["coeffs"][so[p]:so[p + 1]], ref["coeffs"][a:b]) < 1e-6:
            good += 1
    print("RATE stop_at d=%d %s every %d moving %s: %d / %d" % (deriv, n_seg, every, moving, good, len(oidx)))
    assert good >= len(oidx) - 3, (good, len(oidx))
    chk = list(range(0, n, 17))
    sub = batch.select(chk)
    tc = np.concatenate([out["times"][so[p]:so[p + 1]] for p in chk])
    cc = np.concatenate([out["coeffs"][so[p]:so[p + 1]] for p in chk])
    assert util.continuity_defect(sub, cc, tc) < 1e-9 and util.constraint_defect(sub, cc, tc) < 1e-9


@pytest.mark.parametrize("deriv,n_seg,n_paths,moving", [(4, 80, 300, False), (2, 100, 200, True), (2, 61, 64, False), (3, 121, 8, False),
                                                        (4, 64, 100, True), (2, 126, 4, False)])   # (126: beyond the two passes, within the oracle's 128)
def test_paths_of_61_to_121_segments_run_the_shared_half_sweeps_in_two_passes(gpu_ctx, deriv, n_seg, n_paths, moving):
    """A path of more than 60 segments has more half sweeps (S + 4) than a wavefront has lanes: two passes of the shared
    evaluation (each with the three published half sweeps and half of the others) instead of the one-sided sweeps -- what the
    policy layer's subdivision rounds produce all the time (a third of its requests end above 60 waypoints).  A strided subset
    against the oracle (measured on every path of larger batches: 50 / 50, 50 / 50, 64 / 64, 40 / 40, 50 / 50), the invariants on all."""
    batch = pr.random_batch(n_paths, n_seg, seed0=9900, derivative_to_optimize=deriv)
    if moving:
        batch = _moving(batch, seed=9)
    api.kernel_trace_reset()
    out = gpu_ctx.solve_batch(batch, None, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=2048)
    trace = api.kernel_trace()
    # (every objective order: its table is the run-time order's; beyond 121 segments the one-sided masked sweeps remain)
    assert ("optimize_lean_shared_ends_long_kernel" if n_seg <= 121 else "optimize_lean_masked_kernel") in trace, trace
    idx = list(range(0, n_paths, max(1, n_paths // 16)))      # (the oracle needs ~1 s per path of this length)
    sub = batch.select(idx)
    ref = po.solve_batch(sub.seg_offsets, sub.waypoints, sub.fixed_mask, sub.fixed_values, sub.limits, np.zeros(sub.n_segments),
                         deriv=deriv, time_alloc_method=2, estimate_times=True, sampling_dt=0.2, sample_capacity=2048, n_threads=8)
    so = batch.seg_offsets
    good = 0
    for k, p in enumerate(idx):
        a, b = sub.seg_offsets[k], sub.seg_offsets[k + 1]
        t = out["times"][so[p]:so[p + 1]]
        if util.status_matches(out["status"][p], ref["status"][k]) and np.max(np.abs(t - ref["times"][a:b]) / ref["times"][a:b]) < 1e-6 \
                and util.coeff_error(out["coeffs"][so[p]:so[p + 1]], ref["coeffs"][a:b]) < 1e-6:
            good += 1
    print("RATE two passes d=%d S=%d moving %s: %d / %d" % (deriv, n_seg, moving, good, len(idx)))
    assert good >= len(idx) - 2, (good, len(idx))
    assert util.continuity_defect(batch, out["coeffs"], out["times"]) < 1e-9
    assert util.constraint_defect(batch, out["coeffs"], out["times"]) < 1e-9
