"""Boundary features added for ABI 2, through the C ABI on the GPU: several devices, the nlopt-style time budget
(MAXTIME_REACHED), the cached plan of the one-call interface, per-dispatch kernel timing, and the policy layer's handling
of one unsolvable request next to healthy ones."""
import ctypes as C
import time

import numpy as np
import pytest
import torch

from mrs_uav_trajectory_generation_amd import api, problem as pr
from tests import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_seg,n_paths,n_dev", [(10, 37, 2), ("ragged", 41, 3), (6, 5, 4)])
def test_multi_device_solve_equals_single_device(gpu_ctx, n_seg, n_paths, n_dev):
    """mrs_tg_multi_solve_batch (here: several contexts on the one GPU of the box, one host thread each) returns, path
    for path, exactly what mrs_tg_solve_batch returns: same kernels per path, so bit-identical."""
    batch = pr.random_batch(n_paths, n_seg, seed0=4200)
    multi = api.MultiContext([0] * n_dev)
    assert multi.n_devices == n_dev
    shard = multi.shard(batch.seg_offsets)
    counts = np.diff(batch.seg_offsets)
    assert set(shard.tolist()) == set(range(n_dev))
    if n_seg == "ragged":  # balanced on the segment count: no shard carries more than the lightest one + the longest path
        load = np.array([counts[shard == r].sum() for r in range(n_dev)])
        assert load.max() - load.min() <= counts.max()
    else:  # contiguous ranges whose sizes differ by at most one
        assert np.all(np.diff(shard) >= 0)
        sizes = np.bincount(shard, minlength=n_dev)
        assert sizes.max() - sizes.min() <= 1
    for kwargs in (dict(), dict(time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=512)):
        one = gpu_ctx.solve_batch(batch, None, **kwargs)
        many = multi.solve_batch(batch, None, **kwargs)
        for key in ("times", "coeffs", "status", "cost"):
            assert np.array_equal(one[key], many[key]), key
        if kwargs:
            assert np.array_equal(one["n_samples"], many["n_samples"])
            for p in range(n_paths):
                n = min(one["n_samples"][p], 512)
                assert np.array_equal(one["samples"][p, :n], many["samples"][p, :n])
    multi.close()


def test_multi_device_solve_with_position_free_vertices(gpu_ctx):
    """every shard sees its own masks: the shards that hold a position-free vertex switch the general route on for themselves,
    in the fixed-times mode and in the time-allocation modes; path for path what one device returns"""
    base = pr.random_batch(23, 6, seed0=4300)
    m = base.fixed_mask.copy()
    for p in (2, 11, 12, 20):
        m[base.seg_offsets[p] + p + 1 + p % 5, 0] = 0
    batch = pr.Batch(base.seg_offsets, base.waypoints, m, base.fixed_values, base.limits)
    multi = api.MultiContext([0, 0, 0])
    for kwargs in (dict(), dict(time_alloc_method=api.TIME_ALLOC_MELLINGER), dict(time_alloc_method=api.TIME_ALLOC_RICHTER_TIME)):
        one = gpu_ctx.solve_batch(batch, None, **kwargs)
        many = multi.solve_batch(batch, None, **kwargs)
        assert np.all(one["status"] >= 1)
        for key in ("times", "coeffs", "status", "cost"):
            assert np.array_equal(one[key], many[key]), (key, kwargs)
    multi.close()


def test_multi_device_reports_the_failing_shard(gpu_ctx):
    batch = pr.random_batch(6, 4, seed0=1)
    multi = api.MultiContext([0, 0])
    with pytest.raises(api.MrsTgError, match="device"):
        multi.solve_batch(batch, None, time_alloc_method=9)
    multi.close()


def test_time_budget_stops_the_outer_loop_with_maxtime(gpu_ctx):
    """max_time_s is nlopt's maxtime (src/mrs_trajectory_generation.cpp:899): a search that is still running when it has
    passed stops at its last evaluated point with code 6, which the nodelet rejects (:1138-1149).  A generous budget
    changes nothing."""
    batch = pr.random_batch(64, 10, seed0=910)
    free = gpu_ctx.solve_batch(batch, None, time_alloc_method=api.TIME_ALLOC_MELLINGER)
    roomy = gpu_ctx.solve_batch(batch, None, time_alloc_method=api.TIME_ALLOC_MELLINGER, max_time_s=30.0)
    assert np.array_equal(free["status"], roomy["status"]) and np.array_equal(free["times"], roomy["times"])
    tight = gpu_ctx.solve_batch(batch, None, time_alloc_method=api.TIME_ALLOC_MELLINGER, max_time_s=1e-7)
    assert np.all(tight["status"] == 6), tight["status"]
    assert np.all(np.isfinite(tight["coeffs"])) and np.all(tight["times"] >= 0.01)
    # the trajectory that comes back is still a valid solve of the linear QP at the returned times
    assert util.continuity_defect(batch, tight["coeffs"], tight["times"]) < 1e-7
    for mode in (api.TIME_ALLOC_RICHTER_TIME, api.TIME_ALLOC_SQUARED_TIME_AND_CONSTRAINTS):
        t2 = gpu_ctx.solve_batch(batch, None, time_alloc_method=mode, max_time_s=1e-7)
        assert np.all(t2["status"] == 6), (mode, t2["status"])
    # findTrajectory's gate turns code 6 into "no trajectory"
    wp = pr.random_box_waypoints(6, 3)
    out = gpu_ctx.find_trajectory(wp, max_time_s=1e-7)
    assert out["status"] == 6 and out["n_samples"] == 0


def test_policy_falls_back_to_the_sampler_when_out_of_time(gpu_ctx):
    """optimize() (src/mrs_trajectory_generation.cpp:702-716): a round that starts late runs findTrajectoryFallback -- the
    request still succeeds, with the fallback sampler's trajectory; fallback_sampling = 1 never looks at the clock."""
    paths = [pr.random_walk_waypoints(6, 70 + i) for i in range(4)]
    ok = api.optimize_paths(gpu_ctx, paths, sample_capacity=2048)
    assert ok["success"].sum() >= 3
    fb = api.optimize_paths(gpu_ctx, paths, policy=api.default_policy_options(fallback_sampling=1), sample_capacity=2048)
    assert fb["success"].sum() == 4
    late = api.optimize_paths(gpu_ctx, paths, policy=api.default_policy_options(max_execution_time_s=1e-9), sample_capacity=2048)
    assert late["success"].sum() == 4 and np.array_equal(late["n_samples"], fb["n_samples"])
    for a, b, n in zip(late["samples"], fb["samples"], fb["n_samples"]):
        assert np.array_equal(a[:n], b[:n])
    both = api.optimize_paths(gpu_ctx, paths, policy=api.default_policy_options(fallback_sampling=1, max_execution_time_s=1e-9),
                              sample_capacity=2048)
    assert both["success"].sum() == 4 and np.array_equal(both["n_samples"], fb["n_samples"])


def test_policy_gives_up_when_the_solve_comes_back_late(gpu_ctx):
    """findTrajectory's checks behind the solve (:1085, :1156, :1171): a budget that is still open when the round starts but
    spent when the solve returns ends the request as "failed to find trajectory" (the next attempt then falls back)."""
    paths = [pr.random_walk_waypoints(30, 170 + i) for i in range(32)]
    # overtime() holds from 0.95 * max - 0.01 s on: 0.01474 s leaves the round 4 ms to start in (the preprocessing of 32 short
    # lists takes a fraction of a millisecond), and a round that has to set up and bring back 67 MB of sample buffers
    # (capacity 65536 per path) takes several times that
    late = api.optimize_paths(gpu_ctx, paths, policy=api.default_policy_options(max_execution_time_s=0.01474), sample_capacity=65536)
    assert late["success"].sum() == 0 and late["n_samples"].sum() == 0
    roomy = api.optimize_paths(gpu_ctx, paths, policy=api.default_policy_options(max_execution_time_s=30.0), sample_capacity=65536)
    assert roomy["success"].sum() >= 30


def test_one_oversized_request_does_not_fail_the_batch(gpu_ctx):
    """A request whose waypoint list is longer than the longest path a plan accepts fails on its own
    (the reference treats requests independently); its neighbours are solved."""
    long_path = np.zeros((300, 4))
    long_path[:, 0] = np.arange(300) * 0.7
    long_path[:, 1] = np.sin(np.arange(300) * 0.3)
    long_path[:, 2] = 3.0
    healthy = pr.random_walk_waypoints(6, 11)
    out = api.optimize_paths(gpu_ctx, [healthy, long_path, healthy], sample_capacity=4096)
    assert out["success"].tolist() == [1, 0, 1]
    assert out["n_samples"][1] == 0 and out["n_samples"][0] == out["n_samples"][2] > 0
    assert np.array_equal(out["samples"][0], out["samples"][2])


def test_policy_reports_why_it_refused(gpu_ctx):
    pol = api.default_policy_options(solver=dict(derivative_to_optimize=1))
    with pytest.raises(api.MrsTgError, match="derivative_to_optimize must be 2, 3 or 4"):
        api.optimize_paths(gpu_ctx, [pr.random_walk_waypoints(5, 1)], policy=pol)


def test_one_call_interface_reuses_its_plan(gpu_ctx):
    """Same batch shape again: no analysis, no structure upload, no workspace allocation; another shape in between
    replaces the cached plan; results never depend on which of the two happened."""
    a = pr.random_batch(256, 10, seed0=60)
    b = pr.random_batch(100, 7, seed0=61)
    first = gpu_ctx.solve_batch(a, None)
    t0 = time.perf_counter()
    for _ in range(20):
        again = gpu_ctx.solve_batch(a, None)
    warm = (time.perf_counter() - t0) / 20
    assert np.array_equal(first["coeffs"], again["coeffs"])
    other = gpu_ctx.solve_batch(b, None)
    back = gpu_ctx.solve_batch(a, None)
    assert np.array_equal(first["coeffs"], back["coeffs"]) and np.all(other["status"] == 1)
    assert warm < 5e-3, warm  # a 256-path call is a fraction of a millisecond once the plan exists


def test_per_dispatch_kernel_timing(gpu_ctx):
    """mrs_tg_last_kernel_ms reads the time stamps of the kernel dispatch itself: positive, and far below the
    launch-to-synchronise wall time of the call that made it."""
    batch = pr.random_batch(1024, 10, seed0=0)
    plan = api.Plan(gpu_ctx, batch.seg_offsets)
    db = api.DeviceBatch(batch, "cuda:0", sample_capacity=256)
    est = api.default_options(estimate_times=1)
    plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints,
               limits=db.limits)
    torch.cuda.synchronize()
    gpu_ctx.set_profiling(True)
    try:
        with pytest.raises(api.MrsTgError):
            gpu_ctx.last_kernel_ms(api.KERNEL_NONLINEAR)  # nothing of that family has run under profiling yet
        H = torch.empty(plan.block_doubles, dtype=torch.float64, device="cuda")
        A = torch.empty(plan.block_doubles, dtype=torch.float64, device="cuda")
        for _ in range(40):  # queued back to back: every launch carries its own pair of events
            plan.assemble(4, db.seg_times, H, A)
        hist = gpu_ctx.kernel_ms_history(api.KERNEL_ASSEMBLE)
        # (6 us each on an idle device; one of forty was seen above 0.1 ms on a busy box)
        assert len(hist) == 40 and all(1e-3 < v < 1.0 for v in hist) and float(np.median(hist)) < 0.1, hist
        ms_asm = gpu_ctx.last_kernel_ms(api.KERNEL_ASSEMBLE)
        assert ms_asm == hist[-1]
        plan.solve(api.default_options(), db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost)
        ms_solve = gpu_ctx.last_kernel_ms(api.KERNEL_SOLVE_LINEAR)
        nl = api.default_options(time_alloc_method=api.TIME_ALLOC_MELLINGER)
        plan.solve(nl, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, limits=db.limits)
        ms_outer = gpu_ctx.last_kernel_ms(api.KERNEL_NONLINEAR)
    finally:
        gpu_ctx.set_profiling(False)
    assert 1e-3 < ms_asm < 0.1, ms_asm        # ~5 us for 16 MB
    assert 1e-3 < ms_solve < 0.2, ms_solve    # ~8 us
    assert 5e-3 < ms_outer < 2.0, ms_outer    # ~80 us
    plan.close()


def test_shared_device_hint_changes_the_launch_shape_not_the_result(gpu_ctx):
    """MRS_TG_FLAG_SHARED_DEVICE packs two paths into a wavefront for small batches; every path's arithmetic is the same."""
    batch = pr.random_batch(1024, 10, seed0=0)
    ragged = pr.random_batch(300, "ragged", seed0=3)
    for bt in (batch, ragged):
        alone = gpu_ctx.solve_batch(bt, None)
        shared = gpu_ctx.solve_batch(bt, None, flags=api.FLAG_SHARED_DEVICE)
        for key in ("coeffs", "status", "cost", "times"):
            assert np.array_equal(alone[key], shared[key]), key
    nl = dict(time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=512)
    alone = gpu_ctx.solve_batch(ragged, None, **nl)
    shared = gpu_ctx.solve_batch(ragged, None, flags=api.FLAG_SHARED_DEVICE, **nl)
    for key in ("coeffs", "status", "times", "n_samples"):
        assert np.array_equal(alone[key], shared[key]), key
    for p in range(ragged.n_paths):  # (beyond a path's last sample the buffer is whatever the allocator handed out)
        n = min(int(alone["n_samples"][p]), 512)
        assert np.array_equal(alone["samples"][p, :n], shared["samples"][p, :n])


def test_issue_loop_in_c_round_robin_over_streams(gpu_ctx):
    """mrs_tg_bound_solve_launch_many: launch k goes to bound[k % n]; the same results as one launch per call, and the
    first failing solve's code comes back."""
    batch = pr.random_batch(128, 7, seed0=12)
    streams = [torch.cuda.Stream() for _ in range(3)]
    ctxs, plans, dbs, calls = [], [], [], []
    est = api.default_options(estimate_times=1)
    lin = api.default_options(flags=api.FLAG_SHARED_DEVICE)
    for st in streams:
        with torch.cuda.stream(st):
            c = api.Context(0)
            c.use_torch_stream()
            pl = api.Plan(c, batch.seg_offsets)
            db = api.DeviceBatch(batch, "cuda:0", sample_capacity=16)
            pl.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints,
                     limits=db.limits)
            ctxs.append(c), plans.append(pl), dbs.append(db)
            calls.append(pl.bind_solve(lin, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost))
    torch.cuda.synchronize()
    for db in dbs:
        db.coeffs.zero_()
    api.RoundRobin(calls)(7)   # 7 launches over 3 streams: every stream has run at least twice
    torch.cuda.synchronize()
    ref = gpu_ctx.solve_batch(batch, None)
    for db in dbs:
        assert np.array_equal(db.coeffs.cpu().numpy().reshape(ref["coeffs"].shape), ref["coeffs"])
        assert np.all(db.status.cpu().numpy() == 1)
    for pl in plans:
        pl.close()
    for c in ctxs:
        c.close()


def test_issue_loop_on_several_host_threads(gpu_ctx):
    """mrs_tg_bound_solve_launch_many_mt: the launches of a bound solve stay on one thread (stream order kept), the threads
    issue concurrently; same results as the single-threaded loop, errors still come back."""
    batch = pr.random_batch(200, 8, seed0=77)
    streams = [torch.cuda.Stream() for _ in range(4)]
    ctxs, plans, dbs, calls = [], [], [], []
    est = api.default_options(estimate_times=1)
    lin = api.default_options(flags=api.FLAG_SHARED_DEVICE)
    for st in streams:
        with torch.cuda.stream(st):
            c = api.Context(0)
            c.use_torch_stream()
            pl = api.Plan(c, batch.seg_offsets)
            db = api.DeviceBatch(batch, "cuda:0", sample_capacity=16)
            pl.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints,
                     limits=db.limits)
            ctxs.append(c), plans.append(pl), dbs.append(db)
            calls.append(pl.bind_solve(lin, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost))
    torch.cuda.synchronize()
    ref = gpu_ctx.solve_batch(batch, None)
    for threads in (2, 3, 4):   # (3 does not divide 4 bound solves: the library lowers it to 2)
        for db in dbs:
            db.coeffs.zero_()
        torch.cuda.synchronize()
        api.RoundRobin(calls, threads=threads)(41)
        torch.cuda.synchronize()
        for db in dbs:
            assert np.array_equal(db.coeffs.cpu().numpy().reshape(ref["coeffs"].shape), ref["coeffs"])
    for pl in plans:
        pl.close()
    for c in ctxs:
        c.close()


def test_pinned_block_lives_as_long_as_any_view_of_it(gpu_ctx):
    """the pinned block used to be freed when the FIRST array handed out died, although slices / ravel() /
    reshape of it (which numpy attaches to the underlying buffer object, not to that array) were still alive"""
    import gc
    import weakref
    a = api.pinned_copy(np.arange(4096.0))
    buf = a
    while isinstance(buf, np.ndarray):   # every view ends at the buffer object the block was wrapped in
        buf = buf.base
    owner = weakref.ref(buf._owner)
    del buf
    view = a.reshape(64, 64)[3:5].ravel()[8:16]
    expect = view.copy()
    del a
    gc.collect()
    assert owner() is not None, "the block was released while a view of it is alive"
    others = [api.pinned_copy(np.full(4096, -1.0)) for _ in range(8)]   # would land in a released block
    assert np.array_equal(view, expect)
    del view, others
    gc.collect()
    assert owner() is None, "the block is not released with its last view"


def test_pinned_pageable_and_mixed_host_buffers_give_identical_results(gpu_ctx):
    """mrs_tg_solve_batch moves every array the way its location allows (pinned: DMA in place; small pageable: packed
    through the context's staging block; large pageable: hipMemcpyAsync on the caller's pages).  Whatever the mix, the
    results are the same bits -- small batch (everything staged), large batch (coefficients not staged), pinned arrays,
    a registered pageable array, and outputs re-used across calls."""
    L = api.load_library()
    for P, n_seg in ((24, 6), (2048, 10)):
        batch = pr.random_batch(P, n_seg, seed0=300)
        kw = dict(time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=256)
        ref = gpu_ctx.solve_batch(batch, None, **kw)
        # all inputs and outputs pinned
        pb = pr.Batch(batch.seg_offsets, api.pinned_copy(batch.waypoints), api.pinned_copy(batch.fixed_mask),
                      api.pinned_copy(batch.fixed_values), api.pinned_copy(batch.limits), batch.derivative_to_optimize)
        out = dict(times=api.pinned_empty(batch.n_segments), coeffs=api.pinned_empty(ref["coeffs"].shape),
                   status=api.pinned_empty(P, np.int32), cost=api.pinned_empty(P), n_samples=api.pinned_empty(P, np.int32),
                   samples=api.pinned_empty(ref["samples"].shape))
        for k in out:
            out[k][...] = 0
        for rep in range(2):     # the second call re-uses the arenas and the cached plan
            got = gpu_ctx.solve_batch(pb, None, out=out, **kw)
            for k in ("times", "coeffs", "status", "cost", "n_samples"):
                assert np.array_equal(got[k], ref[k]), (P, k, rep)
            n = np.minimum(ref["n_samples"], 256)
            assert all(np.array_equal(got["samples"][p, :n[p]], ref["samples"][p, :n[p]]) for p in range(0, P, max(1, P // 64)))
        # mixed: pinned values, pageable everything else; and a pageable coefficient array pinned in place by registration
        mixed = pr.Batch(batch.seg_offsets, batch.waypoints, batch.fixed_mask, pb.fixed_values, batch.limits, batch.derivative_to_optimize)
        keep = gpu_ctx.solve_batch(mixed, None, **kw)
        assert np.array_equal(keep["coeffs"], ref["coeffs"]) and np.array_equal(keep["times"], ref["times"])
        assert L.mrs_tg_host_register(keep["coeffs"].ctypes.data, keep["coeffs"].nbytes) == 0
        keep["coeffs"][...] = 0
        again = gpu_ctx.solve_batch(mixed, None, out=keep, **kw)
        assert L.mrs_tg_host_unregister(keep["coeffs"].ctypes.data) == 0
        assert np.array_equal(again["coeffs"], ref["coeffs"]) and np.array_equal(again["status"], ref["status"])
    assert L.mrs_tg_host_register(None, 0) != 0 and b"nothing to register" in L.mrs_tg_last_error(None)


def test_two_bound_solves_of_one_context_are_issued_by_one_thread(gpu_ctx):
    """The multi-threaded issue loop partitions by CONTEXT: two bound solves that share a plan (and its workspaces) must not
    be driven by two threads at once -- with one context there is nothing to spread, and the results are those of the
    single-threaded loop; a NULL entry is refused with a message."""
    batch = pr.random_batch(300, 9, seed0=5)
    plan = api.Plan(gpu_ctx, batch.seg_offsets)
    dbs = [api.DeviceBatch(batch, "cuda:0", sample_capacity=64) for _ in range(2)]
    nl = api.default_options(time_alloc_method=api.TIME_ALLOC_MELLINGER, estimate_times=1)
    calls = [plan.bind_solve(nl, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints,
                             limits=db.limits) for db in dbs]
    ref = gpu_ctx.solve_batch(batch, None, time_alloc_method=api.TIME_ALLOC_MELLINGER)
    for threads in (1, 2, 4):
        for db in dbs:
            db.coeffs.zero_()
        api.RoundRobin(calls, threads=threads)(6)
        torch.cuda.synchronize()
        for db in dbs:
            assert np.array_equal(db.coeffs.cpu().numpy().reshape(ref["coeffs"].shape), ref["coeffs"]), threads
            assert np.array_equal(db.status.cpu().numpy(), ref["status"])
    L = api.load_library()
    arr = (C.c_void_p * 2)(calls[0].handle, None)
    assert L.mrs_tg_bound_solve_launch_many_mt(arr, 2, 4, 2) == -1 and b"bound solve 1 is NULL" in L.mrs_tg_last_error(None)
    assert L.mrs_tg_bound_solve_launch_many(arr, 2, 4) == -1
    plan.close()


def test_outer_loop_timing_spans_every_outer_loop_launch_and_the_gradient_free_modes(gpu_ctx):
    """Kernel family 2 under profiling: one pair of events from the first outer-loop launch of a call to the last (lean kernel
    + the general kernel for the paths it flags), and the gradient-free searches report a time again."""
    mixed = pr.random_mixed_batch(4096, seed0=9)          # > 2560 paths: lean kernel first, general kernel for stop_at / moving starts
    plain = pr.random_batch(4096, 10, seed0=9)
    gpu_ctx.set_profiling(True)
    try:
        gpu_ctx.solve_batch(plain, None, time_alloc_method=api.TIME_ALLOC_MELLINGER)
        ms_plain = gpu_ctx.last_kernel_ms(api.KERNEL_NONLINEAR)
        gpu_ctx.solve_batch(mixed, None, time_alloc_method=api.TIME_ALLOC_MELLINGER)
        ms_mixed = gpu_ctx.last_kernel_ms(api.KERNEL_NONLINEAR)
        small = pr.random_batch(64, 6, seed0=2)
        gpu_ctx.solve_batch(small, None, time_alloc_method=api.TIME_ALLOC_RICHTER_TIME)
        ms_dfo = gpu_ctx.last_kernel_ms(api.KERNEL_NONLINEAR)
    finally:
        gpu_ctx.set_profiling(False)
    assert 0.02 < ms_plain < 5.0 and 0.02 < ms_mixed < 20.0 and 0.01 < ms_dfo < 20.0, (ms_plain, ms_mixed, ms_dfo)
