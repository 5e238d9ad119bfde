"""Error behaviour at the C-ABI boundary: every misuse is reported through the return code and
mrs_tg_last_error(), nothing throws or crashes, and a failed call leaves the context usable
(include/mrs_tg.h: "no exceptions across the ABI, errors via return codes")."""
import ctypes as C

import numpy as np
import pytest

from mrs_uav_trajectory_generation_amd import api, problem as pr

pytestmark = pytest.mark.gpu

OK, INVALID = 0, -1


def _raw_solve(ctx, batch, opt, *, so=None, n_paths=None, mask=True, vals=True, times=True, coeffs=True, status=True,
               n_samples=None):
    L = ctx._L
    nS, P = batch.n_segments, batch.n_paths
    t = np.ones(nS)
    c = np.zeros((nS, 4, 10))
    st = np.zeros(P, dtype=np.int32)
    so = batch.seg_offsets if so is None else so
    return L.mrs_tg_solve_batch(ctx._h, P if n_paths is None else n_paths, api._np_ptr(so), api._np_ptr(batch.waypoints),
                                api._np_ptr(batch.fixed_mask) if mask else None,
                                api._np_ptr(batch.fixed_values) if vals else None, api._np_ptr(batch.limits),
                                C.byref(opt), api._np_ptr(t) if times else None, api._np_ptr(c) if coeffs else None,
                                api._np_ptr(st) if status else None, None, n_samples, None)


def _err(ctx):
    return ctx._L.mrs_tg_last_error(ctx._h).decode()


def test_argument_validation(gpu_ctx):
    batch = pr.random_batch(3, 4, seed0=5)
    opt = api.default_options(derivative_to_optimize=4)
    assert _raw_solve(gpu_ctx, batch, opt) == OK
    for kw, text in ((dict(mask=False), "required"), (dict(vals=False), "required"), (dict(times=False), "required"),
                     (dict(coeffs=False), "required"), (dict(status=False), "required")):
        assert _raw_solve(gpu_ctx, batch, opt, **kw) == INVALID and text in _err(gpu_ctx)
    assert _raw_solve(gpu_ctx, batch, opt, n_paths=-1) == INVALID and "negative" in _err(gpu_ctx)
    bad = batch.seg_offsets.copy()
    bad[0] = 1
    assert _raw_solve(gpu_ctx, batch, opt, so=bad) == INVALID and "seg_offsets[0]" in _err(gpu_ctx)
    bad = batch.seg_offsets.copy()
    bad[2] = bad[1]                                              # a path with zero segments
    assert _raw_solve(gpu_ctx, batch, opt, so=bad) == INVALID and "segments" in _err(gpu_ctx)
    assert _raw_solve(gpu_ctx, batch, opt, n_paths=0) == OK      # an empty batch is not an error
    for d in (-1, 5):
        o = api.default_options(derivative_to_optimize=d)
        assert _raw_solve(gpu_ctx, batch, o) == INVALID and "derivative_to_optimize" in _err(gpu_ctx)
    o = api.default_options(derivative_to_optimize=4, sampling_dt=0.2, sample_capacity=16)
    assert _raw_solve(gpu_ctx, batch, o) == INVALID and "n_samples_out" in _err(gpu_ctx)
    o = api.default_options(derivative_to_optimize=4, sampling_dt=0.2, sample_capacity=-3)
    assert _raw_solve(gpu_ctx, batch, o) == INVALID and "sample_capacity" in _err(gpu_ctx)
    assert gpu_ctx._L.mrs_tg_solve_batch(None, 0, None, None, None, None, None, None, None, None, None, None, None, None) == INVALID
    # the context still works
    out = gpu_ctx.solve_batch(batch, None)
    assert np.all(out["status"] == 1)


def test_modes_that_need_limits_say_so(gpu_ctx):
    batch = pr.random_batch(2, 3, seed0=6)
    L = gpu_ctx._L
    t = np.ones(batch.n_segments)
    c = np.zeros((batch.n_segments, 4, 10))
    st = np.zeros(2, dtype=np.int32)
    for mode in (0, 1, 2, 3, 4):
        opt = api.default_options(derivative_to_optimize=4, time_alloc_method=mode)
        rc = L.mrs_tg_solve_batch(gpu_ctx._h, 2, api._np_ptr(batch.seg_offsets), api._np_ptr(batch.waypoints),
                                  api._np_ptr(batch.fixed_mask), api._np_ptr(batch.fixed_values), None, C.byref(opt),
                                  api._np_ptr(t), api._np_ptr(c), api._np_ptr(st), None, None, None)
        assert rc == INVALID and "limits" in _err(gpu_ctx), mode


def test_find_trajectory_argument_validation(gpu_ctx):
    with pytest.raises(api.MrsTgError, match="at least 2 waypoints"):
        gpu_ctx.find_trajectory(pr.CONFIG1_WAYPOINTS[:1])
    with pytest.raises(api.MrsTgError, match="derivative_to_optimize"):
        gpu_ctx.find_trajectory(pr.CONFIG1_WAYPOINTS, derivative_to_optimize=1)
    r = gpu_ctx.find_trajectory(pr.CONFIG1_WAYPOINTS)
    assert r["status"] >= 1
