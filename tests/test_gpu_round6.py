"""Round 6: the advisor's findings of round 5 as tests, and the sampler of large launches against the oracle."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po
from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n_paths", [64, 6400])
def test_solve_batch_takes_positions_are_waypoints_with_explicit_times(gpu_ctx, n_paths):
    """mrs_tg_solve_batch used to upload the waypoints only for the time estimate, so the public flag
    MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS with caller-given segment times was refused ("needs the waypoints array") although the
    caller had passed them.  Fixed times and Mellinger from given times, below and above the 6144 paths from which the flag
    changes the kernel's loads: same bits as without the flag."""
    batch = pr.random_batch(n_paths, 10, seed0=123)
    t = util.oracle_times(batch)
    plain = gpu_ctx.solve_batch(batch, t)
    flagged = gpu_ctx.solve_batch(batch, t, flags=api.FLAG_POSITIONS_ARE_WAYPOINTS)
    assert np.all(flagged["status"] == 1)
    assert np.array_equal(plain["coeffs"], flagged["coeffs"]) and np.array_equal(plain["cost"], flagged["cost"])
    nl_plain = gpu_ctx.solve_batch(batch, t, time_alloc_method=api.TIME_ALLOC_MELLINGER)
    nl_flag = gpu_ctx.solve_batch(batch, t, time_alloc_method=api.TIME_ALLOC_MELLINGER, flags=api.FLAG_POSITIONS_ARE_WAYPOINTS)
    assert np.array_equal(nl_plain["times"], nl_flag["times"]) and np.array_equal(nl_plain["status"], nl_flag["status"])
    assert np.array_equal(nl_plain["coeffs"], nl_flag["coeffs"])


def test_kernel_trace_with_a_small_array_returns_the_newest_names(gpu_ctx):
    batch = pr.random_batch(16, 5, seed0=1)
    api.kernel_trace_reset()
    gpu_ctx.solve_batch(batch, None, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=256)
    full = api.kernel_trace()
    assert len(full) >= 3
    buf = (C.c_char_p * 2)()
    n = api.load_library().mrs_tg_kernel_trace(buf, 2)
    assert n == 2 and [buf[i].decode() for i in range(2)] == full[-2:]


def test_verify_flags_catches_a_statement_that_became_false():
    """MRS_TG_VERIFY_FLAGS=1 (read once per process: a child process): a bound solve whose value array is rewritten in place so
    that MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS no longer holds is refused at the next mrs_tg_plan_solve instead of being
    solved as the waypoints' problem."""
    code = r"""
import sys
import numpy as np
import torch
sys.path.insert(0, %r)
from mrs_uav_trajectory_generation_amd import api, problem as pr
ctx = api.Context(0)
ctx.use_torch_stream()
batch = pr.random_batch(64, 10, seed0=5)
plan = api.Plan(ctx, batch.seg_offsets)
db = api.DeviceBatch(batch, torch.device("cuda", 0), sample_capacity=0)
est = api.default_options(derivative_to_optimize=4, estimate_times=1)
plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints, limits=db.limits)
opt = api.default_options(derivative_to_optimize=4, flags=api.FLAG_POSITIONS_ARE_WAYPOINTS)
plan.solve(opt, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints)
torch.cuda.synchronize()
print("first ok")
db.fixed_values.view(-1)[0] += 1.0      # the position constraint of vertex 0 is no longer its waypoint
try:
    plan.solve(opt, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints)
    print("second accepted")
except api.MrsTgError as ex:
    print("second refused:", ex)
""" % ROOT
    env = dict(os.environ, MRS_TG_VERIFY_FLAGS="1")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "first ok" in p.stdout and "second refused" in p.stdout and "1 vertices" in p.stdout, p.stdout


@pytest.mark.parametrize("shape,n_paths,dt,cap,kernel", [(10, 8192, 0.2, 512, "sample_group_kernel<8"), ("ragged", 4200, 0.2, 640, "sample_group_kernel<16"),
                                                         (10, 4608, 0.05, 1024, "sample_group_kernel<16"), (3, 9000, 0.3, 96, "sample_group_kernel<8"),
                                                         (15, 8200, 0.2, 300, "sample_group_kernel<8"), (30, 4100, 0.2, 2048, "sample_group_kernel<16")])
def test_group_sampler_against_the_oracles_walk(gpu_ctx, monkeypatch, shape, n_paths, dt, cap, kernel):
    """The sampler of large launches (sample_group_kernel: 8 or 16 lanes per path, walk and evaluation fused) against the oracle's
    restatement of Trajectory::evaluateRange on the SAME coefficients and times: sample counts equal (capacity + 1 where the
    trajectory does not fit), positions to 1e-11, heading to 1e-11 on the circle -- and the kernel trace says which kernel ran."""
    # (not the default: measured slower than the one-wavefront-per-path kernel, profiles/round6_sampler_group_ab.txt)
    monkeypatch.setenv("MRS_TG_SAMPLE_GROUP", "8" if "<8" in kernel else "16")
    batch = pr.random_batch(n_paths, shape, seed0=555)
    lin = gpu_ctx.solve_batch(batch, None)
    times = lin["times"].copy()
    times[:batch.seg_offsets[1]] = 20 * dt / (batch.seg_offsets[1])   # path 0 ends exactly on a sample time (in exact arithmetic)
    api.kernel_trace_reset()
    out = gpu_ctx.solve_batch(batch, times, sampling_dt=dt, sample_capacity=cap)
    trace = api.kernel_trace()
    assert any(k.startswith(kernel) for k in trace), trace
    so = batch.seg_offsets
    checked = overflow = 0
    for p in list(range(0, n_paths, max(1, n_paths // 150))) + [0, n_paths - 1]:
        a, b = so[p], so[p + 1]
        s, n = po.sample_trajectory(out["coeffs"][a:b], out["times"][a:b], dt, 0, cap)
        assert min(n, cap + 1) == out["n_samples"][p], (p, n, out["n_samples"][p])
        overflow += n > cap
        n = min(n, cap)
        got = out["samples"][p, :n]
        assert np.max(np.abs(got[:, :3] - s[:n, :3])) < 1e-11, p
        yaw = np.array([po.wrap_yaw(y) for y in s[:n, 3]])
        dy = np.abs(got[:, 3] - yaw)
        assert np.max(np.minimum(dy, 2 * np.pi - dy)) < 1e-11, p
        checked += 1
    assert checked > 100
    print("sampler %s x %d dt %g capacity %d: %d paths checked, %d overflow the capacity" % (shape, n_paths, dt, cap, checked, overflow))


@pytest.mark.parametrize("shape,n_paths,moving", [(10, 6400, False), ("short", 7000, False), (2, 6400, False), (3, 8192, False),
                                                  (11, 6400, True), (15, 6200, False)])
def test_two_sided_solve_against_the_oracle(gpu_ctx, shape, n_paths, moving):
    """solve_duo_kernel: eight lanes per path (side x dimension), the vertex chain eliminated from both ends towards the middle
    vertex -- the launch shape of dispatches that would leave SIMDs idle as quad wavefronts (6144 .. 20479 paths).  Against the
    oracle in the reference's arithmetic (1e-7; 1e-6 on paths with a segment below 0.5 s, where that route itself is off) and in
    113 bits (the HIP path's own error: median below 1e-13, worst 1e-8), costs included; continuity and constraints on every path."""
    if shape == "short":   # ragged, 3 .. 15 segments (the LDS record of the four- and eight-lane kernels holds 15)
        rag = pr.random_batch(3 * n_paths, "ragged", seed0=777)
        keep = [p for p in range(rag.n_paths) if rag.seg_offsets[p + 1] - rag.seg_offsets[p] <= 15][:n_paths]
        assert len(keep) == n_paths
        batch = rag.select(keep)
    else:
        batch = pr.random_batch(n_paths, shape, seed0=777)
    if moving:
        from tests.test_gpu_large_batches import _moving
        batch = _moving(batch, seed=3)
    api.kernel_trace_reset()
    out = gpu_ctx.solve_batch(batch, None)
    trace = api.kernel_trace()
    assert any(k.startswith("solve_duo_kernel") for k in trace), trace
    assert np.all(out["status"] == 1)
    so = batch.seg_offsets
    idx = list(range(0, n_paths, max(1, n_paths // 257)))
    sub = batch.select(idx)
    t = np.concatenate([out["times"][so[p]:so[p + 1]] for p in idx])
    got = np.concatenate([out["coeffs"][so[p]:so[p + 1]] for p in idx])
    ref = util.oracle_linear(sub, t)
    po.lib().mto_set_arithmetic(po.QUAD_PRECISION)
    try:
        refq = util.oracle_linear(sub, t)
    finally:
        po.lib().mto_set_arithmetic(po.REFERENCE_ARITHMETIC)
    errs, errq = [], []
    for k in range(len(idx)):
        a, b = sub.seg_offsets[k], sub.seg_offsets[k + 1]
        e = util.coeff_error(got[a:b], ref["coeffs"][a:b])
        assert e < (1e-7 if t[a:b].min() >= 0.5 else 1e-6), (idx[k], e)
        errq.append(util.coeff_error(got[a:b], refq["coeffs"][a:b]))
        errs.append(e)
    print("ERR duo %s x %d moving %s: vs oracle max %.2e; vs 113-bit median %.2e max %.2e" % (shape, n_paths, moving, max(errs),
                                                                                       float(np.median(errq)), max(errq)))
    assert np.median(errq) < 1e-12 and max(errq) < 1e-8
    assert np.max(np.abs(out["cost"][idx] - refq["cost"]) / np.abs(refq["cost"])) < 1e-9
    chk = list(range(0, n_paths, 13))
    csub = batch.select(chk)
    tc = np.concatenate([out["times"][so[p]:so[p + 1]] for p in chk])
    cc = np.concatenate([out["coeffs"][so[p]:so[p + 1]] for p in chk])
    assert util.continuity_defect(csub, cc, tc) < 1e-9 and util.constraint_defect(csub, cc, tc) < 1e-9
