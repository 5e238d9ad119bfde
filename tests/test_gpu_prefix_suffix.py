"""The prefix / suffix evaluation of the Mellinger gradient (optimize_ps_kernel, cost_gradient_ps_kernel; MRS_TG_PS=1):
every perturbed time vector shares the sweeps over B' = max(T - h/(S-1), 0.01), so an evaluation costs ~5 S elimination
steps instead of S (S + 1).  Same J and gradient as the sweeping kernels and as the oracle
(getCostAndGradientMellinger, /root/reference/include/eth_trajectory_generation/impl/polynomial_optimization_nonlinear_impl.h:257-333);
paths it does not take (moving start, stop vertices, fewer than four segments) fall through to the sweeping kernel."""
import numpy as np
import pytest
import torch

from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po
from tests import util

pytestmark = pytest.mark.gpu


def _gradient(ctx, batch, t):
    plan = api.Plan(ctx, batch.seg_offsets)
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    cost = torch.zeros(batch.n_paths, dtype=torch.float64, device="cuda")
    grad = torch.zeros(batch.n_segments, dtype=torch.float64, device="cuda")
    plan.cost_gradient(batch.derivative_to_optimize, dv(batch.fixed_mask), dv(batch.fixed_values), dv(t), cost, grad)
    torch.cuda.synchronize()
    plan.close()
    return cost.cpu().numpy(), grad.cpu().numpy()


@pytest.mark.parametrize("n_seg", [4, 5, 10, 11, "ragged"])
def test_same_cost_and_gradient_as_the_sweeps_and_the_oracle(gpu_ctx, monkeypatch, n_seg):
    batch = pr.random_batch(70, n_seg, seed0=300)
    t = util.oracle_times(batch)
    monkeypatch.setenv("MRS_TG_PS", "0")
    J0, g0 = _gradient(gpu_ctx, batch, t)
    monkeypatch.setenv("MRS_TG_PS", "1")
    J1, g1 = _gradient(gpu_ctx, batch, t)
    # (the cost is a difference that cancels 1-4 digits, the gradient a difference of such costs: DESIGN.md section 5)
    assert np.max(np.abs(J1 - J0) / np.abs(J0)) < 1e-9
    for p in range(batch.n_paths):
        a, b = batch.seg_offsets[p], batch.seg_offsets[p + 1]
        assert np.max(np.abs(g1[a:b] - g0[a:b])) <= 1e-7 * np.max(np.abs(g0[a:b])), p
    so = batch.seg_offsets
    for p in range(0, batch.n_paths, 7):
        _, m, v = batch.path(p)
        Jo, go = po.cost_and_gradient(4, m, v, t[so[p]:so[p + 1]])
        assert abs(J1[p] - Jo) <= 1e-8 * abs(Jo)
        assert np.max(np.abs(g1[so[p]:so[p + 1]] - go)) <= 1e-6 * np.max(np.abs(go))


def test_paths_it_does_not_take_fall_through(gpu_ctx, monkeypatch):
    """mixed constraint patterns (moving starts, stop vertices, 1..30 segments): identical results either way"""
    batch = pr.random_mixed_batch(300, 4, seed0=5)
    t = util.oracle_times(batch)
    monkeypatch.setenv("MRS_TG_PS", "0")
    J0, g0 = _gradient(gpu_ctx, batch, t)
    monkeypatch.setenv("MRS_TG_PS", "1")
    J1, g1 = _gradient(gpu_ctx, batch, t)
    assert np.max(np.abs(J1 - J0) / np.maximum(np.abs(J0), 1e-12)) < 1e-9
    for p in range(batch.n_paths):
        a, b = batch.seg_offsets[p], batch.seg_offsets[p + 1]
        assert np.max(np.abs(g1[a:b] - g0[a:b])) <= 1e-7 * max(np.max(np.abs(g0[a:b])), 1e-12), p


def test_outer_loop_on_the_prefix_suffix_evaluation(gpu_ctx, monkeypatch):
    batch = pr.random_batch(512, 10, seed0=40)
    nl = dict(time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=512)
    monkeypatch.setenv("MRS_TG_PS", "0")
    a = gpu_ctx.solve_batch(batch, None, **nl)
    monkeypatch.setenv("MRS_TG_PS", "1")
    b = gpu_ctx.solve_batch(batch, None, **nl)
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["n_samples"], b["n_samples"])
    so = batch.seg_offsets
    dt = np.array([np.max(np.abs(a["times"][so[p]:so[p + 1]] - b["times"][so[p]:so[p + 1]]) / a["times"][so[p]:so[p + 1]])
                   for p in range(batch.n_paths)])
    assert np.mean(dt < 1e-6) >= 0.99, np.mean(dt < 1e-6)  # (a line search may branch on a 1e-12 difference in J)
