"""mrs_tg_plan_sample_states: sampleWholeTrajectory with every field of the sampled state
(/root/reference/src/eth_trajectory_generation/trajectory_sampling.cpp:49-104) against the oracle's evaluateRange for each
derivative order, through the C ABI on the GPU."""
import numpy as np
import pytest
import torch

from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_seg,dt", [(10, 0.2), ("ragged", 0.2), (6, 0.05)])
def test_sampled_states_match_the_oracle_for_every_derivative_order(gpu_ctx, n_seg, dt):
    batch = pr.random_batch(48, n_seg, seed0=77)
    cap = 4096
    # a solved trajectory with its own positions + heading samples (order 0 of the states, bit for bit)
    out = gpu_ctx.solve_batch(batch, None, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=dt, sample_capacity=cap)
    plan = api.Plan(gpu_ctx, batch.seg_offsets)
    coeffs = torch.from_numpy(out["coeffs"]).cuda()
    times = torch.from_numpy(out["times"]).cuda()
    n_dev = torch.zeros(batch.n_paths, dtype=torch.int32, device="cuda")
    states = torch.full((batch.n_paths, cap, api.STATE_ORDERS, 4), float("nan"), dtype=torch.float64, device="cuda")
    plan.sample_states(coeffs, times, dt, cap, n_dev, states)
    torch.cuda.synchronize()
    n_got = n_dev.cpu().numpy()
    st = states.cpu().numpy()
    assert np.array_equal(n_got, out["n_samples"])
    for p in range(batch.n_paths):
        a, b = batch.seg_offsets[p], batch.seg_offsets[p + 1]
        n = min(int(n_got[p]), cap)
        assert np.array_equal(st[p, :n, 0, :], out["samples"][p, :n])
        assert np.all(np.isnan(st[p, n:]))  # nothing is written beyond the last sample
        for k in range(api.STATE_ORDERS):
            ref, n_ref = po.sample_trajectory(out["coeffs"][a:b], out["times"][a:b], dt, k, cap)
            assert min(n_ref, cap + 1) == n_got[p]
            got = st[p, :n, k, :]
            scale = max(1.0, float(np.max(np.abs(ref))))
            cols = slice(0, 3) if k == 0 else slice(0, 4)  # yaw (order 0, column 3) is wrapped: compared below
            assert np.max(np.abs(got[:, cols] - ref[:n, cols])) < 1e-11 * scale, (p, k)
            if k == 0:
                yaw = np.array([po.wrap_yaw(y) for y in ref[:n, 3]])
                dy = np.abs(got[:, 3] - yaw)
                assert np.max(np.minimum(dy, 2 * np.pi - dy)) < 1e-11
    plan.close()


def test_sampled_states_are_consistent_derivatives(gpu_ctx):
    """Order k + 1 is the derivative of order k: central differences of the samples inside a segment agree."""
    batch = pr.random_batch(4, 3, seed0=5)
    out = gpu_ctx.solve_batch(batch, None)
    plan = api.Plan(gpu_ctx, batch.seg_offsets)
    dt, cap = 1e-3, 20000
    n_dev = torch.zeros(batch.n_paths, dtype=torch.int32, device="cuda")
    states = torch.zeros((batch.n_paths, cap, api.STATE_ORDERS, 4), dtype=torch.float64, device="cuda")
    plan.sample_states(torch.from_numpy(out["coeffs"]).cuda(), torch.from_numpy(out["times"]).cuda(), dt, cap, n_dev, states)
    torch.cuda.synchronize()
    st = states.cpu().numpy()
    t0 = out["times"][0]
    n_in = int(0.9 * t0 / dt)  # samples well inside the first segment of path 0
    for k in range(api.STATE_ORDERS - 1):
        num = (st[0, 2:n_in, k, :3] - st[0, :n_in - 2, k, :3]) / (2 * dt)
        ana = st[0, 1:n_in - 1, k + 1, :3]
        assert np.max(np.abs(num - ana)) < 1e-4 * max(1.0, np.max(np.abs(ana))), k
    plan.close()


def test_sample_states_count_only_and_argument_checks(gpu_ctx):
    batch = pr.random_batch(3, 4, seed0=9)
    out = gpu_ctx.solve_batch(batch, None, sampling_dt=0.2, sample_capacity=512)
    plan = api.Plan(gpu_ctx, batch.seg_offsets)
    coeffs = torch.from_numpy(out["coeffs"]).cuda()
    times = torch.from_numpy(out["times"]).cuda()
    n_dev = torch.zeros(batch.n_paths, dtype=torch.int32, device="cuda")
    plan.sample_states(coeffs, times, 0.2, 0, n_dev, None)  # capacity 0: counts only, "more than fit" = 1
    torch.cuda.synchronize()
    assert np.all(n_dev.cpu().numpy() == 1)
    with pytest.raises(api.MrsTgError):
        plan.sample_states(coeffs, times, 0.0, 16, n_dev, torch.zeros(1, device="cuda", dtype=torch.float64))
    with pytest.raises(api.MrsTgError):
        plan.sample_states(coeffs, times, 0.2, 16, n_dev, None)
    plan.close()


def test_sampled_states_of_a_large_batch_equal_those_of_its_parts(gpu_ctx):
    """a path's sampled states do not depend on the batch it travels in: 4100 ragged paths in one call against the same paths
    in batches of 1000, every derivative order of every sample bit for bit, counts-only call included.  (Round 4 built a
    sampler with eight paths per wavefront and lane-private walks for large batches; bit-identical by this test and SLOWER --
    8192 x 10: 52 -> 150 us, 65536 x 10: 327 -> 1080 us: the lane-private walk's control flow compiles to ~90 instructions per
    sample, against ~5 of the chunked 64-lane walk.  Not kept; HISTORY.md.)"""
    batch = pr.random_batch(4100, "ragged", seed0=88)
    out = gpu_ctx.solve_batch(batch, None)
    dt, cap = 0.25, 96
    so = batch.seg_offsets

    def states_of(sub, coeffs, times, with_states=True):
        plan = api.Plan(gpu_ctx, sub.seg_offsets)
        n_dev = torch.zeros(sub.n_paths, dtype=torch.int32, device="cuda")
        st = torch.full((sub.n_paths, cap, api.STATE_ORDERS, 4), float("nan"), dtype=torch.float64, device="cuda") if with_states else None
        plan.sample_states(torch.from_numpy(coeffs).cuda(), torch.from_numpy(times).cuda(), dt, cap if with_states else 0, n_dev, st)
        torch.cuda.synchronize()
        plan.close()
        return n_dev.cpu().numpy(), (st.cpu().numpy() if with_states else None)
    n_big, st_big = states_of(batch, out["coeffs"], out["times"])
    assert (n_big == cap + 1).any() and (n_big <= cap).any()
    n_cnt, _ = states_of(batch, out["coeffs"], out["times"], with_states=False)
    assert np.all(n_cnt == 1)   # capacity 0: "more than fit"
    for a in range(0, batch.n_paths, 1000):
        idx = list(range(a, min(a + 1000, batch.n_paths)))
        sub = batch.select(idx)
        n_small, st_small = states_of(sub, out["coeffs"][so[idx[0]]:so[idx[-1] + 1]], out["times"][so[idx[0]]:so[idx[-1] + 1]])
        assert np.array_equal(n_small, n_big[idx])
        assert np.array_equal(st_small, st_big[idx], equal_nan=True)
