"""Several batches in flight: independent contexts + plans on different HIP streams of one device run concurrently
(bench.py --in-flight) and must give bit-identical results to the same solves issued one after the other."""
import numpy as np
import pytest
import torch

from mrs_uav_trajectory_generation_amd import api, problem as pr

pytestmark = pytest.mark.gpu


def _lane(batch, stream):
    with torch.cuda.stream(stream):
        ctx = api.Context(0)
        ctx.use_torch_stream()
        plan = api.Plan(ctx, batch.seg_offsets)
        db = api.DeviceBatch(batch, "cuda:0", sample_capacity=256)
        est = api.default_options(derivative_to_optimize=4, estimate_times=1)
        plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints,
                   limits=db.limits)
    stream.synchronize()
    return ctx, plan, db, db.seg_times.clone()


@pytest.mark.parametrize("nonlinear", [False, True])
def test_concurrent_streams_match_serial_execution(gpu_ctx, nonlinear):
    batches = [pr.random_batch(1024, 10, seed0=50000), pr.random_batch(700, "ragged", seed0=51000),
               pr.random_batch(1024, 10, seed0=52000)]
    streams = [torch.cuda.Stream(device="cuda:0") for _ in batches]
    lanes = [_lane(b, s) for b, s in zip(batches, streams)]
    if nonlinear:
        opt = api.default_options(derivative_to_optimize=4, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2,
                                  sample_capacity=256)
    else:
        opt = api.default_options(derivative_to_optimize=4)
    calls = []
    for (ctx, plan, db, t0) in lanes:
        calls.append(plan.bind_solve(opt, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost,
                                     limits=db.limits, n_samples=db.n_samples, samples=db.samples))

    def run(order):
        for i in order:
            ctx, plan, db, t0 = lanes[i]
            with torch.cuda.stream(streams[i]):
                db.seg_times.copy_(t0)
            calls[i]()
        torch.cuda.synchronize()
        return [(l[2].coeffs.cpu().numpy().copy(), l[2].seg_times.cpu().numpy().copy(), l[2].status.cpu().numpy().copy(),
                 l[2].n_samples.cpu().numpy().copy()) for l in lanes]

    # serial reference: one lane at a time, each followed by a device-wide synchronisation
    serial = []
    for i in range(len(lanes)):
        serial.append(run([i])[i])
    # concurrent: 20 rounds over all lanes without any synchronisation in between
    got = run([i for _ in range(20) for i in range(len(lanes))])
    for (c, t, s, n), (c0, t0, s0, n0) in zip(got, serial):
        assert np.array_equal(s, s0) and np.array_equal(t, t0) and np.array_equal(c, c0)
        if nonlinear:
            assert np.array_equal(n, n0)
    for ctx, plan, db, t0 in lanes:
        plan.close()
        ctx.close()


def test_grouped_launch_packs_the_steps_of_a_round_into_one_dispatch(gpu_ctx):
    """mrs_tg_bound_solve_launch_group: bound solves of ONE plan with their own input / output arrays; the launches of a round go
    out as one kernel whose workgroups are divided among the batches.  Same results as one launch per step, bit for bit
    (1 and 2 paths per wavefront, a ragged batch, more launches than a multiple of the group, two plans in one run); what
    does not fit is refused with a message."""
    import ctypes as C
    for batch, n_slots, n_launches in ((pr.random_batch(300, 10, seed0=3), 4, 11), (pr.random_batch(1024, 10, seed0=0), 4, 8),
                                       (pr.random_batch(200, "ragged", seed0=8), 3, 7), (pr.random_batch(64, 5, seed0=1), 8, 8)):
        plan = api.Plan(gpu_ctx, batch.seg_offsets)
        est = api.default_options(estimate_times=1, derivative_to_optimize=4)
        lin = api.default_options(derivative_to_optimize=4)
        dbs, calls, refs = [], [], []
        for s in range(n_slots):
            # every slot solves its own problem: the same structure, other waypoints
            other = pr.random_batch(batch.n_paths, 10, seed0=1000 * (s + 1)) if batch.seg_offsets[-1] == 10 * batch.n_paths else batch
            b = pr.Batch(batch.seg_offsets, other.waypoints, other.fixed_mask, other.fixed_values, other.limits, 4) if other is not batch else batch
            db = api.DeviceBatch(b, "cuda:0", sample_capacity=16)
            plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints,
                       limits=db.limits)
            torch.cuda.synchronize()
            refs.append((db.coeffs.cpu().numpy().copy(), db.cost.cpu().numpy().copy()))
            db.coeffs.zero_()
            db.cost.zero_()
            db.status.zero_()
            dbs.append(db)
            calls.append(plan.bind_solve(lin, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost))
        api.RoundRobin(calls, grouped=True)(n_launches)
        torch.cuda.synchronize()
        for s, db in enumerate(dbs):
            touched = s < n_launches
            assert np.array_equal(db.coeffs.cpu().numpy(), refs[s][0] if touched else np.zeros_like(refs[s][0])), s
            assert np.array_equal(db.cost.cpu().numpy(), refs[s][1]) and np.all(db.status.cpu().numpy() == 1)
        # a solve that samples, or belongs to another plan, cannot be grouped
        smp = api.default_options(derivative_to_optimize=4, sampling_dt=0.2, sample_capacity=16)
        bad = plan.bind_solve(smp, dbs[0].fixed_mask, dbs[0].fixed_values, dbs[0].seg_times, dbs[0].coeffs, dbs[0].status, dbs[0].cost,
                              n_samples=dbs[0].n_samples, samples=dbs[0].samples)
        with pytest.raises(api.MrsTgError, match="fixed-times default solve without sampling"):
            api.RoundRobin([calls[0], bad], grouped=True)(2)
        # bound solves of another plan form their own dispatch: [A0, A1, B0] issues (A0 A1), (B0), (A0 A1), ...
        plan2 = api.Plan(gpu_ctx, batch.seg_offsets)
        last = dbs[-1]
        foreign = plan2.bind_solve(lin, last.fixed_mask, last.fixed_values, last.seg_times, last.coeffs, last.status, last.cost)
        for db in dbs:
            db.coeffs.zero_()
        api.RoundRobin([calls[0], calls[1], foreign], grouped=True)(6)
        torch.cuda.synchronize()
        for s in (0, 1, n_slots - 1):
            assert np.array_equal(dbs[s].coeffs.cpu().numpy(), refs[s][0]), s
        plan2.close()
        plan.close()


def test_a_bind_reserves_no_factor_stores_for_a_group_it_may_never_launch(gpu_ctx):
    """binding a fixed-times solve of a large plan reserved the factor stores of a 16-batch grouped launch
    (1.26 GB at 8192 x 10, ten at 65536 x 10) whether or not the solve was ever launched in a group.  Now a bind of such a
    plan reserves nothing, and mrs_tg_bound_solve_launch_group sizes the store for the group it launches (two batches here:
    2 x 78.6 MB), with the results of the single launches."""
    batch = pr.random_batch(8192, 10, seed0=4100)
    plan = api.Plan(gpu_ctx, batch.seg_offsets)
    db = api.DeviceBatch(batch, "cuda:0")
    est = api.default_options(derivative_to_optimize=4, estimate_times=1)
    plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints, limits=db.limits)
    lin = api.default_options(derivative_to_optimize=4)
    plan.solve(lin, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost)
    torch.cuda.synchronize()
    ref = db.coeffs.cpu().numpy().copy()
    c2 = torch.zeros_like(db.coeffs)
    free0 = torch.cuda.mem_get_info()[0]
    calls = [plan.bind_solve(lin, db.fixed_mask, db.fixed_values, db.seg_times, cc, db.status, db.cost) for cc in (db.coeffs, c2)]
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20, (free0 - free1) >> 20          # (round 4: 1.26 GB here)
    db.coeffs.zero_()
    api.RoundRobin(calls, grouped=True)(2)
    torch.cuda.synchronize()
    free2 = torch.cuda.mem_get_info()[0]
    assert free1 - free2 < 400 << 20, (free1 - free2) >> 20          # two stores of 78.6 MB (+ the pool's rounding), not sixteen
    assert np.array_equal(db.coeffs.cpu().numpy(), ref) and np.array_equal(c2.cpu().numpy(), ref)
    plan.close()
