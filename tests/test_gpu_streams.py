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
