"""One routing table, checked (round-5 VERDICT item 8).  mrs_tg_plan_explain runs the library's own launch functions dry --
every size rule, environment knob and hint as in a real call, kernels noted instead of enqueued -- so the route of a call can be
ASKED, and these tests pin it for every BASELINE config and for what a nodelet sends by default (min-acceleration, moving start,
stop_at, a request subdivided to 80 segments).  Each pinned route is also compared with the kernel trace of a REAL solve of the
same shape: the dry run and the launch cannot drift apart."""
import numpy as np
import pytest

from mrs_uav_trajectory_generation_amd import api, problem as pr

pytestmark = pytest.mark.gpu

MEL = dict(time_alloc_method=api.TIME_ALLOC_MELLINGER, estimate_times=1, sampling_dt=0.2, sample_capacity=512)


def _uniform(n, S):
    return (np.arange(n + 1, dtype=np.int64) * S).astype(np.int32)


def _has(names, *prefixes):
    return all(any(n.startswith(p) for n in names) for p in prefixes)


def test_routes_of_the_baseline_configs(gpu_ctx):
    # configs[0]: one 3-segment path, fixed times -- the one-lane-per-unknown solve
    p = api.Plan(gpu_ctx, _uniform(1, 3))
    assert p.explain(api.default_options(derivative_to_optimize=4)) == ["solve_rows_kernel<0>"]
    p.close()
    # configs[1]: 1024 x 10, fixed times: one batch = the rows kernel; the headline's dispatch of ten batches = the two-sided kernel
    p = api.Plan(gpu_ctx, _uniform(1024, 10))
    assert p.explain(api.default_options(derivative_to_optimize=4)) == ["solve_rows_kernel<0>"]
    assert p.explain(api.default_options(derivative_to_optimize=4), group_size=10) == ["solve_duo_group_kernel<false>"]
    assert p.explain(api.default_options(derivative_to_optimize=4, flags=api.FLAG_POSITIONS_ARE_WAYPOINTS), group_size=10) == \
        ["solve_duo_group_kernel<true>"]
    assert p.explain(api.default_options(derivative_to_optimize=4), group_size=4) == ["solve_rows_group_kernel"]
    assert p.explain(api.default_options(derivative_to_optimize=4, flags=api.FLAG_MATERIALIZED_BLOCKS))[0] == "assemble_blocks_uniform_kernel"
    # configs[2]: the same batch with the Mellinger outer loop, scaling and sampling: one wavefront per path, then the pipeline kernel
    r = p.explain(api.default_options(derivative_to_optimize=4, **MEL))
    assert r == ["optimize_wave_kernel", "solve_rows_pipeline_kernel"], r
    p.close()
    # configs[3]: 65536 x 10 (and its 8192-path shard): lean outer loop, four-lane solves at saturation / eight-lane below, own sampler
    p = api.Plan(gpu_ctx, _uniform(65536, 10))
    r = p.explain(api.default_options(derivative_to_optimize=4, **MEL))
    assert _has(r, "optimize_lean_shared", "solve_quad_kernel", "segment_maxima_scaling_kernel", "sample_"), r
    assert not _has(r, "solve_duo"), r
    p.close()
    p = api.Plan(gpu_ctx, _uniform(8192, 10))
    r = p.explain(api.default_options(derivative_to_optimize=4, **MEL))
    assert _has(r, "optimize_lean_shared", "solve_duo_kernel", "segment_maxima_scaling_kernel", "sample_"), r
    assert p.explain(api.default_options(derivative_to_optimize=4)) == ["solve_duo_kernel<false>"]
    p.close()
    # configs[4]: 8192 ragged 3..30: too long for the four- / eight-lane records, the rows kernel solves
    rag = pr.random_batch(8192, "ragged", seed0=0)
    p = api.Plan(gpu_ctx, rag.seg_offsets)
    r = p.explain(api.default_options(derivative_to_optimize=4, **MEL))
    assert _has(r, "optimize_lean", "solve_rows_kernel", "sample_") and not _has(r, "solve_quad") and not _has(r, "solve_duo"), r
    p.close()


def test_routes_of_what_a_nodelet_sends(gpu_ctx):
    # the default config: min-acceleration; a replanning request in flight starts from a moving state; stop_at waypoints; one
    # request after a few subdivision rounds has ~80 segments
    one = api.Plan(gpu_ctx, _uniform(1, 80))
    r = one.explain(api.default_options(derivative_to_optimize=2, **MEL))
    assert _has(r, "optimize_lean_shared_ends_long_kernel"), r
    one.close()
    # a server's batch of such requests under min-acceleration: the shared half sweeps with free end slots, the quad kernel's
    # ENDS instantiation at saturation
    p = api.Plan(gpu_ctx, _uniform(65536, 10))
    r = p.explain(api.default_options(derivative_to_optimize=2, **MEL))
    assert _has(r, "optimize_lean_shared_ends_kernel", "solve_quad_kernel<false, true>"), r
    r = p.explain(api.default_options(derivative_to_optimize=4, flags=api.FLAG_CONSTRAINED_SLOTS, **MEL))
    assert _has(r, "optimize_lean_shared_ends_kernel", "solve_quad_kernel<false, true>"), r
    p.close()
    # round 6: <= 1536 paths of 13-15 segments -- the plan's choice is the dimension split, whose kernel hands paths with free
    # end slots / constrained slots to its general step; such a CALL now goes to the lane groups (MRS_TG_REGROUP)
    p = api.Plan(gpu_ctx, _uniform(1024, 14))
    plain = p.explain(api.default_options(derivative_to_optimize=4, **MEL))
    below = p.explain(api.default_options(derivative_to_optimize=2, **MEL))
    hinted = p.explain(api.default_options(derivative_to_optimize=4, flags=api.FLAG_CONSTRAINED_SLOTS, **MEL))
    assert _has(plain, "optimize_split_kernel") or _has(plain, "optimize_wave_kernel"), plain
    assert _has(below, "optimize_lean") and not _has(below, "optimize_split_kernel"), below
    assert _has(hinted, "optimize_lean") and not _has(hinted, "optimize_split_kernel"), hinted
    p.close()


@pytest.mark.parametrize("shape,n,kw", [(10, 1024, dict(derivative_to_optimize=4)), (10, 1024, dict(derivative_to_optimize=4, **MEL)),
                                        (10, 8192, dict(derivative_to_optimize=4, **MEL)), ("ragged", 4200, dict(derivative_to_optimize=2, **MEL)),
                                        (14, 600, dict(derivative_to_optimize=2, **MEL)), (80, 40, dict(derivative_to_optimize=3, **MEL)),
                                        (6, 300, dict(derivative_to_optimize=4, time_alloc_method=1, estimate_times=1, sampling_dt=0.2,
                                                      sample_capacity=512))])
def test_the_dry_run_names_what_a_real_solve_launches(gpu_ctx, shape, n, kw):
    import torch
    batch = pr.random_batch(n, shape, seed0=31, derivative_to_optimize=kw["derivative_to_optimize"])
    plan = api.Plan(gpu_ctx, batch.seg_offsets)
    db = api.DeviceBatch(batch, "cuda:0", sample_capacity=512)
    opt = api.default_options(**kw)
    if not kw.get("estimate_times"):
        est = api.default_options(derivative_to_optimize=kw["derivative_to_optimize"], estimate_times=1)
        plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints, limits=db.limits)
    torch.cuda.synchronize()
    told = plan.explain(opt)
    api.kernel_trace_reset()
    plan.solve(opt, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints, limits=db.limits,
               n_samples=db.n_samples, samples=db.samples)
    ran = api.kernel_trace()
    torch.cuda.synchronize()
    # (a first sampling call also builds the accumulated-time table of its dt: a one-off kernel the dry run does not plan)
    ran = [k for k in ran if k != "sample_acc_table_kernel"]
    assert told == ran, (told, ran)
    plan.close()
