"""Row a10 in full: any fixed / free pattern of the vertex constraints, position-free vertices included
(setupConstraintReorderingMatrix, /root/reference/include/eth_trajectory_generation/impl/polynomial_optimization_linear_impl.h:184-257).
The fast kernels return such a path with status -2; the general kernel (mrs_tg_general.hip, MRS_TG_FLAG_GENERAL_PATTERNS)
solves it with 5 x 5 vertex blocks.  Checked against the oracle's reference-style route (dense R = C^T H C, QR) and against
the problem's own optimality conditions."""
import numpy as np
import pytest
import torch

from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po
from tests import util

pytestmark = pytest.mark.gpu


def _free_some_positions(batch, rng, share=0.3):
    """interior vertices lose their position constraint with probability `share`; some also fix a velocity instead"""
    m = batch.fixed_mask.copy()
    v = batch.fixed_values.copy()
    so = batch.seg_offsets
    touched = []
    for p in range(batch.n_paths):
        v0 = so[p] + p
        S = so[p + 1] - so[p]
        hit = False
        for k in range(1, S):
            if rng.random() < share:
                m[v0 + k, 0] = 0
                hit = True
                if rng.random() < 0.4:  # a velocity pinned where the position floats
                    m[v0 + k, 1] = 1
                    v[v0 + k, 1, :] = rng.normal(size=4)
        if hit:
            touched.append(p)
    return pr.Batch(batch.seg_offsets, batch.waypoints, m, v, batch.limits, batch.derivative_to_optimize), touched


@pytest.mark.parametrize("d", [4, 3, 2])
@pytest.mark.parametrize("n_seg", [5, "ragged"])
def test_position_free_vertices_match_the_oracle(gpu_ctx, n_seg, d):
    rng = np.random.default_rng(100 + d)
    base = pr.random_batch(60, n_seg, seed0=700, derivative_to_optimize=d)
    batch, touched = _free_some_positions(base, rng)
    assert len(touched) > 20
    t = util.oracle_times(base)
    out = gpu_ctx.solve_batch(batch, t)  # the host interface sees the masks and switches the general solver on
    assert np.all(out["status"] == 1)
    so = batch.seg_offsets
    for p in range(batch.n_paths):
        _, m, v = batch.path(p)
        oc = po.solve_linear(d, m, v, t[so[p]:so[p + 1]])
        scale = np.max(np.abs(oc))
        assert np.max(np.abs(out["coeffs"][so[p]:so[p + 1]] - oc)) <= 2e-7 * scale, (p, p in touched)
        Jo = po.compute_cost(d, t[so[p]:so[p + 1]], oc)
        assert abs(out["cost"][p] - Jo) <= 1e-7 * max(abs(Jo), 1e-9), p
    # the untouched paths are the fast kernels' results, bit for bit
    ref = gpu_ctx.solve_batch(base, t)
    for p in range(batch.n_paths):
        if p not in touched:
            assert np.array_equal(out["coeffs"][so[p]:so[p + 1]], ref["coeffs"][so[p]:so[p + 1]])
    # continuity of derivatives 0..4 at every interior vertex, fixed values met
    assert util.continuity_defect(batch, out["coeffs"], t) < 1e-7
    assert util.constraint_defect(batch, out["coeffs"], t) < 1e-7


def test_freeing_a_position_can_only_lower_the_cost(gpu_ctx):
    base = pr.random_batch(40, 6, seed0=40)
    t = util.oracle_times(base)
    m = base.fixed_mask.copy()
    for p in range(base.n_paths):
        m[base.seg_offsets[p] + p + 3, 0] = 0  # vertex 3 of every path
    freed = pr.Batch(base.seg_offsets, base.waypoints, m, base.fixed_values, base.limits)
    a = gpu_ctx.solve_batch(base, t)
    b = gpu_ctx.solve_batch(freed, t)
    assert np.all(b["status"] == 1)
    assert np.all(b["cost"] <= a["cost"] * (1 + 1e-12))
    assert np.mean(b["cost"] < 0.999 * a["cost"]) > 0.9


def test_device_interface_needs_the_flag_and_nonlinear_modes_refuse(gpu_ctx):
    base = pr.random_batch(8, 5, seed0=9)
    m = base.fixed_mask.copy()
    m[base.seg_offsets[2] + 2 + 2, 0] = 0  # path 2, vertex 2
    batch = pr.Batch(base.seg_offsets, base.waypoints, m, base.fixed_values, base.limits)
    plan = api.Plan(gpu_ctx, batch.seg_offsets)
    db = api.DeviceBatch(batch, "cuda:0", sample_capacity=256)
    db.seg_times.copy_(torch.from_numpy(util.oracle_times(base)))
    plan.solve(api.default_options(), db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost)
    torch.cuda.synchronize()
    assert db.status.cpu().numpy().tolist() == [1, 1, -2, 1, 1, 1, 1, 1]
    opt = api.default_options(flags=api.FLAG_GENERAL_PATTERNS, sampling_dt=0.2, sample_capacity=256)
    plan.solve(opt, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, n_samples=db.n_samples,
               samples=db.samples)
    torch.cuda.synchronize()
    assert np.all(db.status.cpu().numpy() == 1) and np.all(db.n_samples.cpu().numpy() > 5)
    plan.close()
    out = gpu_ctx.solve_batch(batch, None, time_alloc_method=api.TIME_ALLOC_MELLINGER)
    assert out["status"][2] < 0 and np.all(np.delete(out["status"], 2) >= 1)
