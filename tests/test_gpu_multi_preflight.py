"""Multi-GPU pre-flight on the one GPU of the box: BASELINE configs[3]'s per-GPU shard size (8192 random 10-segment paths,
Mellinger outer loop + feasibility scaling + sampling) through the C ABI's device list (mrs_tg_create_multi /
mrs_tg_multi_solve_batch, two contexts on device 0, one host thread each), and the sharded result against the ORACLE -- the
other multi-device tests compare with the single-device solve only, at <= 41 paths."""
import numpy as np
import pytest

from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po
from tests import util

pytestmark = pytest.mark.gpu
CAP = 192


def test_config3_shard_over_a_device_list_against_the_oracle(gpu_ctx):
    batch = pr.random_batch(8192, 10, seed0=0)
    multi = api.MultiContext([0, 0])
    shard = multi.shard(batch.seg_offsets)
    sizes = np.bincount(shard, minlength=2)
    assert sizes.tolist() == [4096, 4096] and np.all(np.diff(shard) >= 0)   # contiguous halves
    opts = dict(time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2, sample_capacity=CAP)
    many = multi.solve_batch(batch, None, **opts)
    multi.close()
    so = batch.seg_offsets
    assert np.all(np.isin(many["status"], (1, 3, 4, 5, api.STATUS_ROUNDOFF_LIMITED)))
    assert np.all(np.isfinite(many["coeffs"])) and np.all(many["times"] >= 0.01)
    # a strided subset that straddles the shard boundary, against the oracle
    idx = sorted(set(range(0, 8192, 32)) | set(range(4090, 4102)))
    sub = batch.select(idx)
    ref = po.solve_batch(sub.seg_offsets, sub.waypoints, sub.fixed_mask, sub.fixed_values, sub.limits,
                         np.zeros(sub.n_segments), deriv=4, time_alloc_method=2, estimate_times=True, sampling_dt=0.2,
                         sample_capacity=CAP, n_threads=8)
    good = 0
    for k, p in enumerate(idx):
        a, b = sub.seg_offsets[k], sub.seg_offsets[k + 1]
        t = many["times"][so[p]:so[p + 1]]
        c = many["coeffs"][so[p]:so[p + 1]]
        same = (util.status_matches(many["status"][p], ref["status"][k]) and many["n_samples"][p] == ref["n_samples"][k]
                and np.max(np.abs(t - ref["times"][a:b]) / ref["times"][a:b]) < 1e-6
                and util.coeff_error(c, ref["coeffs"][a:b]) < 1e-6)
        if same:
            n = min(int(ref["n_samples"][k]), CAP)
            same = np.max(np.abs(many["samples"][p, :n, :3] - ref["samples"][k, :n, :3])) < 1e-5
        good += bool(same)
    print("RATE device list [0, 0], 8192 x 10 Mellinger: %d / %d paths agree with the oracle" % (good, len(idx)))
    assert good >= len(idx) - 1, (good, len(idx))   # measured: every path of the subset
    # and with the single-device solve: the same outer-loop kernel per path, but the closing linear solve of 8192 paths in one
    # launch is the four-lanes-per-path kernel and that of a 4096-path shard the rows kernel -- agreement to rounding, not bit for bit
    one = gpu_ctx.solve_batch(batch, None, **opts)
    rel = np.abs(one["times"] - many["times"]) / one["times"]
    assert np.mean(rel < 1e-9) >= 0.999 and np.max(rel) < 1e-3   # (a path whose feasibility scaling takes one more pass: 1e-4)
    so9 = [(a, b) for a, b in zip(so[:-1], so[1:]) if np.max(rel[a:b]) < 1e-9]
    assert max(util.coeff_error(many["coeffs"][a:b], one["coeffs"][a:b]) for a, b in so9[::16]) < 1e-9
    assert np.mean(one["status"] == many["status"]) >= 0.999 and np.mean(one["n_samples"] == many["n_samples"]) >= 0.999
