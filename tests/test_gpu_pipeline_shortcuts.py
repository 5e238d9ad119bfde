"""Two shortcuts of the time-allocation pipelines that must not change a bit of the results:

* solve_rows_pipeline_kernel (mrs_tg_rows.hip): the closing stages -- solve, per-segment maxima, feasibility scaling, second
  solve, cost, status, sampling -- in one launch for batches of up to 1024 paths;
* segment_maxima_scaling_kernel (mrs_tg_nonlinear.hip), larger batches: a (segment, k, group) maximum whose Bernstein bound is
  below its limit is not searched for (it cannot move the scaling).

Every stage is the code of the separate launches / of the full search on the same numbers, so the results must be THE SAME
BITS as with MRS_TG_ROWS_PIPELINE=0 MRS_TG_MAXIMA_BOUNDS=0 (read once per process: that setting runs in a child process); and,
like them, agree with the oracle."""
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
CAP = 256
CASES = [("uniform10", 300, 10, 4, 2), ("ragged", 200, "ragged12", 4, 2), ("snap3", 64, 3, 4, 2), ("jerk8", 96, 8, 3, 2),
         ("acc6", 96, 6, 2, 2), ("dfo0", 64, 6, 4, 0), ("one", 1, 10, 4, 2),
         # paths of more than 12 segments (another outer-loop kernel in front) and of more than 64 (the closing stages' loops)
         ("ragged30", 400, "ragged", 4, 2), ("long100", 12, 100, 4, 2),
         # more than 1024 paths: separate launches, the maxima through segment_maxima_scaling_kernel
         ("big10", 3000, 10, 4, 2), ("bigragged", 1500, "ragged", 4, 2), ("bigmixed", 2048, "mixed30", 4, 2),
         ("bigjerk", 1300, 7, 3, 2)]


def _batch(n, n_seg, d):
    if n_seg == "ragged12":
        return pr.random_mixed_batch(n, d, seed0=77, max_segments=12)
    if n_seg == "mixed30":   # limits scaled by 0.3 .. 3 per path: tight limits leave few entries to the bounds
        return pr.random_mixed_batch(n, d, seed0=177, max_segments=30)
    return pr.random_batch(n, n_seg, seed0=500, derivative_to_optimize=d)


def _solve(ctx, name):
    _, n, n_seg, d, mode = next(c for c in CASES if c[0] == name)
    batch = _batch(n, n_seg, d)
    out = ctx.solve_batch(batch, None, time_alloc_method=mode, sampling_dt=0.2, sample_capacity=CAP)
    return batch, out


CHILD = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
from mrs_uav_trajectory_generation_amd import api
from tests.test_gpu_pipeline_shortcuts import _solve, CASES
ctx = api.Context(0)
res = {}
for c in CASES:
    _, out = _solve(ctx, c[0])
    for k in ("times", "coeffs", "status", "cost", "n_samples", "samples"):
        res[c[0] + "/" + k] = out[k]
np.savez(sys.argv[1], **res)
"""


@pytest.fixture(scope="module")
def separate_launches(tmp_path_factory):
    path = str(tmp_path_factory.mktemp("pipeline_shortcuts") / "separate.npz")
    env = dict(os.environ, MRS_TG_ROWS_PIPELINE="0", MRS_TG_MAXIMA_BOUNDS="0")
    subprocess.run([sys.executable, "-c", CHILD % ROOT, path], check=True, env=env, cwd=ROOT, timeout=600)
    return np.load(path)


@pytest.mark.parametrize("name", [c[0] for c in CASES])
def test_shortcuts_give_the_bits_of_the_separate_launches_with_every_maximum_searched(gpu_ctx, separate_launches, name):
    if os.environ.get("MRS_TG_ROWS_PIPELINE", "1") == "0" or os.environ.get("MRS_TG_MAXIMA_BOUNDS", "1") == "0":
        pytest.skip("a shortcut is switched off in this process")
    batch, out = _solve(gpu_ctx, name)
    ref = separate_launches
    assert np.array_equal(out["status"], ref[name + "/status"])
    assert np.array_equal(out["times"], ref[name + "/times"])
    assert np.array_equal(out["coeffs"], ref[name + "/coeffs"])
    assert np.array_equal(out["cost"], ref[name + "/cost"], equal_nan=True)
    assert np.array_equal(out["n_samples"], ref[name + "/n_samples"])
    for p in range(batch.n_paths):
        n = min(int(out["n_samples"][p]), CAP)
        assert np.array_equal(out["samples"][p, :n], ref[name + "/samples"][p, :n]), p


def test_one_launch_against_the_oracle(gpu_ctx):
    batch, out = _solve(gpu_ctx, "uniform10")
    sub = batch.select(range(0, 300, 5))
    ref = po.solve_batch(sub.seg_offsets, sub.waypoints, sub.fixed_mask, sub.fixed_values, sub.limits, np.zeros(sub.n_segments),
                         deriv=4, time_alloc_method=2, estimate_times=True, sampling_dt=0.2, sample_capacity=CAP, n_threads=8)
    so, good = batch.seg_offsets, 0
    for k, p in enumerate(range(0, 300, 5)):
        a, b = sub.seg_offsets[k], sub.seg_offsets[k + 1]
        t, c = out["times"][so[p]:so[p + 1]], out["coeffs"][so[p]:so[p + 1]]
        good += bool(util.status_matches(out["status"][p], ref["status"][k]) and out["n_samples"][p] == ref["n_samples"][k]
                     and np.max(np.abs(t - ref["times"][a:b]) / ref["times"][a:b]) < 1e-6
                     and util.coeff_error(c, ref["coeffs"][a:b]) < 1e-6)
    print("RATE rows pipeline kernel, 300 x 10 Mellinger: %d / 60 paths agree with the oracle" % good)
    assert good >= 59
