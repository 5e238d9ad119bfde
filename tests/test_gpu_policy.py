"""GPU parity of the path-policy layer (mrs_tg_optimize_paths) against the oracle's optimize() restatement:
preprocessing, solve, length sanity check, spatial validation, mid-point subdivision rounds, fallback sampler,
override_heading_atan2.  Tolerances: sample positions 1e-6 m (the optimiser sits in the loop), identical
success / waypoint counts / iteration counts for >= 90 % of the paths (a sample that lands within 1e-9 of the
0.05 m deviation threshold may flip a subdivision decision between the two arithmetic routes)."""
import numpy as np
import pytest

from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _paths(n, seed0, gen):
    return [gen(4 + (i % 5), seed0 + i) for i in range(n)]


@pytest.mark.parametrize("deriv", [2, 4])
def test_policy_loop_matches_oracle(gpu_ctx, deriv):
    paths = _paths(24, 500, pr.random_walk_waypoints)
    pol = api.default_policy_options(solver=dict(derivative_to_optimize=deriv))
    out = api.optimize_paths(gpu_ctx, paths, policy=pol, sample_capacity=2048)
    same = 0
    for p, wp in enumerate(paths):
        ref = po.optimize_path(wp, limits=pr.DEFAULT_LIMITS, deriv=deriv, capacity=2048)
        assert out["success"][p] == ref["success"]
        if (out["n_waypoints"][p] == ref["n_waypoints"] and out["iterations"][p] == ref["iterations"]
                and out["n_samples"][p] == ref["n_samples"]):
            n = ref["n_samples"]
            if n == 0 or np.max(np.abs(out["samples"][p, :n, :3] - ref["samples"][:, :3])) < 1e-6:
                same += 1
                assert abs(out["max_deviation"][p] - ref["max_deviation"]) < 1e-6
    print("RATE policy deriv %d: %d / %d" % (deriv, same, len(paths)))
    assert same >= len(paths) - 1, same   # measured 24 / 24 for both objectives


def test_policy_with_initial_state_stop_at_and_relaxed_heading(gpu_ctx):
    paths = _paths(8, 900, pr.random_box_waypoints)
    stops = [[(i % 3 == 1) for i in range(len(p))] for p in paths]
    inits = [dict(heading=p[0, 3], velocity=[0.2, 0.1, 0.0, 0.0], acceleration=[0.0, 0.0, 0.0, 0.0], jerk=[0.0, 0.0, 0.0, 0.0])
             if k % 2 == 0 else None for k, p in enumerate(paths)]
    relax = np.array([k % 3 == 0 for k in range(len(paths))], dtype=np.uint8)
    out = api.optimize_paths(gpu_ctx, paths, stop_flags=stops, initial_states=inits, relax_heading=relax, sample_capacity=4096)
    ok = 0
    for p, wp in enumerate(paths):
        ref = po.optimize_path(wp, stop_at=stops[p], initial_state=inits[p], limits=pr.DEFAULT_LIMITS,
                               relax_heading=bool(relax[p]), capacity=4096)
        assert out["success"][p] == ref["success"]
        if out["n_samples"][p] == ref["n_samples"] and out["n_waypoints"][p] == ref["n_waypoints"]:
            n = ref["n_samples"]
            if n == 0 or np.max(np.abs(out["samples"][p, :n, :3] - ref["samples"][:, :3])) < 1e-6:
                ok += 1
    assert ok >= len(paths) - 1


def test_fallback_sampler_and_atan2_heading_match_oracle(gpu_ctx):
    paths = _paths(10, 40, pr.random_box_waypoints)
    stops = [[(i == 2) for i in range(len(p))] for p in paths]
    pol = api.default_policy_options(fallback_sampling=1, override_heading_atan2=1)
    out = api.optimize_paths(gpu_ctx, paths, stop_flags=stops, policy=pol, sample_capacity=4096)
    for p, wp in enumerate(paths):
        ref = po.optimize_path(wp, stop_at=stops[p], limits=pr.DEFAULT_LIMITS,
                               policy=po.default_policy(fallback_sampling=1, override_heading_atan2=1), capacity=4096)
        assert out["success"][p] == ref["success"] == 1
        n = ref["n_samples"]
        assert out["n_samples"][p] == n
        assert np.max(np.abs(out["samples"][p, :n, :3] - ref["samples"][:, :3])) < 1e-12
        dh = np.abs(out["samples"][p, :n, 3] - ref["samples"][:, 3])
        assert np.max(np.minimum(dh, 2 * np.pi - dh)) < 1e-9


def test_degenerate_and_overflowing_paths_fail_cleanly(gpu_ctx):
    paths = [np.array([[0.0, 0.0, 1.0, 0.0]]),                       # single waypoint
             np.array([[0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 1.0, 0.0]]),   # two identical waypoints
             pr.random_walk_waypoints(5, 1)]
    out = api.optimize_paths(gpu_ctx, paths, sample_capacity=8)      # far too small a capacity for the third
    assert out["success"][0] == 0 and out["success"][2] == 0
    assert np.all(out["n_samples"][out["success"] == 0] == 0)


def test_response_arrays_kept_between_calls_hold_the_second_call_s_answers(gpu_ctx):
    """api.optimize_paths(out=): a server's loop writes batch after batch into the same response arrays; a request that fails
    in the second batch must not show the first batch's samples"""
    first = [pr.random_box_waypoints(4 + (i % 5), 300 + i) for i in range(70)]
    second = [pr.random_walk_waypoints(4 + (i % 6), 900 + i) for i in range(70)]
    second[3] = np.array([[0.0, 0.0, 1.0, 0.0]])    # a single waypoint: no trajectory
    kept = api.optimize_paths(gpu_ctx, first, sample_capacity=1024)
    assert kept["success"].sum() >= 69
    again = api.optimize_paths(gpu_ctx, second, sample_capacity=1024, out=kept)
    fresh = api.optimize_paths(gpu_ctx, second, sample_capacity=1024)
    assert again["samples"] is kept["samples"]
    for k in ("success", "n_samples", "max_deviation", "n_waypoints", "iterations"):
        assert np.array_equal(again[k], fresh[k]), k
    assert again["success"][3] == 0 and again["n_samples"][3] == 0
    for p in range(70):
        n = int(fresh["n_samples"][p])
        assert np.array_equal(again["samples"][p, :n], fresh["samples"][p, :n]), p


def test_waypoint_trajectory_idxs(gpu_ctx):
    import ctypes as C
    wp = pr.random_walk_waypoints(5, 2)
    out = api.optimize_paths(gpu_ctx, [wp], sample_capacity=2048)
    n = int(out["n_samples"][0])
    smp = np.ascontiguousarray(out["samples"][0, :n])
    idx = api.waypoint_trajectory_idxs(smp, wp)
    k = len(idx)
    ref = np.zeros(16, dtype=np.int32)
    kr = po.lib().mto_waypoint_trajectory_idxs(po._dp(smp), n, po._dp(np.ascontiguousarray(wp)), wp.shape[0],
                                               ref.ctypes.data_as(C.POINTER(C.c_int32)))
    assert k == kr and np.array_equal(idx[:k], ref[:kr]) and k >= wp.shape[0] - 1


# ---- batches of requests: the policy's per-path host work runs on several threads (mrs_tg_policy.hip::parallel_ranges, from a
# few hundred requests on) and its arrays live in pinned scratch memory of the context.  The requests are independent, so the
# results must be THE SAME BITS as on one thread with the rounds' arrays in ordinary memory (MRS_TG_POLICY_THREADS=1
# MRS_TG_POLICY_PINNED=0 -- the route taken when the runtime refuses the pinned block --, read once per process: a child process), and a
# strided subset must agree with the oracle as the small batches above do.
POLICY_CHILD = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
from mrs_uav_trajectory_generation_amd import api
from tests.test_gpu_policy import _many_requests
ctx = api.Context(0)
out = _many_requests(ctx)
np.savez(sys.argv[1], **out)
"""


def _many_requests(ctx):
    paths = [(pr.random_box_waypoints if i % 3 else pr.random_walk_waypoints)(4 + (i % 6), 3100 + i) for i in range(700)]
    stops = [[(i % 4 == 2 and 0 < k < len(p) - 1) for k in range(len(p))] for i, p in enumerate(paths)]
    return api.optimize_paths(ctx, paths, stop_flags=stops, sample_capacity=1024)


def test_batch_of_requests_on_threads_gives_the_bits_of_one_thread_and_agrees_with_the_oracle(gpu_ctx, tmp_path):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = _many_requests(gpu_ctx)
    again = _many_requests(gpu_ctx)            # the context's scratch block is reused: nothing of the first call may linger
    for k in out:
        assert np.array_equal(out[k], again[k]), k
    ref_path = str(tmp_path / "one_thread.npz")
    subprocess.run([sys.executable, "-c", POLICY_CHILD % root, ref_path], check=True, cwd=root, timeout=600,
                   env=dict(os.environ, MRS_TG_POLICY_THREADS="1", MRS_TG_POLICY_PINNED="0"))   # (and ordinary memory for the rounds' arrays)
    ref = np.load(ref_path)
    for k in out:
        if k != "samples":
            assert np.array_equal(out[k], ref[k]), k
    for p in range(out["success"].size):   # (rows beyond a request's count are not part of the result)
        n = int(out["n_samples"][p])
        assert np.array_equal(out["samples"][p, :n], ref["samples"][p, :n]), p
    assert out["success"].mean() > 0.98 and out["iterations"].max() >= 3      # (the subdivision rounds did run)
    paths = [(pr.random_box_waypoints if i % 3 else pr.random_walk_waypoints)(4 + (i % 6), 3100 + i) for i in range(700)]
    same = n_checked = 0
    for p in range(0, 700, 35):
        stops = [(p % 4 == 2 and 0 < k < len(paths[p]) - 1) for k in range(len(paths[p]))]
        o = po.optimize_path(paths[p], stop_at=stops, limits=pr.DEFAULT_LIMITS, capacity=1024)
        n_checked += 1
        assert out["success"][p] == o["success"]
        if (out["n_waypoints"][p] == o["n_waypoints"] and out["iterations"][p] == o["iterations"] and out["n_samples"][p] == o["n_samples"]):
            n = o["n_samples"]
            if n == 0 or np.max(np.abs(out["samples"][p, :n, :3] - o["samples"][:, :3])) < 1e-6:
                same += 1
    print("RATE policy batch of 700: %d / %d" % (same, n_checked))
    assert same >= n_checked - 2, (same, n_checked)


def test_device_rounds_give_the_bits_of_the_host_rounds(gpu_ctx, tmp_path):
    """Round 6: the batch-sized work of a policy round -- vertex expansion, the two gates of findTrajectory, the scan of
    validateTrajectorySpatial (/root/reference/src/mrs_trajectory_generation.cpp:923-977, 1138-1149, 1178-1199, 1401-1455) -- runs on
    the device (mrs_tg_policy_dev.hip) from 64 active requests on; MRS_TG_POLICY_DEVICE=0 (read once per process: a child process)
    keeps it on the policy's host threads as until round 5, MRS_TG_POLICY_DEVICE=1 sends single requests there too.  Same decisions, same deviations, same samples: bit for bit on every request (the arithmetic
    of the scan is the host's, operation by operation).  Rows of the caller's sample array beyond a request's n_samples are not
    compared: the host route leaves earlier rounds' longer trajectories there, the device route brings down the final one only."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = _many_requests(gpu_ctx)
    ref_path = str(tmp_path / "host_rounds.npz")
    subprocess.run([sys.executable, "-c", POLICY_CHILD % root, ref_path], check=True, cwd=root, timeout=600,
                   env=dict(os.environ, MRS_TG_POLICY_DEVICE="0"))
    ref = np.load(ref_path)
    for k in ("success", "n_samples", "n_waypoints", "iterations", "max_deviation"):
        assert np.array_equal(out[k], ref[k]), k
    for p in range(out["success"].size):
        n = int(out["n_samples"][p])
        assert np.array_equal(out["samples"][p, :n], ref["samples"][p, :n]), p
    assert out["success"].mean() > 0.98 and out["iterations"].max() >= 3


DEVICE_ROUTE_CHILD = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
from mrs_uav_trajectory_generation_amd import api
from tests.test_gpu_policy import _mixed_requests
ctx = api.Context(0)
np.savez(sys.argv[1], **_mixed_requests(ctx))
"""


def _mixed_requests(ctx):
    """40 requests with everything a vertex can carry: stop_at waypoints, moving initial states, relaxed headings, per-request limits,
    and under min-snap as well as the default min-acceleration -- two calls, results concatenated"""
    paths = _paths(40, 900, pr.random_box_waypoints)
    stops = [[(i % 3 == 1) for i in range(len(p))] for p in paths]
    inits = [dict(heading=p[0, 3] + 0.4, velocity=[0.2, 0.1, -0.1, 0.05], acceleration=[0.1, 0.0, 0.05, 0.0], jerk=[0.0, 0.02, 0.0, 0.0])
             if k % 2 == 0 else None for k, p in enumerate(paths)]
    relax = np.array([k % 3 == 0 for k in range(len(paths))], dtype=np.uint8)
    lim = np.stack([pr.DEFAULT_LIMITS * (0.7 + 0.1 * (k % 5)) for k in range(len(paths))])
    out = {}
    for d in (2, 4):
        pol = api.default_policy_options(solver=dict(derivative_to_optimize=d))
        r = api.optimize_paths(ctx, paths, limits=lim, stop_flags=stops, initial_states=inits, relax_heading=relax, policy=pol,
                               sample_capacity=2048)
        for k, v in r.items():
            out["%s_d%d" % (k, d)] = v
    return out


def test_device_route_on_small_batches_with_initial_states_stop_at_and_relaxed_headings(gpu_ctx, tmp_path):
    """MRS_TG_POLICY_DEVICE=1 sends every round to the device route, whatever its size (the default keeps fewer than 64 active
    requests on the host route): policy_expand_kernel with moving initial states, stop_at flags, relaxed heading limits and
    per-request limits, d = 2 and 4 -- the same bits as the host route of this process, and the oracle's decisions."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = _mixed_requests(gpu_ctx)          # 40 requests: the host route
    ref_path = str(tmp_path / "device_route.npz")
    subprocess.run([sys.executable, "-c", DEVICE_ROUTE_CHILD % root, ref_path], check=True, cwd=root, timeout=600,
                   env=dict(os.environ, MRS_TG_POLICY_DEVICE="1"))
    dev = np.load(ref_path)
    for d in (2, 4):
        for k in ("success", "n_samples", "n_waypoints", "iterations", "max_deviation"):
            assert np.array_equal(out["%s_d%d" % (k, d)], dev["%s_d%d" % (k, d)]), (k, d)
        for p in range(40):
            n = int(out["n_samples_d%d" % d][p])
            assert np.array_equal(out["samples_d%d" % d][p, :n], dev["samples_d%d" % d][p, :n]), (p, d)
        assert out["success_d%d" % d].sum() >= 36
