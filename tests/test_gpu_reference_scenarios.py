"""The reference's own test scenarios, run through the C ABI.

The reference holds six rostests (test/{get_path_before_takeoff, get_path_after_takeoff, path_before_takeoff,
service_fly_now, topic_fly_now, fallback_sampling}/test.cpp) that all send the same four-waypoint path
(-5,-5,5,1) (-5,5,5,2) (5,-5,5,3) (5,5,5,4) with use_heading and assert, through
test/include/get_path_test.h:27-87, that the returned TrajectoryReference visits every waypoint in order within
POS_TOLERANCE 0.5 m and HDG_TOLERANCE 0.2 rad, and (checkWaypintIdxs :89-106) that every waypoint gets a non-zero
trajectory index; trajectory_generation_test.h does the same while flying.  They need a simulated UAV; the checks
themselves only need the trajectory, so they are restated here over mrs_tg_optimize_paths (the nodelet's optimize()).
"""
import ctypes as C

import numpy as np
import pytest

from mrs_uav_trajectory_generation_amd import api, problem as pr

pytestmark = pytest.mark.gpu

POS_TOLERANCE = 0.5   # get_path_test.h:10
HDG_TOLERANCE = 0.2   # get_path_test.h:11
TEST_PATH = np.array([[-5.0, -5.0, 5.0, 1.0], [-5.0, 5.0, 5.0, 2.0], [5.0, -5.0, 5.0, 3.0], [5.0, 5.0, 5.0, 4.0]])


def _sradians_diff(a, b):
    d = (a - b + np.pi) % (2.0 * np.pi) - np.pi
    return d


def check_trajectory(samples, path, use_heading=True):
    """GetPathTest::checkTrajectory (get_path_test.h:27-87), starting_from_current_pos = false"""
    waypoint_idx = 0
    for s in samples:
        if waypoint_idx == len(path):
            break
        w = path[waypoint_idx]
        points_dist = np.linalg.norm(s[:3] - w[:3])
        hdg_dist = abs(_sradians_diff(w[3], s[3])) if use_heading else 0.0
        if points_dist < POS_TOLERANCE and hdg_dist < HDG_TOLERANCE:
            waypoint_idx += 1
    return waypoint_idx == len(path)


def check_waypoint_idxs(idxs, path):
    """GetPathTest::checkWaypintIdxs (get_path_test.h:89-106)"""
    return len(idxs) == len(path) and all(i != 0 for i in idxs)


def _waypoint_idxs(ctx, samples, path):
    return api.waypoint_trajectory_idxs(samples, path).tolist()


# the UAV hovers at the take-off spot when the service is called; the nodelet prepends that state (:2095-2110)
CURRENT_STATE = np.array([0.0, 0.0, 3.0, 0.5])


@pytest.mark.parametrize("time_alloc,deriv", [(2, 2), (2, 4), (0, 2), (3, 2)])
def test_get_path_visits_every_waypoint(gpu_ctx, time_alloc, deriv):
    """get_path_{before,after}_takeoff/test.cpp: the shipping configuration (Mellinger, minimum acceleration) and the
    other time-allocation modes"""
    path = np.vstack([CURRENT_STATE, TEST_PATH])
    pol = api.default_policy_options(solver=dict(derivative_to_optimize=deriv, time_alloc_method=time_alloc))
    init = dict(heading=CURRENT_STATE[3], velocity=[0, 0, 0, 0], acceleration=[0, 0, 0, 0], jerk=[0, 0, 0, 0])
    out = api.optimize_paths(gpu_ctx, [path], initial_states=[init], policy=pol, sample_capacity=4096)
    assert out["success"][0] == 1
    n = int(out["n_samples"][0])
    samples = out["samples"][0, :n]
    assert np.linalg.norm(samples[0, :3] - CURRENT_STATE[:3]) < POS_TOLERANCE     # "initial condition" check :31-44
    assert abs(_sradians_diff(samples[0, 3], CURRENT_STATE[3])) < HDG_TOLERANCE
    assert check_trajectory(samples, TEST_PATH)
    idxs = _waypoint_idxs(gpu_ctx, samples, path)
    assert check_waypoint_idxs(idxs[1:], TEST_PATH)       # indices of the four requested waypoints
    assert all(a < b for a, b in zip(idxs[:-1], idxs[1:]))
    if time_alloc == 2:
        # the deviation loop of the shipping configuration (check_trajectory_deviation: 0.05 m, up to 6 re-solves) keeps
        # the result near the polyline; the gradient-free modes have no feasibility scaling and are returned as they are
        assert out["max_deviation"][0] < POS_TOLERANCE


def test_fallback_sampling_visits_every_waypoint(gpu_ctx):
    """fallback_sampling/test.cpp: the constant-velocity fallback sampler must satisfy the same check"""
    path = np.vstack([CURRENT_STATE, TEST_PATH])
    pol = api.default_policy_options(fallback_sampling=1)
    out = api.optimize_paths(gpu_ctx, [path], policy=pol, sample_capacity=4096)
    assert out["success"][0] == 1
    samples = out["samples"][0, :int(out["n_samples"][0])]
    assert check_trajectory(samples, TEST_PATH)
    assert check_waypoint_idxs(_waypoint_idxs(gpu_ctx, samples, path)[1:], TEST_PATH)


def test_stop_at_waypoints_and_loop_like_paths(gpu_ctx):
    """service_fly_now / topic_fly_now send the same path with stop_at_waypoints toggled by the launch files"""
    path = np.vstack([CURRENT_STATE, TEST_PATH])
    stops = [[False, True, True, True, True]]
    out = api.optimize_paths(gpu_ctx, [path], stop_flags=stops, sample_capacity=4096)
    assert out["success"][0] == 1
    samples = out["samples"][0, :int(out["n_samples"][0])]
    assert check_trajectory(samples, TEST_PATH)
    # the trajectory comes to rest at every stop waypoint: successive samples around the closest approach coincide
    for w in TEST_PATH[:-1]:
        i = int(np.argmin(np.linalg.norm(samples[:, :3] - w[:3], axis=1)))
        i = min(max(i, 1), len(samples) - 2)
        v = np.linalg.norm(samples[i + 1, :3] - samples[i - 1, :3]) / 0.4   # dt = 0.2
        assert v < 0.6, (w, v)


def test_batch_of_reference_paths_is_the_single_path_repeated(gpu_ctx):
    """one request per call in the reference; here 64 requests in one call must each equal the single-request answer"""
    path = np.vstack([CURRENT_STATE, TEST_PATH])
    one = api.optimize_paths(gpu_ctx, [path], sample_capacity=2048)
    many = api.optimize_paths(gpu_ctx, [path] * 64, sample_capacity=2048)
    n = int(one["n_samples"][0])
    assert np.all(many["success"] == 1) and np.all(many["n_samples"] == n)
    for p in range(64):
        assert np.array_equal(many["samples"][p, :n], one["samples"][0, :n])
