"""The C++ host side of the drop-in: examples/path_service_host.cpp (include/mrs_tg_service.hpp, the nodelet's
service callbacks without ROS) is built with plain g++ against libmrs_tg.so -- no HIP, no torch in the host -- run on
the GPU, and its TrajectoryReference output is checked the way the reference's rostests check the nodelet's
(test/include/get_path_test.h:27-106) together with the service-level error strings of callbackPathSrv
(src/mrs_trajectory_generation.cpp:1976-2041, 2113)."""
import json
import os
import subprocess

import numpy as np
import pytest

from tests.test_gpu_reference_scenarios import TEST_PATH, check_trajectory, check_waypoint_idxs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_output(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("host") / "path_service_host")
    libdir = os.path.join(ROOT, "mrs_uav_trajectory_generation_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "path_service_host.cpp"), "-o", exe, "-L", libdir, "-lmrs_tg",
                           "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=300).stdout
    return json.loads(out)


def test_error_strings_of_the_service(host_output):
    assert host_output["missing_constraints"] == dict(host_output["missing_constraints"], success=False, message="missing constraints")
    assert (host_output["empty"]["success"], host_output["empty"]["message"]) == (False, "received an empty message")
    assert (host_output["nan"]["success"], host_output["nan"]["message"]) == (False, "invalid path")
    for k in ("missing_constraints", "empty", "nan"):
        assert host_output[k]["points"] == []


@pytest.mark.parametrize("case", ["plain", "override", "no_state", "fallback"])
def test_trajectory_reference_passes_the_reference_checks(host_output, case):
    r = host_output[case]
    assert r["success"] and r["message"] == "trajectory generated"
    assert r["dt"] == 0.2 and r["use_heading"] and r["frame_id"] == "uav1/world_origin"
    pts = np.array(r["points"])
    assert check_trajectory(pts, TEST_PATH)
    if case not in ("no_state", "fallback"):    # those two run after clearCurrentState()
        # the trajectory starts at the current state, every requested waypoint has a non-zero index (get_path_test.h:89-106)
        assert np.linalg.norm(pts[0, :3] - [0.0, 0.0, 3.0]) < 0.5
        assert check_waypoint_idxs(r["idxs"], TEST_PATH)
        assert r["fly_now"]
    else:
        assert r["idxs"][0] == 0 and len(r["idxs"]) == 4 and not r["fly_now"]   # no initial condition: "fly now" dropped
    assert all(a < b for a, b in zip(r["idxs"][:-1], r["idxs"][1:]))


def test_batch_members_keep_their_own_request_fields(host_output):
    assert host_output["plain"]["input_id"] == 7 and host_output["loop_stop"]["input_id"] == 8 and host_output["override"]["input_id"] == 9
    loop = host_output["loop_stop"]
    assert loop["success"] and loop["loop"] and len(loop["idxs"]) == 5       # the first waypoint is appended once more
    pts = np.array(loop["points"])
    assert check_trajectory(pts, np.vstack([TEST_PATH, TEST_PATH[:1]]))
    # user limits of 4 m/s instead of 2 m/s: a shorter trajectory; and its own, looser deviation bound was honoured
    assert len(host_output["override"]["points"]) < len(host_output["plain"]["points"])
    assert host_output["override"]["max_deviation"] <= 0.2 + 1e-9 or host_output["override"]["success"]
    v = np.linalg.norm(np.diff(np.array(host_output["override"]["points"])[:, :2], axis=0), axis=1) / 0.2
    assert 2.2 < v.max() < 4.0 * 1.05


def test_a_late_request_gets_the_fallback_trajectory(host_output):
    """optimize() with overtime() true at its start (src/mrs_trajectory_generation.cpp:711-713) runs the fallback sampler:
    the request succeeds, and the trajectory is the one `fallback_sampling` alone produces"""
    r = host_output["late_request"]
    assert r["success"] and r["message"] == "trajectory generated"
    assert host_output["late_equals_fallback"] is True
    assert check_trajectory(np.array(r["points"]), TEST_PATH)


def test_many_requests_with_the_default_deadline_are_all_answered(host_output):
    """ServiceParams' default max_time (0.5 s, config/public/trajectory_generation.yaml:4) on a call of 600 requests in three
    deviation groups: a request's clock starts with its own group's GPU call, and running late means fallback sampling,
    not failure -- nobody is dropped"""
    m = host_output["many"]
    assert m["requests"] == 600 and m["success"] == 600 and m["own_fields"] == 600


def test_plain_c99_host(tmp_path):
    """examples/solve_batch_host.c: the header is C (not C++), the library needs nothing but itself at link time"""
    exe = str(tmp_path / "solve_batch_host")
    libdir = os.path.join(ROOT, "mrs_uav_trajectory_generation_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "solve_batch_host.c"), "-o", exe, "-L", libdir, "-lmrs_tg",
                           "-Wl,-rpath," + libdir])
    r = json.loads(subprocess.run([exe], check=True, capture_output=True, text=True, timeout=300).stdout)
    assert r["abi"] == 5 and len(r["paths"]) == 2
    # the device-list call (two contexts on this box's one GPU): a ragged batch is balanced on the segment count, and the
    # results are bit-identical to the single-device call
    assert r["multi"] == {"devices": 2, "shard": [0, 1], "identical": True}
    for p, first, last in zip(r["paths"], ([-5, -5, 5], [0, 0, 2]), ([5, 5, 5], [20, 0, 2])):
        assert p["status"] in (1, 3, 4, 5) and p["n_samples"] > 10 and p["cost"] > 0
        assert np.allclose(p["first"], first, atol=1e-9)
        assert np.linalg.norm(np.array(p["last"]) - last) < 0.5       # the last sample precedes the end by < dt
        assert all(t >= 0.01 for t in p["times"])
    # the straight line: symmetric segment times
    t = r["paths"][1]["times"]
    assert abs(t[0] - t[1]) < 1e-6 * t[0]


def test_cpp_stream_server_keeps_batches_in_flight(tmp_path):
    """examples/stream_server_host.cpp: a g++-built host with four contexts / streams, bound solves issued by
    mrs_tg_bound_solve_launch_many under MRS_TG_FLAG_SHARED_DEVICE; every lane ends with the one-call interface's result
    bit for bit (the program's exit code says so) and the JSON line carries the throughput."""
    exe = str(tmp_path / "stream_server_host")
    libdir = os.path.join(ROOT, "mrs_uav_trajectory_generation_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                           "-I", "/opt/rocm/include", os.path.join(ROOT, "examples", "stream_server_host.cpp"), "-o", exe,
                           "-L", libdir, "-lmrs_tg", "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, "1024", "4", "400"], check=True, capture_output=True, text=True, timeout=300).stdout
    r = json.loads(out.strip().splitlines()[-1])
    assert r["paths"] == 1024 and r["in_flight"] == 4 and r["steps"] == 400
    assert r["max_abs_diff_vs_one_call_interface"] == 0.0
    assert r["trajectories_per_s"] > 2.0e7   # (hundreds of millions on an idle MI355X; a loose floor for a shared box)


def test_cpp_request_latency_host(tmp_path):
    """examples/request_latency_host.cpp: one request through mrs_tg_find_trajectory from a g++-built host, repeated; the
    request succeeds, yields samples, and a call is a fraction of a millisecond (60-100 us on an idle MI355X)."""
    exe = str(tmp_path / "request_latency_host")
    libdir = os.path.join(ROOT, "mrs_uav_trajectory_generation_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "request_latency_host.cpp"), "-o", exe, "-L", libdir, "-lmrs_tg",
                           "-Wl,-rpath," + libdir])
    out = subprocess.run([exe, "11", "100"], check=True, capture_output=True, text=True, timeout=300).stdout
    import re
    m = re.search(r"10 segments \(status (-?\d+), (\d+) samples, ([\d.]+) s of trajectory\): median ([\d.]+) us", out)
    assert m, out
    assert int(m.group(1)) >= 1 and int(m.group(2)) > 10 and float(m.group(3)) > 1.0
    assert float(m.group(4)) < 1000.0, out


def test_cpp_wrapper_returns_nullopt_where_the_reference_returns_nothing(tmp_path):
    """examples/find_trajectory_host.cpp over include/mrs_tg.hpp (the wrapper INTEGRATION.md section 2 puts into the nodelet): the
    reference tests' path is found; path 2843 of the box generator -- an accepted optimiser code, 3.12 x its Baca estimate long --
    comes back as nullopt with the reference's message, and is found once the check's upper side is switched off; the same
    decisions as oracle/mto_policy.c::mto_find_trajectory."""
    from mrs_uav_trajectory_generation_amd import problem as pr
    from oracle import pyoracle as po
    exe = str(tmp_path / "find_trajectory_host")
    libdir = os.path.join(ROOT, "mrs_uav_trajectory_generation_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "find_trajectory_host.cpp"), "-o", exe, "-L", libdir, "-lmrs_tg",
                           "-Wl,-rpath," + libdir])

    def run(wp, *factors):
        path = str(tmp_path / "wp.txt")
        np.savetxt(path, wp, fmt="%.17g")
        out = subprocess.run([exe, path] + [str(f) for f in factors], check=True, capture_output=True, text=True, timeout=300).stdout
        return json.loads(out.strip().splitlines()[-1])

    good = run(pr.CONFIG1_WAYPOINTS)
    ref = po.find_trajectory(pr.CONFIG1_WAYPOINTS, limits=pr.DEFAULT_LIMITS, deriv=4)
    assert good["found"] == 1 and good["samples"] == ref["n_samples"] and good["rejection"] == 0
    assert abs(good["baca"] - ref["baca_total_time"]) <= 1e-12 * ref["baca_total_time"]
    wp = pr.random_box_waypoints(10, 2843)
    long_ = run(wp)
    ref = po.find_trajectory(wp, limits=pr.DEFAULT_LIMITS, deriv=4)
    assert ref["success"] == 0 and ref["rejection"] == 2
    assert long_["found"] == 0 and long_["rejection"] == 2 and long_["status"] >= 1 and "too long" in long_["message"]
    free = run(wp, 0.0)
    assert free["found"] == 1 and free["samples"] == ref["raw_n_samples"]
