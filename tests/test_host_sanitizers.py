"""Sanitizers on the CPU build (round-5 VERDICT item 6; GPU AddressSanitizer is not available on the pool):

* the C oracle -- every numeric result the parity tests rest on -- under AddressSanitizer + UndefinedBehaviorSanitizer and under
  ThreadSanitizer (tests/host/oracle_san_harness.c: every time-allocation mode, the three arithmetic routes, the pthread pool
  with its per-thread scratch stacks; one thread = the pool = the pool again, bit for bit), its checksum equal to the
  unsanitised build's;
* the PRODUCT's pure-host code -- csrc/mrs_tg_policy_host.hpp (the policy layer on 16 threads) and include/mrs_tg_service.hpp --
  compiled with g++ under the same sanitizers with the oracle as the solver (tests/host/policy_host_harness.cpp), every request
  compared bit for bit with oracle/mto_policy.c::mto_optimize_path
  (/root/reference/src/mrs_trajectory_generation.cpp:431-500, 729-785, 1215-1455).

`make -C oracle SAN=<...> san` builds both programs into oracle/_san/<SAN>/."""
import os
import struct
import subprocess

import numpy as np
import pytest

from mrs_uav_trajectory_generation_amd import problem as pr

# sized for the CPU tier: the thread-sanitised policy run is the longest leg (~1 minute on 8 cores); the harness takes any number
# (280 requests through every leg: 47 s plain, 2 min 50 s under ASan + UBSan, all bit-equal to the oracle's policy loop)
N_REQUESTS, N_SERVICE = 48, 10

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")


def _build(san):
    subprocess.check_call(["make", "-C", ORACLE, "-s", "SAN=%s" % san, "san"])
    return os.path.join(ORACLE, "_san", san)


def _run(cmd, timeout):
    env = dict(os.environ)
    # (leak check on: a thread of the oracle gives its scratch blocks back when it ends, mto_scratch.c)
    env["ASAN_OPTIONS"] = "detect_leaks=1:abort_on_error=0"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    env["TSAN_OPTIONS"] = "halt_on_error=1:second_deadlock_stack=1"
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, env=env)
    return p.returncode, p.stdout, p.stderr


def _write_requests(path, n_req):
    """box-generator and random-walk requests of 2 .. 10 waypoints; every fourth with stop_at waypoints, every fifth from a
    moving initial state, every seventh with relax_heading, limits scaled per request; two degenerate ones"""
    rng = np.random.default_rng(2024)
    with open(path, "wb") as f:
        f.write(struct.pack("<i", n_req))
        for i in range(n_req):
            n_seg = 1 + i % 9
            wp = pr.random_box_waypoints(n_seg, 31000 + i) if i % 2 == 0 else pr.random_walk_waypoints(n_seg, 31000 + i)
            if i == 5:
                wp = wp[:1]                                   # a single waypoint: "the path is empty (after postprocessing)"
            if i == 9:
                wp = np.vstack([wp[:2], wp[1:2] + 1e-3, wp[2:]])   # two waypoints 1 mm apart: preprocessPath drops one
            n = wp.shape[0]
            stop = np.zeros(n, dtype=np.uint8)
            if i % 4 == 1 and n > 2:
                stop[1 + (i // 4) % (n - 2)] = 1
            has_init = 1 if (i % 5 == 2) else 0
            init = np.zeros(13)
            if has_init:
                init[0] = wp[0, 3]
                init[1:4] = rng.uniform(-1, 1, 3)
                init[4] = 0.1
                init[5:8] = rng.uniform(-0.5, 0.5, 3)
                init[9:12] = rng.uniform(-0.2, 0.2, 3)
            lim = pr.DEFAULT_LIMITS * (0.6 + 0.2 * (i % 4))
            f.write(struct.pack("<i", n))
            f.write(np.ascontiguousarray(wp, dtype="<f8").tobytes())
            f.write(stop.tobytes())
            f.write(struct.pack("<B", has_init))
            f.write(np.ascontiguousarray(init, dtype="<f8").tobytes())
            f.write(struct.pack("<B", 1 if i % 7 == 3 else 0))
            f.write(np.ascontiguousarray(lim, dtype="<f8").tobytes())


@pytest.fixture(scope="module")
def plain(tmp_path_factory):
    """the unsanitised build: its output is what the sanitised runs must reproduce"""
    d = _build("none")
    req = str(tmp_path_factory.mktemp("san") / "requests.bin")
    _write_requests(req, N_REQUESTS)
    rc, out, err = _run([os.path.join(d, "oracle_san_harness")], 600)
    assert rc == 0 and out.startswith("OK checksum"), (rc, out, err[-2000:])
    rc2, out2, err2 = _run([os.path.join(d, "policy_host_harness"), req, "16", str(N_SERVICE)], 900)
    assert rc2 == 0 and "OK %d requests" % N_REQUESTS in out2, (rc2, out2, err2[-2000:])
    return dict(requests=req, oracle_line=out.strip(), policy_line=out2.strip().splitlines()[-1])


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_oracle_under_sanitizers(plain, san):
    d = _build(san)
    rc, out, err = _run([os.path.join(d, "oracle_san_harness")], 1500)
    assert rc == 0, (rc, out, err[-3000:])
    assert "Sanitizer" not in err and "runtime error" not in err, err[-3000:]
    assert out.strip() == plain["oracle_line"]          # the same bits as the unsanitised build


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_policy_host_code_under_sanitizers_with_the_oracle_as_solver(plain, san):
    d = _build(san)
    rc, out, err = _run([os.path.join(d, "policy_host_harness"), plain["requests"], "16", str(N_SERVICE)], 2400)
    assert rc == 0, (rc, out, err[-3000:])
    assert "Sanitizer" not in err and "runtime error" not in err and "MISMATCH" not in err, err[-3000:]
    assert out.strip().splitlines()[-1] == plain["policy_line"]
    assert "policy threads 16, ranges per call 16" in out     # (MRS_TG_POLICY_GRAIN: the 40 requests really ran on 16 threads)
