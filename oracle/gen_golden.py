#!/usr/bin/env python3
"""Generate tests/golden/*.json: 60-digit ground truth of the hot path's formulas.

TEST INFRASTRUCTURE (see oracle/mrs_tg_oracle.h).  The reference ships no golden vectors for
this path and cannot be built here, so the fixtures are produced by evaluating the SAME
formulas (/root/reference/include/eth_trajectory_generation/impl/
polynomial_optimization_linear_impl.h:113-121,148-177,184-257,311-373,606-618 and
impl/polynomial_optimization_nonlinear_impl.h:257-333) in 60-digit mpmath arithmetic and
rounding once to double.  They pin both the C oracle and the HIP path.

Run from the repo root:  python3 oracle/gen_golden.py
"""
import json
import os
import sys

import mpmath as mp
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrs_uav_trajectory_generation_amd import problem as pr  # noqa: E402  (input construction only)

mp.mp.dps = 60
N, D, HALF = 10, 4, 5
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def base(r, k):
    return mp.factorial(k) / mp.factorial(k - r) if k >= r else mp.mpf(0)


def mapping(T):
    A = mp.zeros(N, N)
    for r in range(HALF):
        A[r, r] = base(r, r)
        for k in range(r, N):
            A[HALF + r, k] = base(r, k) * mp.mpf(T) ** (k - r)
    return A


def cost_matrix(d, T):
    Q = mp.zeros(N, N)
    for i in range(d, N):
        for j in range(d, N):
            e = i + j - 2 * d + 1
            Q[i, j] = 2 * base(d, i) * base(d, j) * mp.mpf(T) ** e / e
    return Q


def exact_solve(mask, vals, times, d):
    """-> coeffs [S][4][10] (mpf), cost J_d, per-segment H and A^-1"""
    S = len(times)
    V = S + 1
    n_all = HALF * V
    flat_mask = [int(x) for x in np.asarray(mask).reshape(-1)]
    col = {}
    cf, cp = 0, sum(flat_mask)
    n_fixed = cp
    for i in range(n_all):
        if flat_mask[i]:
            col[i] = cf
            cf += 1
        else:
            col[i] = cp
            cp += 1
    R = mp.zeros(n_all, n_all)
    Ainv, Hs, Qs = [], [], []
    for i in range(S):
        Ai = mp.inverse(mapping(times[i]))
        Q = cost_matrix(d, times[i])
        H = Ai.T * Q * Ai
        Ainv.append(Ai)
        Hs.append(H)
        Qs.append(Q)
        for r in range(N):
            cr = col[(i + r // HALF) * HALF + r % HALF]
            for c in range(N):
                cc = col[(i + c // HALF) * HALF + c % HALF]
                R[cr, cc] += H[r, c]
    vals = np.asarray(vals, dtype=np.float64).reshape(n_all, D)
    n_free = n_all - n_fixed
    dall = mp.zeros(n_all, D)
    for i in range(n_all):
        if flat_mask[i]:
            for k in range(D):
                dall[col[i], k] = mp.mpf(float(vals[i, k]))
    if n_free > 0:
        Rpp = R[n_fixed:, n_fixed:]
        Rpf = R[n_fixed:, :n_fixed]
        rhs = -(Rpf * dall[:n_fixed, :])
        Rpp_inv = mp.inverse(Rpp)
        sol = Rpp_inv * rhs
        for r in range(n_free):
            for k in range(D):
                dall[n_fixed + r, k] = sol[r, k]
    coeffs = [[[None] * N for _ in range(D)] for _ in range(S)]
    J = mp.mpf(0)
    for i in range(S):
        for k in range(D):
            dseg = mp.matrix([dall[col[(i + r // HALF) * HALF + r % HALF], k] for r in range(N)])
            c = Ainv[i] * dseg
            J += (c.T * Qs[i] * c)[0, 0]
            for r in range(N):
                coeffs[i][k][r] = c[r]
    return coeffs, J / 2, Hs, Ainv


def to_f(x):
    return float(x)


def coeffs_to_list(c):
    return [[[to_f(v) for v in dim] for dim in seg] for seg in c]


def mellinger_gradient(mask, vals, times, d, h=mp.mpf("0.1")):
    S = len(times)
    _, J0, _, _ = exact_solve(mask, vals, times, d)
    g = []
    for n in range(S):
        tb = []
        for i in range(S):
            t = mp.mpf(times[i]) + (h if i == n else -h / (S - 1))
            tb.append(max(t, mp.mpf("0.01")))
        _, Jb, _, _ = exact_solve(mask, vals, tb, d)
        g.append((Jb - J0) / h)
    return J0, g


def poly_eval(c, t, deriv):
    return sum(base(deriv, k) * c[k] * mp.mpf(t) ** (k - deriv) for k in range(deriv, N))


def max_magnitude(seg_c, T, deriv, dims):
    """max over [0,T] of the norm of derivative `deriv` over dims, via all critical points."""
    # g(t) = sum_dim p^(k) p^(k+1): build coefficient list and find all roots at high precision
    def dcoefs(c, r):
        return [base(r, k) * c[k] for k in range(r, N)]
    acc = None
    for q in dims:
        a = dcoefs(seg_c[q], deriv)
        b = dcoefs(seg_c[q], deriv + 1)
        prod = [mp.mpf(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                prod[i + j] += x * y
        acc = prod if acc is None else [u + v for u, v in zip(acc, prod)]
    # critical points = sign changes of g on a fine grid, bisected to 1e-45 (robust for the
    # degenerate polynomials of symmetric paths where a global root finder stalls)
    def g(t):
        s = mp.mpf(0)
        for cf in reversed(acc):
            s = s * t + cf
        return s
    cands = [mp.mpf(0), mp.mpf(T)]
    M = 4000
    Tm = mp.mpf(T)
    prev_t, prev_g = mp.mpf(0), g(mp.mpf(0))
    for i in range(1, M + 1):
        t = Tm * i / M
        gt = g(t)
        if prev_g == 0:
            cands.append(prev_t)
        elif gt != 0 and (prev_g < 0) != (gt < 0):
            lo, hi, glo = prev_t, t, prev_g
            for _ in range(160):
                mid = (lo + hi) / 2
                gm = g(mid)
                if gm == 0:
                    lo = hi = mid
                    break
                if (gm < 0) == (glo < 0):
                    lo, glo = mid, gm
                else:
                    hi = mid
            cands.append((lo + hi) / 2)
        prev_t, prev_g = t, gt
    best = mp.mpf(0)
    for t in cands:
        m = mp.sqrt(sum(poly_eval(seg_c[q], t, deriv) ** 2 for q in dims))
        best = max(best, m)
    return best


def case_record(name, wp, mask, vals, times, d, with_gradient=False, with_maxima=False, with_blocks=False):
    coeffs, J, Hs, Ainv = exact_solve(mask, vals, times, d)
    rec = dict(name=name, derivative_to_optimize=d, waypoints=np.asarray(wp).tolist(),
               fixed_mask=np.asarray(mask).astype(int).tolist(), fixed_values=np.asarray(vals).tolist(),
               seg_times=[float(t) for t in times], coeffs=coeffs_to_list(coeffs), cost=to_f(J))
    if with_gradient:
        J0, g = mellinger_gradient(mask, vals, times, d)
        rec["gradient"] = [to_f(x) for x in g]
    if with_maxima:
        groups = [[0, 1], [2], [3]]
        rec["maxima"] = [[[to_f(max_magnitude(coeffs[s], times[s], k, grp)) for grp in groups] for k in (1, 2, 3)]
                         for s in range(len(times))]
    if with_blocks:
        rec["H"] = [[[to_f(Hs[s][r, c]) for c in range(N)] for r in range(N)] for s in range(len(times))]
        rec["Ainv"] = [[[to_f(Ainv[s][r, c]) for c in range(N)] for r in range(N)] for s in range(len(times))]
    return rec


def euclid_times(wp, lim):
    """Double-precision Euclidean estimate (input to the fixtures, not a checked quantity)."""
    from oracle import pyoracle
    return pyoracle.estimate_times(wp, lim)


def bench_batch_case():
    """Path 74 of bench.py's own batch (random_batch(1024, 10, seed0=0), Euclidean times): a 0.23 s segment between 8.9 s
    and 4.4 s ones.  It is the path that sets bench.py's max_coeff_err_vs_cpu_ref (2.5e-8..3.9e-8): the fixture shows whose
    error that is -- the reference-style oracle is 2.5e-8 off the 60-digit solution here, the HIP path 2e-9 .. 4e-9
    ((8.9 / 0.23)^7 = 1e11 between neighbouring blocks of R_pp: no double-precision route reaches 1e-11 on this path)."""
    batch = pr.random_batch(1024, 10, seed0=0)
    wp, m, v = batch.path(74)
    return case_record("bench1024_path74_short_segment", wp, m, v, [float(x) for x in euclid_times(wp, batch.limits[74])], 4)


def bench_slot_case():
    """Path 237 of slot 15 of bench.py's twenty batches in flight (random_batch(1024, 10, seed0=15 * 1024), Euclidean times): a
    0.179 s segment between 4.7 s and 4.0 s ones -- the worst-conditioned path of the 20 480 the headline's timed region
    solves.  It sets bench.py's in_flight_slots_vs_cpu_ref.max_coeff_err_vs_cpu_ref (5.3e-7 .. 5.4e-7): against the 60-digit
    solution the reference-style oracle is 5.4e-7 off and the HIP path ~2e-9 (tests/test_gpu_headline_kernel.py)."""
    batch = pr.random_batch(1024, 10, seed0=15 * 1024)
    wp, m, v = batch.path(237)
    return case_record("bench_slot15_path237_short_segment", wp, m, v, [float(x) for x in euclid_times(wp, batch.limits[237])], 4)


def append_missing():
    """Add the cases that the committed fixture file does not hold yet (the others are left as they are)."""
    path = os.path.join(OUT, "linear_qp_cases.json")
    with open(path) as f:
        doc = json.load(f)
    names = {c["name"] for c in doc["cases"]}
    changed = False
    for make in (bench_batch_case, bench_slot_case):
        rec = make()
        if rec["name"] not in names:
            doc["cases"].append(rec)
            changed = True
            print("appended", rec["name"])
    if changed:
        with open(path, "w") as f:
            json.dump(doc, f)


def main():
    if "--append" in sys.argv:
        return append_missing()
    os.makedirs(OUT, exist_ok=True)
    cases = []
    lim = pr.DEFAULT_LIMITS
    # config 1: the reference tests' path, Euclidean times [5, 7.0711, 5]
    wp, m, v = pr.build_vertices(pr.CONFIG1_WAYPOINTS, pr.SNAP)
    t1 = [5.0, float(np.hypot(10.0, 10.0)) / 2.0, 5.0]
    cases.append(case_record("config1_snap", wp, m, v, t1, 4, with_gradient=True, with_maxima=True, with_blocks=True))
    for d in (2, 3):
        wp, m, v = pr.build_vertices(pr.CONFIG1_WAYPOINTS, d)
        cases.append(case_record("config1_d%d" % d, wp, m, v, t1, d, with_gradient=True))
    # closed form: single rest-to-rest segment (n_free = 0 path)
    wp, m, v = pr.build_vertices(np.array([[0.0, 0.0, 0.0, 0.0], [1.0, -2.0, 3.0, 0.5]]), pr.SNAP)
    cases.append(case_record("single_rest_to_rest", wp, m, v, [2.0], 4, with_blocks=True))
    # seeded box paths
    for seed in range(3):
        wp, m, v = pr.build_vertices(pr.random_box_waypoints(10, seed), pr.SNAP)
        t = [float(x) for x in euclid_times(wp, lim)]
        cases.append(case_record("box10_seed%d" % seed, wp, m, v, t, 4, with_gradient=(seed == 0), with_maxima=(seed == 0)))
    wp, m, v = pr.build_vertices(pr.random_box_waypoints(30, 7), pr.SNAP)
    cases.append(case_record("box30_seed7", wp, m, v, [float(x) for x in euclid_times(wp, lim)], 4))
    # short segment times stress conditioning (T in [0.05, 0.6])
    wp, m, v = pr.build_vertices(pr.random_box_waypoints(10, 11), pr.SNAP)
    rng = pr.SplitMix64(99)
    cases.append(case_record("box10_short_times", wp, m, v, [rng.uniform(0.05, 0.6) for _ in range(10)], 4, with_blocks=True))
    # shipping default (minimise acceleration) with an initial state and a stop_at vertex:
    # free derivatives per vertex vary
    init = dict(heading=0.3, velocity=[0.5, -0.2, 0.1, 0.05], acceleration=[0.1, 0.0, -0.1, 0.0], jerk=[0.0, 0.2, 0.0, 0.0])
    stop = [False, False, False, True, False, False, False]
    wp6 = pr.random_box_waypoints(6, 21)
    for d in (2, 3, 4):
        wp, m, v = pr.build_vertices(wp6, d, stop_at=stop, initial_state=init)
        cases.append(case_record("mixed6_d%d" % d, wp, m, v, [float(x) for x in euclid_times(wp, lim)], d,
                                 with_gradient=(d == 2)))
    # random-walk path (PathRandomFlier-like)
    wp, m, v = pr.build_vertices(pr.random_walk_waypoints(10, 5), pr.SNAP)
    cases.append(case_record("walk10_seed5", wp, m, v, [float(x) for x in euclid_times(wp, lim)], 4))
    cases.append(bench_batch_case())
    cases.append(bench_slot_case())
    with open(os.path.join(OUT, "linear_qp_cases.json"), "w") as f:
        json.dump(dict(generator="oracle/gen_golden.py", mp_dps=mp.mp.dps, limits=lim.tolist(), cases=cases), f)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
