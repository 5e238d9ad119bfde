"""ctypes binding of the CPU oracle (oracle/libmrs_tg_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the product package (mrs_uav_trajectory_generation_amd).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmrs_tg_oracle.so")

N, D, HALF = 10, 4, 5


def build(force=False):
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


class _Path(C.Structure):
    _fields_ = [("n_seg", C.c_int), ("derivative_to_optimize", C.c_int),
                ("fixed_mask", C.POINTER(C.c_uint8)), ("fixed_values", C.POINTER(C.c_double))]


class NloptParams(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("f_rel", C.c_double), ("f_abs", C.c_double),
                ("x_rel", C.c_double), ("x_abs", C.c_double)]


class Options(C.Structure):
    _fields_ = [("derivative_to_optimize", C.c_int), ("time_alloc_method", C.c_int),
                ("estimate_times", C.c_int), ("nlopt", NloptParams), ("sampling_dt", C.c_double),
                ("time_penalty", C.c_double), ("use_soft_constraints", C.c_int), ("soft_constraint_weight", C.c_double),
                ("initial_stepsize_rel", C.c_double)]


class DfoParams(C.Structure):
    _fields_ = [("time_alloc_method", C.c_int), ("nlopt", NloptParams), ("time_penalty", C.c_double),
                ("use_soft_constraints", C.c_int), ("soft_constraint_weight", C.c_double), ("initial_stepsize_rel", C.c_double)]


class PolicyParams(C.Structure):
    _fields_ = [("check_deviation_enabled", C.c_int), ("max_deviation", C.c_double), ("max_deviation_iterations", C.c_int),
                ("max_deviation_first_segment", C.c_int), ("min_waypoint_distance", C.c_double),
                ("path_straightener_enabled", C.c_int), ("path_straightener_max_deviation", C.c_double),
                ("path_straightener_max_hdg_deviation", C.c_double), ("max_trajectory_len_factor", C.c_double),
                ("min_trajectory_len_factor", C.c_double), ("fallback_sampling", C.c_int),
                ("fallback_speed_factor", C.c_double), ("fallback_accel_factor", C.c_double),
                ("fallback_stopping_time", C.c_double), ("override_heading_atan2", C.c_int)]


def default_nlopt(max_iterations=10):
    # f_rel 0.05 / x_rel 0.1: /root/reference/src/mrs_trajectory_generation.cpp:884-885;
    # f_abs/x_abs -1 (disabled): polynomial_optimization_nonlinear.h:42-54
    return NloptParams(max_iterations, 0.05, -1.0, 0.1, -1.0)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.mto_base_coeff.restype = C.c_double
        L.mto_base_coeff.argtypes = [C.c_int, C.c_int]
        L.mto_poly_eval.restype = C.c_double
        L.mto_poly_eval.argtypes = [dp, C.c_int, C.c_double, C.c_int]
        L.mto_find_roots_jenkins_traub.restype = C.c_int
        L.mto_find_roots_jenkins_traub.argtypes = [dp, C.c_int, dp, dp]
        L.mto_segment_hessian.argtypes = [C.c_int, C.c_double, dp, dp]
        L.mto_set_arithmetic.restype = None
        L.mto_set_arithmetic.argtypes = [C.c_int]
        L.mto_set_runaway_rule.restype = None
        L.mto_set_runaway_rule.argtypes = [C.c_int]
        L.mto_get_arithmetic.restype = C.c_int
        L.mto_unit_tables.restype = None
        L.mto_unit_tables.argtypes = [dp, dp]
        L.mto_solve_linear.restype = C.c_int
        L.mto_solve_linear.argtypes = [C.POINTER(_Path), dp, dp]
        L.mto_compute_cost.restype = C.c_double
        L.mto_compute_cost.argtypes = [C.c_int, C.c_int, dp, dp]
        L.mto_cost_and_gradient_mellinger.restype = C.c_double
        L.mto_cost_and_gradient_mellinger.argtypes = [C.POINTER(_Path), dp, dp]
        L.mto_set_optimizer_trace.restype = None
        L.mto_set_optimizer_trace.argtypes = [dp, C.c_int]
        L.mto_optimizer_trace_count.restype = C.c_int
        L.mto_optimize_times_mellinger.restype = C.c_int
        L.mto_optimize_times_mellinger.argtypes = [C.POINTER(_Path), C.POINTER(NloptParams), dp, C.POINTER(C.c_int), dp]
        L.mto_segment_max_magnitude.restype = C.c_double
        L.mto_segment_max_magnitude.argtypes = [dp, C.c_double, C.c_int, C.POINTER(C.c_int), C.c_int]
        L.mto_scale_segment_times_to_meet_constraints.restype = C.c_int
        L.mto_scale_segment_times_to_meet_constraints.argtypes = [C.c_int, dp, dp, dp, C.POINTER(C.c_int)]
        L.mto_sample_trajectory.restype = C.c_int
        L.mto_sample_trajectory.argtypes = [C.c_int, dp, dp, C.c_double, C.c_int, dp, C.c_int]
        L.mto_wrap_yaw.restype = C.c_double
        L.mto_wrap_yaw.argtypes = [C.c_double]
        L.mto_estimate_segment_times_euclidean.argtypes = [C.c_int, dp, dp, dp]
        L.mto_estimate_segment_times_baca.argtypes = [C.c_int, dp, dp, dp]
        L.mto_unwrap_heading.restype = C.c_double
        L.mto_unwrap_heading.argtypes = [C.c_double, C.c_double]
        L.mto_solve_batch.restype = C.c_int
        L.mto_solve_batch.argtypes = [C.c_int, C.POINTER(C.c_int32), dp, C.POINTER(C.c_uint8), dp, dp,
                                      C.POINTER(Options), dp, dp, C.POINTER(C.c_int32), dp,
                                      C.POINTER(C.c_int32), dp, C.c_int, C.c_int]
        L.mto_max_of_magnitude.restype = C.c_double
        L.mto_max_of_magnitude.argtypes = [C.c_int, dp, dp, C.c_int]
        L.mto_objective_time.restype = C.c_double
        L.mto_objective_time.argtypes = [C.POINTER(_Path), dp, dp, C.POINTER(DfoParams), dp]
        L.mto_optimize_time_dfo.restype = C.c_int
        L.mto_optimize_time_dfo.argtypes = [C.POINTER(_Path), dp, C.POINTER(DfoParams), dp, C.POINTER(C.c_int), dp]
        L.mto_count_free_constraints.argtypes = [C.POINTER(_Path)]
        L.mto_solve_linear_free.argtypes = [C.POINTER(_Path), dp, dp, dp]
        L.mto_coeffs_from_free_constraints.argtypes = [C.POINTER(_Path), dp, dp, dp]
        L.mto_objective_time_and_constraints.restype = C.c_double
        L.mto_objective_time_and_constraints.argtypes = [C.POINTER(_Path), dp, dp, C.POINTER(DfoParams), dp, dp]
        L.mto_free_derivative_bounds.argtypes = [C.POINTER(_Path), dp, dp, dp]
        L.mto_optimize_time_and_constraints_dfo.restype = C.c_int
        L.mto_optimize_time_and_constraints_dfo.argtypes = [C.POINTER(_Path), dp, C.POINTER(DfoParams), dp, dp,
                                                            C.POINTER(C.c_int), dp]
        L.mto_default_policy_params.argtypes = [C.POINTER(PolicyParams)]
        L.mto_dist_from_segment.restype = C.c_double
        L.mto_dist_from_segment.argtypes = [dp, dp, dp]
        L.mto_preprocess_path.restype = C.c_int
        L.mto_preprocess_path.argtypes = [dp, C.POINTER(C.c_uint8), C.c_int, C.POINTER(PolicyParams), dp, C.POINTER(C.c_uint8)]
        L.mto_validate_trajectory_spatial.restype = C.c_int
        L.mto_validate_trajectory_spatial.argtypes = [dp, C.c_int, dp, C.c_int, C.POINTER(PolicyParams), C.POINTER(C.c_uint8), dp]
        L.mto_waypoint_trajectory_idxs.restype = C.c_int
        L.mto_waypoint_trajectory_idxs.argtypes = [dp, C.c_int, dp, C.c_int, C.POINTER(C.c_int32)]
        L.mto_fallback_sampling.restype = C.c_int
        L.mto_fallback_sampling.argtypes = [dp, C.POINTER(C.c_uint8), C.c_int, dp, C.c_int, C.POINTER(PolicyParams), C.c_double, dp, C.c_int]
        L.mto_optimize_path.restype = C.c_int
        L.mto_optimize_path.argtypes = [dp, C.POINTER(C.c_uint8), C.c_int, dp, dp, C.c_int, C.POINTER(Options),
                                        C.POINTER(PolicyParams), dp, C.c_int, C.POINTER(C.c_int), dp, C.POINTER(C.c_int),
                                        C.POINTER(C.c_int)]
        _lib = L
    return _lib


def make_options(deriv, time_alloc_method, estimate_times, max_iterations, sampling_dt, time_penalty=100.0,
                 use_soft_constraints=1, soft_constraint_weight=1.5, initial_stepsize_rel=0.1):
    # time_penalty 100, soft weight 1.5: /root/reference/config/private/trajectory_generation.yaml:4-6
    return Options(deriv, time_alloc_method, int(estimate_times), default_nlopt(max_iterations), float(sampling_dt),
                   float(time_penalty), int(use_soft_constraints), float(soft_constraint_weight), float(initial_stepsize_rel))


def default_policy(**overrides):
    p = PolicyParams()
    lib().mto_default_policy_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise TypeError(k)
        setattr(p, k, v)
    return p


def optimize_path(waypoints, stop_at=None, initial_state=None, limits=None, relax_heading=False, policy=None, deriv=2,
                  time_alloc_method=2, max_iterations=10, sampling_dt=0.2, capacity=4096):
    """optimize() for one path. Returns dict(success, samples, n_samples, max_deviation, n_waypoints, iterations)."""
    w = _f64(waypoints).reshape(-1, 4)
    n = w.shape[0]
    st = np.ascontiguousarray(stop_at if stop_at is not None else np.zeros(n), dtype=np.uint8)
    init = None
    if initial_state is not None:
        init = _f64(np.concatenate([[initial_state["heading"]], initial_state["velocity"], initial_state["acceleration"],
                                    initial_state["jerk"]]))
    lim = _f64(limits)
    pol = policy or default_policy()
    opt = make_options(deriv, time_alloc_method, 1, max_iterations, sampling_dt)
    out = np.zeros((capacity, 4))
    ns, nw, it = C.c_int(0), C.c_int(0), C.c_int(0)
    md = C.c_double(0)
    ok = lib().mto_optimize_path(_dp(w), st.ctypes.data_as(C.POINTER(C.c_uint8)), n, _dp(init) if init is not None else None,
                                 _dp(lim), int(bool(relax_heading)), C.byref(opt), C.byref(pol), _dp(out), capacity,
                                 C.byref(ns), C.byref(md), C.byref(nw), C.byref(it))
    return dict(success=int(ok), samples=out[:ns.value].copy(), n_samples=ns.value, max_deviation=md.value,
                n_waypoints=nw.value, iterations=it.value)


def find_trajectory(waypoints, stop_at=None, initial_state=None, limits=None, relax_heading=False, policy=None, deriv=2,
                    time_alloc_method=2, max_iterations=10, sampling_dt=0.2, capacity=4096):
    """findTrajectory() for one path, both gates included (mto_find_trajectory). Returns dict(success, samples, n_samples,
    times, status, baca_total_time, raw_n_samples, rejection)."""
    w = _f64(waypoints).reshape(-1, 4)
    n = w.shape[0]
    st = np.ascontiguousarray(stop_at if stop_at is not None else np.zeros(n), dtype=np.uint8)
    init = None
    if initial_state is not None:
        init = _f64(np.concatenate([[initial_state["heading"]], initial_state["velocity"], initial_state["acceleration"],
                                    initial_state["jerk"]]))
    lim = _f64(limits)
    pol = policy or default_policy()
    opt = make_options(deriv, time_alloc_method, 1, max_iterations, sampling_dt)
    out = np.zeros((capacity, 4))
    times = np.zeros(n - 1)
    ns, raw, rej, status = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int32(0)
    baca = C.c_double(0)
    f = lib().mto_find_trajectory
    f.restype = C.c_int
    ok = f(_dp(w), st.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int(n), _dp(init) if init is not None else None, _dp(lim),
           C.c_int(int(bool(relax_heading))), C.byref(opt), C.byref(pol), _dp(out), C.c_int(capacity), C.byref(ns), _dp(times),
           C.byref(status), C.byref(baca), C.byref(raw), C.byref(rej))
    return dict(success=int(ok), samples=out[:ns.value].copy(), n_samples=ns.value, times=times, status=status.value,
                baca_total_time=baca.value, raw_n_samples=raw.value, rejection=rej.value)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _make_path(n_seg, deriv, fixed_mask, fixed_values):
    m = np.ascontiguousarray(fixed_mask, dtype=np.uint8).reshape(-1)
    v = _f64(fixed_values).reshape(-1)
    assert m.size == (n_seg + 1) * 5 and v.size == (n_seg + 1) * 20
    p = _Path(n_seg, deriv, m.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(v))
    p._keep = (m, v)
    return p


REFERENCE_ARITHMETIC, EXACT_CONSTANTS, QUAD_PRECISION = 0, 1, 2


class arithmetic:
    """with po.arithmetic(po.EXACT_CONSTANTS): ...  -- the oracle's per-segment matrices from exactly rounded unit-time
    tables instead of the reference's numerically inverted mapping matrix (mrs_tg_oracle.h); process-wide."""

    def __init__(self, mode):
        self.mode = int(mode)

    def __enter__(self):
        self.prev = lib().mto_get_arithmetic()
        lib().mto_set_arithmetic(self.mode)
        return self

    def __exit__(self, *exc):
        lib().mto_set_arithmetic(self.prev)
        return False


def unit_tables():
    """(ABAR^-1 [10, 10], HBAR [5, 10, 10]) of the exact-constants route"""
    a = np.zeros((10, 10))
    h = np.zeros((5, 10, 10))
    lib().mto_unit_tables(_dp(a), _dp(h))
    return a, h


def segment_hessian(deriv, T):
    H = np.zeros((N, N))
    Ai = np.zeros((N, N))
    lib().mto_segment_hessian(deriv, float(T), _dp(H), _dp(Ai))
    return H, Ai


def solve_linear(deriv, fixed_mask, fixed_values, seg_times):
    t = _f64(seg_times)
    S = t.size
    p = _make_path(S, deriv, fixed_mask, fixed_values)
    c = np.zeros((S, D, N))
    rc = lib().mto_solve_linear(C.byref(p), _dp(t), _dp(c))
    assert rc == 0, rc
    return c


def compute_cost(deriv, seg_times, coeffs):
    t = _f64(seg_times)
    c = _f64(coeffs)
    return lib().mto_compute_cost(t.size, deriv, _dp(t), _dp(c))


def cost_and_gradient(deriv, fixed_mask, fixed_values, seg_times):
    t = _f64(seg_times)
    p = _make_path(t.size, deriv, fixed_mask, fixed_values)
    g = np.zeros(t.size)
    J = lib().mto_cost_and_gradient_mellinger(C.byref(p), _dp(t), _dp(g))
    return J, g


def optimize_times(deriv, fixed_mask, fixed_values, seg_times, params=None):
    t = _f64(seg_times).copy()
    p = _make_path(t.size, deriv, fixed_mask, fixed_values)
    prm = params or default_nlopt()
    ne = C.c_int(0)
    fc = C.c_double(0)
    rc = lib().mto_optimize_times_mellinger(C.byref(p), C.byref(prm), _dp(t), C.byref(ne), C.byref(fc))
    return rc, t, ne.value, fc.value


def optimize_times_traced(deriv, fixed_mask, fixed_values, seg_times, params=None, cap=64):
    """optimize_times plus the search's decision trace: rows of [evaluation, f, fn, slope, alpha, Armijo margin, ftol margin,
    xtol margin] (mto_set_optimizer_trace)"""
    buf = np.zeros((cap, 8))
    lib().mto_set_optimizer_trace(_dp(buf), cap)
    try:
        rc, t, ne, fc = optimize_times(deriv, fixed_mask, fixed_values, seg_times, params)
        n = lib().mto_optimizer_trace_count()
    finally:
        lib().mto_set_optimizer_trace(None, 0)
    return rc, t, ne, buf[:n].copy()


def objective_time(deriv, fixed_mask, fixed_values, seg_times, limits, mode=0, time_penalty=100.0, soft=1, weight=1.5):
    t = _f64(seg_times)
    p = _make_path(t.size, deriv, fixed_mask, fixed_values)
    prm = DfoParams(mode, default_nlopt(), float(time_penalty), int(soft), float(weight), 0.1)
    parts = np.zeros(3)
    lim = _f64(limits)
    f = lib().mto_objective_time(C.byref(p), _dp(t), _dp(lim), C.byref(prm), _dp(parts))
    return f, parts


def optimize_times_dfo(deriv, fixed_mask, fixed_values, seg_times, limits, mode=0, max_iterations=10, time_penalty=100.0,
                       soft=1, weight=1.5):
    t = _f64(seg_times).copy()
    p = _make_path(t.size, deriv, fixed_mask, fixed_values)
    prm = DfoParams(mode, default_nlopt(max_iterations), float(time_penalty), int(soft), float(weight), 0.1)
    ne = C.c_int(0)
    fl = C.c_double(0)
    lim = _f64(limits)
    rc = lib().mto_optimize_time_dfo(C.byref(p), _dp(lim), C.byref(prm), _dp(t), C.byref(ne), C.byref(fl))
    return rc, t, ne.value, fl.value


def solve_linear_free(deriv, fixed_mask, fixed_values, seg_times):
    """solveLinear + getFreeConstraints: (coeffs [S][4][10], free [4][n_free])"""
    t = _f64(seg_times)
    p = _make_path(t.size, deriv, fixed_mask, fixed_values)
    nf = lib().mto_count_free_constraints(C.byref(p))
    coeffs = np.zeros((t.size, D, N))
    free = np.zeros((D, nf))
    rc = lib().mto_solve_linear_free(C.byref(p), _dp(t), _dp(coeffs), _dp(free))
    assert rc == 0
    return coeffs, free


def coeffs_from_free(deriv, fixed_mask, fixed_values, seg_times, free):
    t = _f64(seg_times)
    p = _make_path(t.size, deriv, fixed_mask, fixed_values)
    f = _f64(free)
    coeffs = np.zeros((t.size, D, N))
    lib().mto_coeffs_from_free_constraints(C.byref(p), _dp(t), _dp(f), _dp(coeffs))
    return coeffs


def objective_time_and_constraints(deriv, fixed_mask, fixed_values, x, limits, mode=3, time_penalty=100.0, soft=1,
                                   weight=1.5):
    x = _f64(x)
    n_seg = np.asarray(fixed_mask).reshape(-1, 5).shape[0] - 1
    p = _make_path(n_seg, deriv, fixed_mask, fixed_values)
    prm = DfoParams(mode, default_nlopt(), float(time_penalty), int(soft), float(weight), 0.1)
    parts = np.zeros(3)
    lim = _f64(limits)
    f = lib().mto_objective_time_and_constraints(C.byref(p), _dp(x), _dp(lim), C.byref(prm), None, _dp(parts))
    return f, parts


def free_derivative_bounds(deriv, fixed_mask, fixed_values, limits):
    n_seg = np.asarray(fixed_mask).reshape(-1, 5).shape[0] - 1
    p = _make_path(n_seg, deriv, fixed_mask, fixed_values)
    nf = lib().mto_count_free_constraints(C.byref(p))
    lo = np.zeros((D, nf))
    hi = np.zeros((D, nf))
    lim = _f64(limits)
    lib().mto_free_derivative_bounds(C.byref(p), _dp(lim), _dp(lo), _dp(hi))
    return lo, hi


def optimize_time_and_constraints_dfo(deriv, fixed_mask, fixed_values, seg_times, limits, mode=3, max_iterations=10,
                                      time_penalty=100.0, soft=1, weight=1.5):
    t = _f64(seg_times).copy()
    p = _make_path(t.size, deriv, fixed_mask, fixed_values)
    prm = DfoParams(mode, default_nlopt(max_iterations), float(time_penalty), int(soft), float(weight), 0.1)
    ne = C.c_int(0)
    fl = C.c_double(0)
    lim = _f64(limits)
    coeffs = np.zeros((t.size, D, N))
    rc = lib().mto_optimize_time_and_constraints_dfo(C.byref(p), _dp(lim), C.byref(prm), _dp(t), _dp(coeffs),
                                                     C.byref(ne), C.byref(fl))
    return rc, t, coeffs, ne.value, fl.value


def max_of_magnitude(coeffs, seg_times, derivative):
    c = _f64(coeffs)
    t = _f64(seg_times)
    return lib().mto_max_of_magnitude(t.size, _dp(c), _dp(t), derivative)


def find_roots(coeffs_increasing):
    c = _f64(coeffs_increasing)
    re = np.zeros(128)
    im = np.zeros(128)
    n = lib().mto_find_roots_jenkins_traub(_dp(c), c.size, _dp(re), _dp(im))
    if n <= 0:
        return np.zeros(0, dtype=complex)
    return re[:n] + 1j * im[:n]


def segment_max_magnitude(seg_coeffs, T, derivative, dims):
    c = _f64(seg_coeffs)
    dd = (C.c_int * len(dims))(*dims)
    return lib().mto_segment_max_magnitude(_dp(c), float(T), derivative, dd, len(dims))


def scale_segment_times(coeffs, seg_times, limits):
    c = _f64(coeffs).copy()
    t = _f64(seg_times).copy()
    lim = _f64(limits)
    sw = C.c_int(0)
    ok = lib().mto_scale_segment_times_to_meet_constraints(t.size, _dp(c), _dp(t), _dp(lim), C.byref(sw))
    return ok, c, t, sw.value


def sample_trajectory(coeffs, seg_times, dt, derivative=0, capacity=4096):
    c = _f64(coeffs)
    t = _f64(seg_times)
    out = np.zeros((capacity, D))
    n = lib().mto_sample_trajectory(t.size, _dp(c), _dp(t), float(dt), derivative, _dp(out), capacity)
    return out[:min(n, capacity)].copy(), n


def wrap_yaw(y):
    return lib().mto_wrap_yaw(float(y))


def estimate_times(waypoints, limits, baca=False):
    w = _f64(waypoints)
    S = w.shape[0] - 1
    out = np.zeros(S)
    lim = _f64(limits)
    f = lib().mto_estimate_segment_times_baca if baca else lib().mto_estimate_segment_times_euclidean
    f(S, _dp(w), _dp(lim), _dp(out))
    return out


def unwrap_heading(what, frm):
    return lib().mto_unwrap_heading(float(what), float(frm))


def solve_batch(seg_offsets, waypoints, fixed_mask, fixed_values, limits, seg_times, *, deriv=4,
                time_alloc_method=-1, estimate_times=False, max_iterations=10, sampling_dt=0.0,
                sample_capacity=0, n_threads=1, time_penalty=100.0, use_soft_constraints=1, soft_constraint_weight=1.5,
                runaway_rule=False):
    """Batch driver in the C-ABI's CSR layout. Returns dict(times, coeffs, status, cost, n_samples, samples).
    runaway_rule: False (the default since round 5) = the REFERENCE's behaviour, a runaway of the feasibility scaling keeps
    the outer loop's own code; True = report it as ROUNDOFF_LIMITED (-4) like the product does (its documented deviation,
    mto_set_runaway_rule).  The parity tests compare on the reference's rule (tests/util.py::status_matches)."""
    so = np.ascontiguousarray(seg_offsets, dtype=np.int32)
    P = so.size - 1
    total_S = int(so[-1])
    w = _f64(waypoints)
    m = np.ascontiguousarray(fixed_mask, dtype=np.uint8)
    v = _f64(fixed_values)
    lim = _f64(limits)
    t = _f64(seg_times).copy()
    coeffs = np.zeros((total_S, D, N))
    status = np.zeros(P, dtype=np.int32)
    cost = np.zeros(P)
    ns = np.zeros(P, dtype=np.int32)
    samples = np.zeros((P, max(sample_capacity, 1), D))
    opt = make_options(deriv, time_alloc_method, estimate_times, max_iterations, sampling_dt, time_penalty,
                       use_soft_constraints, soft_constraint_weight)
    lib().mto_set_runaway_rule(1 if runaway_rule else 0)
    lib().mto_solve_batch(P, so.ctypes.data_as(C.POINTER(C.c_int32)), _dp(w), m.ctypes.data_as(C.POINTER(C.c_uint8)),
                          _dp(v), _dp(lim), C.byref(opt), _dp(t), _dp(coeffs),
                          status.ctypes.data_as(C.POINTER(C.c_int32)), _dp(cost),
                          ns.ctypes.data_as(C.POINTER(C.c_int32)), _dp(samples) if sample_capacity > 0 else None,
                          sample_capacity, n_threads)
    lib().mto_set_runaway_rule(0)
    return dict(times=t, coeffs=coeffs, status=status, cost=cost, n_samples=ns,
                samples=samples if sample_capacity > 0 else None)
