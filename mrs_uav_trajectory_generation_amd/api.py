"""Python host binding of the C ABI (include/mrs_tg.h) -- ctypes only, no numerics here.

The library is the product; this module only marshals buffers.  There is no CPU fallback: if
libmrs_tg.so is missing or no HIP device is usable, the calls raise.

Two ways in, mirroring the ABI:
  * Context.solve_batch(batch, seg_times, ...)  host numpy arrays in / out (the nodelet-style call,
    /root/reference/src/mrs_trajectory_generation.cpp:1046-1169 for a whole batch);
  * Plan(...)  analysis once, then device-resident torch tensors, asynchronous on torch's current
    stream (what bench.py times).
"""
import ctypes as C
import os

import numpy as np

from .problem import Batch, N_COEFF, N_DIM

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MRS_TG_LIB_PATH") or os.path.join(_HERE, "libmrs_tg.so")   # (the override: experiment builds)

TIME_ALLOC_NONE = -1
TIME_ALLOC_SQUARED_TIME = 0
TIME_ALLOC_RICHTER_TIME = 1
TIME_ALLOC_MELLINGER = 2
TIME_ALLOC_SQUARED_TIME_AND_CONSTRAINTS = 3
TIME_ALLOC_RICHTER_TIME_AND_CONSTRAINTS = 4
FLAG_FUSED_ASSEMBLY = 1        # the default since ABI 2
FLAG_MATERIALIZED_BLOCKS = 2   # assembly kernel + solve from the materialised H / A^-1 blocks
FLAG_GENERAL_PATTERNS = 16     # vertices without a position constraint may occur (general 5 x 5 route, every mode)
FLAG_CAREFUL_COST = 8          # Mellinger mode: re-run the paths whose fast cost evaluation failed its guard with primal costs
FLAG_SHARED_DEVICE = 4         # hint: several batches are in flight on this device (results unaffected)
FLAG_POSITIONS_ARE_WAYPOINTS = 32   # every vertex's position constraint is its waypoint (checked at bind time): read the compact array
FLAG_REFERENCE_STATUS = 128         # Mellinger: the outer loop's own code, no runaway rule (MRS_TG_FLAG_REFERENCE_STATUS)
FLAG_CONSTRAINED_SLOTS = 64         # hint: interior vertices may hold constrained derivative slots (stop_at) under min-snap

STATUS_ROUNDOFF_LIMITED = -4   # MRS_TG_STATUS_ROUNDOFF_LIMITED: the feasibility scaling ran away (include/mrs_tg.h)
RUNAWAY_TIME_FACTOR = 25.0     # MRS_TG_RUNAWAY_TIME_FACTOR

# mrs_tg_find_trajectory_info: which of findTrajectory's gates discarded the trajectory (MRS_TG_FIND_*)
FIND_ACCEPTED, FIND_REJECTED_CODE, FIND_REJECTED_TOO_LONG, FIND_REJECTED_TOO_SHORT = 0, 1, 2, 3

STATE_ORDERS = 5   # derivative orders 0..4 per sample of Plan.sample_states (MRS_TG_STATE_ORDERS)
KERNEL_ASSEMBLE, KERNEL_SOLVE_LINEAR, KERNEL_NONLINEAR = 0, 1, 2


class MrsTgError(RuntimeError):
    pass


class Options(C.Structure):
    _fields_ = [("derivative_to_optimize", C.c_int32), ("time_alloc_method", C.c_int32),
                ("estimate_times", C.c_int32), ("max_iterations", C.c_int32),
                ("f_rel", C.c_double), ("f_abs", C.c_double), ("x_rel", C.c_double), ("x_abs", C.c_double),
                ("sampling_dt", C.c_double), ("sample_capacity", C.c_int32), ("flags", C.c_int32),
                ("time_penalty", C.c_double), ("soft_constraint_weight", C.c_double),
                ("use_soft_constraints", C.c_int32), ("reserved_", C.c_int32), ("initial_stepsize_rel", C.c_double),
                ("max_time_s", C.c_double),
                ("max_trajectory_len_factor", C.c_double), ("min_trajectory_len_factor", C.c_double)]   # (ABI 5)


class PolicyOptions(C.Structure):
    _fields_ = [("solver", Options), ("check_deviation_enabled", C.c_int32), ("max_deviation", C.c_double),
                ("max_deviation_iterations", C.c_int32), ("max_deviation_first_segment", C.c_int32),
                ("min_waypoint_distance", C.c_double), ("path_straightener_enabled", C.c_int32),
                ("path_straightener_max_deviation", C.c_double), ("path_straightener_max_hdg_deviation", C.c_double),
                ("max_trajectory_len_factor", C.c_double), ("min_trajectory_len_factor", C.c_double),
                ("fallback_sampling", C.c_int32), ("fallback_speed_factor", C.c_double),
                ("fallback_accel_factor", C.c_double), ("fallback_stopping_time", C.c_double),
                ("override_heading_atan2", C.c_int32), ("reserved_", C.c_int32), ("max_execution_time_s", C.c_double)]


class Waypoint(C.Structure):
    _fields_ = [("coords", C.c_double * 4), ("stop_at", C.c_uint8)]


class InitialState(C.Structure):
    _fields_ = [("heading", C.c_double), ("velocity", C.c_double * 4), ("acceleration", C.c_double * 4),
                ("jerk", C.c_double * 4)]


EXPORTED_SYMBOLS = [
    "mrs_tg_create", "mrs_tg_destroy", "mrs_tg_last_error", "mrs_tg_abi_version", "mrs_tg_capabilities", "mrs_tg_default_options",
    "mrs_tg_kernel_trace_reset", "mrs_tg_kernel_trace", "mrs_tg_plan_explain",
    "mrs_tg_set_stream", "mrs_tg_reset_stream", "mrs_tg_synchronize", "mrs_tg_solve_batch", "mrs_tg_plan_create", "mrs_tg_plan_destroy",
    "mrs_tg_plan_n_paths", "mrs_tg_plan_n_segments", "mrs_tg_plan_max_segments", "mrs_tg_plan_get_order",
    "mrs_tg_plan_assemble", "mrs_tg_plan_block_bytes", "mrs_tg_plan_solve", "mrs_tg_plan_bind_solve",
    "mrs_tg_bound_solve_launch", "mrs_tg_bound_solve_launch_many", "mrs_tg_bound_solve_launch_many_mt", "mrs_tg_bound_solve_launch_group",
    "mrs_tg_bound_solve_destroy", "mrs_tg_host_alloc", "mrs_tg_host_free", "mrs_tg_host_register",
    "mrs_tg_host_unregister", "mrs_tg_plan_cost_gradient",
    "mrs_tg_plan_segment_maxima", "mrs_tg_plan_sample_states", "mrs_tg_plan_careful_count", "mrs_tg_set_profiling", "mrs_tg_last_kernel_ms", "mrs_tg_kernel_ms_history",
    "mrs_tg_find_trajectory", "mrs_tg_find_trajectory_info", "mrs_tg_estimate_times_baca",
    "mrs_tg_default_policy_options", "mrs_tg_optimize_paths", "mrs_tg_waypoint_trajectory_idxs",
    "mrs_tg_create_multi", "mrs_tg_destroy_multi", "mrs_tg_multi_n_devices", "mrs_tg_multi_context", "mrs_tg_multi_shard",
    "mrs_tg_multi_solve_batch", "mrs_tg_multi_last_error",
]

_lib = None


def load_library():
    """dlopen libmrs_tg.so (built in-tree by mrs_uav_trajectory_generation_amd.build). Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MrsTgError("HIP extension %s is missing: run `python -m mrs_uav_trajectory_generation_amd.build` "
                         "(there is no CPU fallback)" % LIB_PATH)
    # PyTorch wheels bundle their own libamdhip64.so.7 / libhsa-runtime64 pair.  The dynamic loader keeps ONE library per
    # SONAME: if libmrs_tg.so pulled in the system libamdhip64.so.7 first, torch would later run its bundled HSA runtime
    # under the system HIP runtime and one of the two fails with "no ROCm-capable device".  In a process that uses both,
    # torch therefore has to be loaded first; a host without torch (the C++ hosts) simply uses the system runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, dp, ip, bp = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p  # raw addresses (host or device)
    L.mrs_tg_create.restype = C.c_int
    L.mrs_tg_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.mrs_tg_destroy.restype = None
    L.mrs_tg_destroy.argtypes = [vp]
    L.mrs_tg_last_error.restype = C.c_char_p
    L.mrs_tg_last_error.argtypes = [vp]
    L.mrs_tg_abi_version.restype = C.c_int
    L.mrs_tg_capabilities.restype = C.c_int
    L.mrs_tg_kernel_trace_reset.restype = None
    L.mrs_tg_kernel_trace.restype = C.c_int
    L.mrs_tg_kernel_trace.argtypes = [C.POINTER(C.c_char_p), C.c_int]
    L.mrs_tg_plan_explain.restype = C.c_int
    L.mrs_tg_plan_explain.argtypes = [vp, C.POINTER(Options), C.c_int32, C.POINTER(C.c_char_p), C.c_int32]
    L.mrs_tg_default_options.restype = None
    L.mrs_tg_default_options.argtypes = [C.POINTER(Options)]
    L.mrs_tg_host_alloc.restype = C.c_int
    L.mrs_tg_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.mrs_tg_host_free.restype = None
    L.mrs_tg_host_free.argtypes = [vp]
    L.mrs_tg_host_register.restype = C.c_int
    L.mrs_tg_host_register.argtypes = [vp, C.c_size_t]
    L.mrs_tg_host_unregister.restype = C.c_int
    L.mrs_tg_host_unregister.argtypes = [vp]
    L.mrs_tg_set_stream.restype = C.c_int
    L.mrs_tg_set_stream.argtypes = [vp, vp]
    L.mrs_tg_reset_stream.restype = C.c_int
    L.mrs_tg_reset_stream.argtypes = [vp]
    L.mrs_tg_synchronize.restype = C.c_int
    L.mrs_tg_synchronize.argtypes = [vp]
    L.mrs_tg_solve_batch.restype = C.c_int
    L.mrs_tg_solve_batch.argtypes = [vp, C.c_int32, ip, dp, bp, dp, dp, C.POINTER(Options), dp, dp, ip, dp, ip, dp]
    L.mrs_tg_plan_create.restype = C.c_int
    L.mrs_tg_plan_create.argtypes = [vp, C.c_int32, ip, C.POINTER(vp)]
    L.mrs_tg_plan_destroy.restype = None
    L.mrs_tg_plan_destroy.argtypes = [vp]
    for name in ("mrs_tg_plan_n_paths", "mrs_tg_plan_n_segments", "mrs_tg_plan_max_segments"):
        getattr(L, name).restype = C.c_int32
        getattr(L, name).argtypes = [vp]
    L.mrs_tg_plan_get_order.restype = C.c_int
    L.mrs_tg_plan_get_order.argtypes = [vp, ip]
    L.mrs_tg_plan_assemble.restype = C.c_int
    L.mrs_tg_plan_assemble.argtypes = [vp, C.c_int32, dp, dp, dp]
    L.mrs_tg_plan_block_bytes.restype = C.c_size_t
    L.mrs_tg_plan_block_bytes.argtypes = [vp]
    L.mrs_tg_plan_solve.restype = C.c_int
    L.mrs_tg_plan_solve.argtypes = [vp, dp, bp, dp, dp, C.POINTER(Options), dp, dp, ip, dp, ip, dp]
    L.mrs_tg_plan_bind_solve.restype = C.c_int
    L.mrs_tg_plan_bind_solve.argtypes = [vp, dp, bp, dp, dp, C.POINTER(Options), dp, dp, ip, dp, ip, dp, C.POINTER(vp)]
    L.mrs_tg_bound_solve_launch.restype = C.c_int
    L.mrs_tg_bound_solve_launch.argtypes = [vp]
    L.mrs_tg_bound_solve_launch_many.restype = C.c_int
    L.mrs_tg_bound_solve_launch_many.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32]
    L.mrs_tg_bound_solve_launch_many_mt.restype = C.c_int
    L.mrs_tg_bound_solve_launch_many_mt.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, C.c_int32]
    L.mrs_tg_bound_solve_launch_group.restype = C.c_int
    L.mrs_tg_bound_solve_launch_group.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32]
    L.mrs_tg_bound_solve_destroy.restype = None
    L.mrs_tg_bound_solve_destroy.argtypes = [vp]
    L.mrs_tg_plan_cost_gradient.restype = C.c_int
    L.mrs_tg_plan_cost_gradient.argtypes = [vp, C.c_int32, bp, dp, dp, dp, dp]
    L.mrs_tg_plan_segment_maxima.restype = C.c_int
    L.mrs_tg_plan_segment_maxima.argtypes = [vp, dp, dp, dp]
    L.mrs_tg_plan_careful_count.restype = C.c_int
    L.mrs_tg_plan_careful_count.argtypes = [vp, ip]
    L.mrs_tg_plan_sample_states.restype = C.c_int
    L.mrs_tg_plan_sample_states.argtypes = [vp, dp, dp, C.c_double, C.c_int32, ip, dp]
    L.mrs_tg_set_profiling.restype = C.c_int
    L.mrs_tg_set_profiling.argtypes = [vp, C.c_int]
    L.mrs_tg_last_kernel_ms.restype = C.c_int
    L.mrs_tg_last_kernel_ms.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.mrs_tg_kernel_ms_history.restype = C.c_int
    L.mrs_tg_kernel_ms_history.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.c_int]
    L.mrs_tg_find_trajectory_info.restype = C.c_int
    L.mrs_tg_find_trajectory_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.mrs_tg_estimate_times_baca.restype = C.c_int
    L.mrs_tg_estimate_times_baca.argtypes = [dp, C.c_int32, dp, dp]
    L.mrs_tg_find_trajectory.restype = C.c_int
    L.mrs_tg_find_trajectory.argtypes = [vp, C.POINTER(Waypoint), C.c_int32, C.POINTER(InitialState), dp,
                                         C.POINTER(Options), C.c_int32, dp, dp, ip, ip, dp]
    L.mrs_tg_default_policy_options.restype = None
    L.mrs_tg_default_policy_options.argtypes = [C.POINTER(PolicyOptions)]
    L.mrs_tg_optimize_paths.restype = C.c_int
    L.mrs_tg_optimize_paths.argtypes = [vp, C.c_int32, ip, C.POINTER(Waypoint), C.POINTER(InitialState), bp, dp, bp,
                                        C.POINTER(PolicyOptions), C.c_int32, ip, ip, dp, dp, ip, ip]
    L.mrs_tg_waypoint_trajectory_idxs.restype = C.c_int32
    L.mrs_tg_waypoint_trajectory_idxs.argtypes = [dp, C.c_int32, C.POINTER(Waypoint), C.c_int32, ip]
    L.mrs_tg_create_multi.restype = C.c_int
    L.mrs_tg_create_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.mrs_tg_destroy_multi.restype = None
    L.mrs_tg_destroy_multi.argtypes = [vp]
    L.mrs_tg_multi_n_devices.restype = C.c_int
    L.mrs_tg_multi_n_devices.argtypes = [vp]
    L.mrs_tg_multi_context.restype = vp
    L.mrs_tg_multi_context.argtypes = [vp, C.c_int]
    L.mrs_tg_multi_shard.restype = C.c_int
    L.mrs_tg_multi_shard.argtypes = [vp, C.c_int32, ip, ip]
    L.mrs_tg_multi_solve_batch.restype = C.c_int
    L.mrs_tg_multi_solve_batch.argtypes = [vp, C.c_int32, ip, dp, bp, dp, dp, C.POINTER(Options), dp, dp, ip, dp, ip, dp]
    L.mrs_tg_multi_last_error.restype = C.c_char_p
    L.mrs_tg_multi_last_error.argtypes = [vp]
    _lib = L
    return L


CAP_CAREFUL_COST = 1   # MRS_TG_CAP_CAREFUL_COST


def capabilities():
    return load_library().mrs_tg_capabilities()


def kernel_trace_reset():
    load_library().mrs_tg_kernel_trace_reset()


def kernel_trace():
    """Names of the kernels this thread has launched through the library since kernel_trace_reset() (newest 32, oldest first)."""
    buf = (C.c_char_p * 32)()
    n = load_library().mrs_tg_kernel_trace(buf, 32)
    return [buf[i].decode() for i in range(n)]


def default_policy_options(solver=None, **overrides):
    opt = PolicyOptions()
    load_library().mrs_tg_default_policy_options(C.byref(opt))
    for k, v in (solver or {}).items():
        if not hasattr(opt.solver, k):
            raise TypeError("unknown solver option %r" % k)
        setattr(opt.solver, k, v)
    for k, v in overrides.items():
        if not hasattr(opt, k):
            raise TypeError("unknown policy option %r" % k)
        setattr(opt, k, v)
    return opt


def estimate_times_baca(waypoints, limits9):
    """estimateSegmentTimesBaca for one path (mrs_tg_estimate_times_baca): waypoints [V][4], headings already unwrapped"""
    wp = np.ascontiguousarray(waypoints, dtype=np.float64).reshape(-1, 4)
    lim = np.ascontiguousarray(limits9, dtype=np.float64)
    out = np.zeros(wp.shape[0] - 1)
    rc = load_library().mrs_tg_estimate_times_baca(_np_ptr(wp), wp.shape[0], _np_ptr(lim), _np_ptr(out))
    if rc != 0:
        raise MrsTgError("mrs_tg_estimate_times_baca failed (%d)" % rc)
    return out


def default_options(**overrides):
    opt = Options()
    load_library().mrs_tg_default_options(C.byref(opt))
    for k, v in overrides.items():
        if not hasattr(opt, k):
            raise TypeError("unknown option %r" % k)
        setattr(opt, k, v)
    return opt


def _np_ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


class _PinnedOwner:
    """frees a block of mrs_tg_host_alloc when the last reference to it is gone"""

    def __init__(self, nbytes):
        self._L = load_library()
        self.ptr = C.c_void_p()
        rc = self._L.mrs_tg_host_alloc(C.c_size_t(nbytes), C.byref(self.ptr))
        if rc:
            raise MrsTgError("mrs_tg_host_alloc failed (%d): %s" % (rc, self._L.mrs_tg_last_error(None).decode()))

    def __del__(self):
        if getattr(self, "ptr", None) is not None and self.ptr.value:
            self._L.mrs_tg_host_free(self.ptr)
            self.ptr = C.c_void_p()


def pinned_empty(shape, dtype=np.float64):
    """A numpy array in pinned host memory (mrs_tg_host_alloc): the GPU's DMA engines read / write it directly, so
    mrs_tg_solve_batch neither stages nor pins it per call.  The block lives as long as ANY array that views it: numpy
    collapses chains of views onto the buffer object they all come from (here the ctypes array), so the owner of the block
    hangs on that object -- not on the first array handed out, which a slice or a ravel() of it does not keep alive."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
    owner = _PinnedOwner(n * dtype.itemsize)
    buf = (C.c_char * max(n * dtype.itemsize, 1)).from_address(owner.ptr.value)
    buf._owner = owner   # buffer object -> owner; every view -> buffer object
    return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)


def pinned_copy(a):
    out = pinned_empty(a.shape, a.dtype)
    out[...] = a
    return out


def _t_ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class Context:
    """One HIP device + stream (mrs_tg_ctx)."""

    def __init__(self, device=0):
        self._L = load_library()
        h = C.c_void_p()
        rc = self._L.mrs_tg_create(int(device), C.byref(h))
        if rc != 0:
            raise MrsTgError("mrs_tg_create failed (%d): %s" % (rc, self._L.mrs_tg_last_error(None).decode()))
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._L.mrs_tg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise MrsTgError("%s failed (%d): %s" % (what, rc, self._L.mrs_tg_last_error(self._h).decode()))

    def use_torch_stream(self):
        """Launch on torch's current HIP stream so torch events / allocator ordering apply."""
        import torch
        self._check(self._L.mrs_tg_set_stream(self._h, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "set_stream")

    def synchronize(self):
        self._check(self._L.mrs_tg_synchronize(self._h), "synchronize")

    def set_profiling(self, enabled):
        self._check(self._L.mrs_tg_set_profiling(self._h, int(bool(enabled))), "set_profiling")

    def last_kernel_ms(self, kernel_id):
        ms = C.c_float(0)
        self._check(self._L.mrs_tg_last_kernel_ms(self._h, int(kernel_id), C.byref(ms)), "last_kernel_ms")
        return ms.value

    def kernel_ms_history(self, kernel_id, capacity=512):
        """Per-dispatch durations (ms) of the newest timed launches since set_profiling(True), oldest first."""
        buf = (C.c_float * capacity)()
        n = self._L.mrs_tg_kernel_ms_history(self._h, int(kernel_id), buf, capacity)
        if n < 0:
            self._check(n, "kernel_ms_history")
        return [buf[i] for i in range(n)]

    def solve_batch(self, batch: Batch, seg_times=None, out=None, **opts):
        """Host arrays in, host arrays out (mrs_tg_solve_batch).  seg_times None => estimate_times.
        out: the dict of a previous call with the same shapes (or arrays from pinned_empty): its arrays are written in place
        instead of allocating new ones -- what a server that calls in a loop does; `times` is then used as given AND
        overwritten unless seg_times is passed."""
        opt = default_options(derivative_to_optimize=batch.derivative_to_optimize, **opts)
        nS, P = batch.n_segments, batch.n_paths
        sampling = opt.sampling_dt > 0
        if out is not None:
            t, coeffs, status, cost = out["times"], out["coeffs"], out["status"], out["cost"]
            n_samples, samples = (out["n_samples"], out["samples"]) if sampling else (None, None)
            if seg_times is None:
                opt.estimate_times = 1
            else:
                t[:] = seg_times
            assert t.size == nS and coeffs.shape == (nS, N_DIM, N_COEFF) and status.size == P
        else:
            if seg_times is None:
                opt.estimate_times = 1
                t = np.zeros(nS)
            else:
                t = np.ascontiguousarray(seg_times, dtype=np.float64).copy()
                assert t.size == nS
            coeffs = np.zeros((nS, N_DIM, N_COEFF))
            status = np.zeros(P, dtype=np.int32)
            cost = np.zeros(P)
            n_samples = np.zeros(P, dtype=np.int32) if sampling else None
            samples = np.zeros((P, max(opt.sample_capacity, 1), N_DIM)) if sampling else None
        rc = self._L.mrs_tg_solve_batch(self._h, P, _np_ptr(batch.seg_offsets), _np_ptr(batch.waypoints),
                                        _np_ptr(batch.fixed_mask), _np_ptr(batch.fixed_values), _np_ptr(batch.limits),
                                        C.byref(opt), _np_ptr(t), _np_ptr(coeffs), _np_ptr(status), _np_ptr(cost),
                                        _np_ptr(n_samples), _np_ptr(samples))
        self._check(rc, "mrs_tg_solve_batch")
        return dict(times=t, coeffs=coeffs, status=status, cost=cost, n_samples=n_samples, samples=samples)

    def bind_find_trajectory(self, waypoints, stop_at=None, initial_state=None, limits=None, relax_heading=False,
                             sample_capacity=4096, **opts):
        """The arguments of mrs_tg_find_trajectory marshalled ONCE: returns (call, result) -- call() is the foreign call alone
        (what a C++ host pays per request; the ctypes structures and output arrays are reused), result() reads its outputs."""
        from .problem import DEFAULT_LIMITS
        wp = np.asarray(waypoints, dtype=np.float64).reshape(-1, 4)
        n = wp.shape[0]
        arr = (Waypoint * n)()
        for i in range(n):
            for k in range(4):
                arr[i].coords[k] = wp[i, k]
            arr[i].stop_at = int(bool(stop_at[i])) if stop_at is not None else 0
        init = None
        if initial_state is not None:
            init = InitialState()
            init.heading = float(initial_state["heading"])
            for k in range(4):
                init.velocity[k] = float(initial_state["velocity"][k])
                init.acceleration[k] = float(initial_state["acceleration"][k])
                init.jerk[k] = float(initial_state["jerk"][k])
        lim = np.ascontiguousarray(DEFAULT_LIMITS if limits is None else limits, dtype=np.float64)
        opts.setdefault("time_alloc_method", TIME_ALLOC_MELLINGER)
        opts.setdefault("sampling_dt", 0.2)
        opt = default_options(sample_capacity=sample_capacity, **opts)
        S = n - 1
        times = np.zeros(S)
        coeffs = np.zeros((S, N_DIM, N_COEFF))
        status = C.c_int32(0)
        ns = C.c_int32(0)
        samples = np.zeros((sample_capacity, N_DIM))
        fn, h, check = self._L.mrs_tg_find_trajectory, self._h, self._check
        a_init = C.byref(init) if init is not None else None
        a_lim, a_opt, a_relax, a_t, a_c = _np_ptr(lim), C.byref(opt), int(bool(relax_heading)), _np_ptr(times), _np_ptr(coeffs)
        a_st, a_ns, a_smp = C.cast(C.byref(status), C.c_void_p), C.cast(C.byref(ns), C.c_void_p), _np_ptr(samples)
        keep = (arr, init, lim, opt, times, coeffs, status, ns, samples)

        def call(_keep=keep):
            rc = fn(h, arr, n, a_init, a_lim, a_opt, a_relax, a_t, a_c, a_st, a_ns, a_smp)
            if rc:
                check(rc, "mrs_tg_find_trajectory")

        L = self._L

        def result():
            rej, baca = C.c_int32(0), C.c_double(0.0)
            L.mrs_tg_find_trajectory_info(h, C.byref(rej), C.byref(baca))
            return dict(times=times.copy(), coeffs=coeffs.copy(), status=status.value, n_samples=ns.value,
                        samples=samples[:min(ns.value, sample_capacity)].copy(), rejection=rej.value,
                        baca_total_time=baca.value,
                        message=(L.mrs_tg_last_error(h).decode() if rej.value else ""))
        return call, result

    def find_trajectory(self, waypoints, stop_at=None, initial_state=None, limits=None, relax_heading=False,
                        sample_capacity=4096, **opts):
        """findTrajectory() for one path (mrs_tg_find_trajectory)."""
        call, result = self.bind_find_trajectory(waypoints, stop_at, initial_state, limits, relax_heading, sample_capacity, **opts)
        call()
        return result()


class MultiContext:
    """Several devices (mrs_tg_multi): a batch is sharded over them, one host thread per device."""

    def __init__(self, devices):
        self._L = load_library()
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        rc = self._L.mrs_tg_create_multi(arr, len(devices), C.byref(h))
        if rc != 0:
            raise MrsTgError("mrs_tg_create_multi failed (%d): %s" % (rc, self._L.mrs_tg_last_error(None).decode()))
        self._h = h
        self.n_devices = self._L.mrs_tg_multi_n_devices(h)

    def close(self):
        if getattr(self, "_h", None):
            self._L.mrs_tg_destroy_multi(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shard(self, seg_offsets):
        so = np.ascontiguousarray(seg_offsets, dtype=np.int32)
        out = np.zeros(so.size - 1, dtype=np.int32)
        rc = self._L.mrs_tg_multi_shard(self._h, so.size - 1, _np_ptr(so), _np_ptr(out))
        if rc != 0:
            raise MrsTgError("mrs_tg_multi_shard failed (%d)" % rc)
        return out

    def solve_batch(self, batch: Batch, seg_times=None, **opts):
        """Context.solve_batch over all devices (mrs_tg_multi_solve_batch)."""
        opt = default_options(derivative_to_optimize=batch.derivative_to_optimize, **opts)
        nS, P = batch.n_segments, batch.n_paths
        if seg_times is None:
            opt.estimate_times = 1
            t = np.zeros(nS)
        else:
            t = np.ascontiguousarray(seg_times, dtype=np.float64).copy()
        coeffs = np.zeros((nS, N_DIM, N_COEFF))
        status = np.zeros(P, dtype=np.int32)
        cost = np.zeros(P)
        sampling = opt.sampling_dt > 0
        n_samples = np.zeros(P, dtype=np.int32) if sampling else None
        samples = np.zeros((P, max(opt.sample_capacity, 1), N_DIM)) if sampling else None
        rc = self._L.mrs_tg_multi_solve_batch(self._h, P, _np_ptr(batch.seg_offsets), _np_ptr(batch.waypoints),
                                              _np_ptr(batch.fixed_mask), _np_ptr(batch.fixed_values), _np_ptr(batch.limits),
                                              C.byref(opt), _np_ptr(t), _np_ptr(coeffs), _np_ptr(status), _np_ptr(cost),
                                              _np_ptr(n_samples), _np_ptr(samples))
        if rc != 0:
            raise MrsTgError("mrs_tg_multi_solve_batch failed (%d): %s" % (rc, self._L.mrs_tg_multi_last_error(self._h).decode()))
        return dict(times=t, coeffs=coeffs, status=status, cost=cost, n_samples=n_samples, samples=samples)


_WAYPOINT_DTYPE = np.dtype([("coords", "<f8", (4,)), ("stop_at", "u1"), ("pad", "u1", (7,))])   # = mrs_tg_waypoint (40 bytes)


def _waypoint_array(paths, stop_flags=None):
    """the requests' waypoints as one mrs_tg_waypoint array + CSR offsets (numpy: a Python loop over 50 000 waypoints of 4096
    requests was a third of the wrapper's time)"""
    assert _WAYPOINT_DTYPE.itemsize == C.sizeof(Waypoint)
    pts = [np.asarray(p, dtype=np.float64).reshape(-1, 4) for p in paths]
    counts = np.array([q.shape[0] for q in pts], dtype=np.int64)
    off = np.zeros(len(pts) + 1, dtype=np.int32)
    off[1:] = np.cumsum(counts)
    arr = np.zeros(int(off[-1]), dtype=_WAYPOINT_DTYPE)
    if pts:
        arr["coords"] = np.concatenate(pts, axis=0) if len(pts) > 1 else pts[0]
    if stop_flags is not None:
        for pi, f in enumerate(stop_flags):
            if f is not None:
                arr["stop_at"][off[pi]:off[pi + 1]] = np.asarray(f, dtype=bool)
    return arr, off


def waypoint_trajectory_idxs(samples, waypoints):
    """getWaypointInTrajectoryIdxs (mrs_trajectory_generation.cpp:1461-1499) for one path: the sample indices at which the
    trajectory passes its waypoints (mrs_tg_waypoint_trajectory_idxs; host arithmetic, no device work)"""
    wp = np.ascontiguousarray(waypoints, dtype=np.float64)
    arr, _ = _waypoint_array([wp])
    smp = np.ascontiguousarray(samples, dtype=np.float64)
    idx = np.zeros(wp.shape[0] + 4, dtype=np.int32)
    k = load_library().mrs_tg_waypoint_trajectory_idxs(_np_ptr(smp), smp.shape[0], arr.ctypes.data_as(C.POINTER(Waypoint)),
                                                       wp.shape[0], _np_ptr(idx))
    return idx[:k].copy()


def optimize_paths(ctx, paths, limits=None, stop_flags=None, initial_states=None, relax_heading=None, policy=None,
                   sample_capacity=4096, out=None):
    """mrs_tg_optimize_paths: the reference's optimize() policy loop for a list of waypoint paths
    (each [n][4] array; its first row is the initial condition when initial_states[p] is given).
    out: the dict of a previous call with the same number of requests and capacity -- its arrays are written in place (a
    server that answers batch after batch keeps its response arrays: a fresh [P][capacity][4] array is 67 MB of page faults
    per 1024 requests at capacity 2048)."""
    from .problem import DEFAULT_LIMITS
    P = len(paths)
    arr, off = _waypoint_array(paths, stop_flags)
    lim = np.ascontiguousarray(np.tile(DEFAULT_LIMITS, (P, 1)) if limits is None else limits, dtype=np.float64).reshape(P, 9)
    inits = (InitialState * max(P, 1))()
    has = np.zeros(max(P, 1), dtype=np.uint8)
    if initial_states is not None:
        for p, st in enumerate(initial_states):
            if st is None:
                continue
            has[p] = 1
            inits[p].heading = float(st["heading"])
            for k in range(4):
                inits[p].velocity[k] = float(st["velocity"][k])
                inits[p].acceleration[k] = float(st["acceleration"][k])
                inits[p].jerk[k] = float(st["jerk"][k])
    relax = np.ascontiguousarray(relax_heading if relax_heading is not None else np.zeros(max(P, 1)), dtype=np.uint8)
    pol = policy or default_policy_options()
    if out is not None:
        success, ns, samples, maxdev, nwp, iters = (out[k] for k in ("success", "n_samples", "samples", "max_deviation",
                                                                      "n_waypoints", "iterations"))
        assert samples.shape == (P, sample_capacity, 4) and success.size == P
    else:
        success = np.zeros(P, dtype=np.int32)
        ns = np.zeros(P, dtype=np.int32)
        samples = np.zeros((P, sample_capacity, 4))
        maxdev = np.zeros(P)
        nwp = np.zeros(P, dtype=np.int32)
        iters = np.zeros(P, dtype=np.int32)
    rc = ctx._L.mrs_tg_optimize_paths(ctx._h, P, _np_ptr(off), arr.ctypes.data_as(C.POINTER(Waypoint)), inits, _np_ptr(has), _np_ptr(lim), _np_ptr(relax),
                                      C.byref(pol), int(sample_capacity), _np_ptr(success), _np_ptr(ns), _np_ptr(samples),
                                      _np_ptr(maxdev), _np_ptr(nwp), _np_ptr(iters))
    ctx._check(rc, "mrs_tg_optimize_paths")
    return dict(success=success, n_samples=ns, samples=samples, max_deviation=maxdev, n_waypoints=nwp, iterations=iters)


class Plan:
    """Batch structure analysed once; device-resident (torch) operands afterwards (mrs_tg_plan)."""

    def __init__(self, ctx: Context, seg_offsets):
        self.ctx = ctx
        self._L = ctx._L
        so = np.ascontiguousarray(seg_offsets, dtype=np.int32)
        h = C.c_void_p()
        ctx._check(self._L.mrs_tg_plan_create(ctx._h, so.size - 1, _np_ptr(so), C.byref(h)), "mrs_tg_plan_create")
        self._h = h
        self.seg_offsets = so
        self.n_paths = self._L.mrs_tg_plan_n_paths(h)
        self.n_segments = self._L.mrs_tg_plan_n_segments(h)
        self.max_segments = self._L.mrs_tg_plan_max_segments(h)
        order = np.zeros(max(self.n_paths, 1), dtype=np.int32)
        ctx._check(self._L.mrs_tg_plan_get_order(h, _np_ptr(order)), "mrs_tg_plan_get_order")
        self.order = order[:self.n_paths]
        self._bound = []

    def explain(self, opt, group_size=0):
        """The kernels a solve of this plan under `opt` WOULD launch, in order (mrs_tg_plan_explain: the launchers run dry);
        group_size 1 .. 16: one dispatch of the grouped issue carrying that many batches."""
        buf = (C.c_char_p * 32)()
        n = self._L.mrs_tg_plan_explain(self._h, C.byref(opt), int(group_size), buf, 32)
        if n < 0:
            self.ctx._check(n, "mrs_tg_plan_explain")
        return [buf[i].decode() for i in range(n)]

    def close(self):
        if getattr(self, "_h", None):
            for b in self._bound:
                self._L.mrs_tg_bound_solve_destroy(b)
            self._bound = []
            self._L.mrs_tg_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def block_doubles(self):
        return self._L.mrs_tg_plan_block_bytes(self._h) // 8

    def assemble(self, derivative, seg_times_dev, H_dev, Ainv_dev):
        self.ctx._check(self._L.mrs_tg_plan_assemble(self._h, int(derivative), _t_ptr(seg_times_dev), _t_ptr(H_dev),
                                                     _t_ptr(Ainv_dev)), "mrs_tg_plan_assemble")

    def blocks_to_segments(self, blocks):
        """Slot-major SoA block buffer (torch or numpy) -> numpy [sum S][10][10] in CSR segment order."""
        arr = blocks.detach().cpu().numpy() if hasattr(blocks, "detach") else np.asarray(blocks)
        P = self.n_paths
        soa = arr.reshape(self.max_segments, 100, P)
        out = np.zeros((self.n_segments, 10, 10))
        for q, p in enumerate(self.order):
            s0, s1 = int(self.seg_offsets[p]), int(self.seg_offsets[p + 1])
            out[s0:s1] = soa[:s1 - s0, :, q].reshape(s1 - s0, 10, 10)
        return out

    def solve(self, opt, fixed_mask, fixed_values, seg_times, coeffs, status, cost=None, waypoints=None, limits=None,
              n_samples=None, samples=None):
        self.ctx._check(self._L.mrs_tg_plan_solve(self._h, _t_ptr(waypoints), _t_ptr(fixed_mask), _t_ptr(fixed_values),
                                                  _t_ptr(limits), C.byref(opt), _t_ptr(seg_times), _t_ptr(coeffs),
                                                  _t_ptr(status), _t_ptr(cost), _t_ptr(n_samples), _t_ptr(samples)),
                        "mrs_tg_plan_solve")

    def bind_solve(self, opt, fixed_mask, fixed_values, seg_times, coeffs, status, cost=None, waypoints=None, limits=None,
                   n_samples=None, samples=None):
        """solve() with its arguments fixed once (mrs_tg_plan_bind_solve): returns a callable that enqueues the same solve
        again on the context's stream for one foreign call with one pointer -- a server that keeps several batches in flight
        on several streams issues steps faster than the GPU finishes them only if the per-call host cost is small.
        The tensors must stay alive and in place for as long as the callable is used."""
        keep = (opt, fixed_mask, fixed_values, seg_times, coeffs, status, cost, waypoints, limits, n_samples, samples)
        h = C.c_void_p()
        self.ctx._check(self._L.mrs_tg_plan_bind_solve(self._h, _t_ptr(waypoints), _t_ptr(fixed_mask), _t_ptr(fixed_values),
                                                       _t_ptr(limits), C.byref(opt), _t_ptr(seg_times), _t_ptr(coeffs),
                                                       _t_ptr(status), _t_ptr(cost), _t_ptr(n_samples), _t_ptr(samples),
                                                       C.byref(h)), "mrs_tg_plan_bind_solve")
        self._bound.append(h)
        fn, check = self._L.mrs_tg_bound_solve_launch, self.ctx._check

        def enqueue(_keep=keep, _h=h):
            rc = fn(_h)
            if rc:
                check(rc, "mrs_tg_bound_solve_launch")
        enqueue.handle = h
        enqueue.ctx = self.ctx
        return enqueue

    def cost_gradient(self, derivative, fixed_mask, fixed_values, seg_times, cost, grad):
        self.ctx._check(self._L.mrs_tg_plan_cost_gradient(self._h, int(derivative), _t_ptr(fixed_mask),
                                                          _t_ptr(fixed_values), _t_ptr(seg_times), _t_ptr(cost),
                                                          _t_ptr(grad)), "mrs_tg_plan_cost_gradient")

    def segment_maxima(self, coeffs, seg_times, maxima):
        self.ctx._check(self._L.mrs_tg_plan_segment_maxima(self._h, _t_ptr(coeffs), _t_ptr(seg_times), _t_ptr(maxima)),
                        "mrs_tg_plan_segment_maxima")

    def careful_count(self):
        n = C.c_int32(0)
        self.ctx._check(self._L.mrs_tg_plan_careful_count(self._h, C.byref(n)), "mrs_tg_plan_careful_count")
        return n.value

    def sample_states(self, coeffs, seg_times, sampling_dt, sample_capacity, n_samples, states):
        """sampleWholeTrajectory with all fields: states [n_paths][capacity][STATE_ORDERS][4] (device tensors)."""
        self.ctx._check(self._L.mrs_tg_plan_sample_states(self._h, _t_ptr(coeffs), _t_ptr(seg_times), float(sampling_dt),
                                                          int(sample_capacity), _t_ptr(n_samples), _t_ptr(states)),
                        "mrs_tg_plan_sample_states")


class RoundRobin:
    """The issue loop of a host that keeps several batches in flight, in C (mrs_tg_bound_solve_launch_many): launch k goes to
    calls[k % len(calls)], each a callable returned by Plan.bind_solve (one per context + stream)."""

    def __init__(self, calls, threads=1, grouped=False):
        """grouped: mrs_tg_bound_solve_launch_group -- the launches of a round go out as ONE dispatch (bound solves of one
        plan, fixed times, default solve)."""
        self._calls = list(calls)
        self._arr = (C.c_void_p * len(self._calls))(*[c.handle for c in self._calls])
        self._fn = load_library().mrs_tg_bound_solve_launch_many_mt
        self._group = load_library().mrs_tg_bound_solve_launch_group if grouped else None
        self._threads = int(threads)

    def __call__(self, n_launches):
        if self._group is not None:
            rc = self._group(self._arr, len(self._calls), int(n_launches))
        else:
            rc = self._fn(self._arr, len(self._calls), int(n_launches), self._threads)
        if rc:
            for c in self._calls:
                c.ctx._check(rc, "mrs_tg_bound_solve_launch_group" if self._group is not None else "mrs_tg_bound_solve_launch_many")


class DeviceBatch:
    """A Batch uploaded to HBM as torch tensors, plus output tensors, for the plan interface."""


    def __init__(self, batch: Batch, device="cuda:0", sample_capacity=0):
        import torch
        dev = torch.device(device)
        self.batch = batch
        self.waypoints = torch.from_numpy(batch.waypoints).to(dev)
        self.fixed_mask = torch.from_numpy(batch.fixed_mask).to(dev)
        self.fixed_values = torch.from_numpy(batch.fixed_values).to(dev)
        self.limits = torch.from_numpy(batch.limits).to(dev)
        nS, P = batch.n_segments, batch.n_paths
        self.seg_times = torch.zeros(nS, dtype=torch.float64, device=dev)
        self.coeffs = torch.zeros((nS, N_DIM, N_COEFF), dtype=torch.float64, device=dev)
        self.status = torch.zeros(P, dtype=torch.int32, device=dev)
        self.cost = torch.zeros(P, dtype=torch.float64, device=dev)
        self.n_samples = torch.zeros(P, dtype=torch.int32, device=dev)
        self.samples = torch.zeros((P, max(sample_capacity, 1), N_DIM), dtype=torch.float64, device=dev)
