// mrs_tg_abi.hip -- implementation of the C ABI declared in include/mrs_tg.h.
// Context / plan bookkeeping, workspace management, host<->device marshalling and the launch
// sequences.  No numerics live here and nothing falls back to the CPU: every failure is reported.
#include <hip/hip_runtime.h>

#include "mrs_tg_pool.h"

#include <algorithm>
#include <thread>
#include <memory>
#include <condition_variable>
#include <atomic>
#include <chrono>
#include <cfloat>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/mrs_tg.h"
#include "mrs_tg_launch.h"
#include "mrs_tg_nonlinear.h"
#include "mrs_tg_policy_host.hpp"

namespace {

std::mutex g_err_mutex;
std::string g_last_error;

void set_global_error(const std::string& s) {
  std::lock_guard<std::mutex> lk(g_err_mutex);
  g_last_error = s;
}

}  // namespace

static std::atomic<int> g_live_contexts{0};

struct mrs_tg_ctx {
  int device = -1;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string last_error;
  bool profiling = false;
  // per-dispatch timing: a ring of event pairs per kernel family, one pair per timed launch (kTimerRing launches can be
  // queued before the oldest is overwritten)
  static constexpr int kTimerRing = 512;
  std::vector<hipEvent_t> ev_start[3], ev_stop[3];
  long long ev_count[3] = {0, 0, 0};   // timed launches since profiling was switched on
  hipDeviceProp_t prop;
  double wall_clock_hz = 1.0e8;  // rate of s_memrealtime (hipDeviceAttributeWallClockRate)
  // plan of the most recent mrs_tg_solve_batch: a caller that sends the same batch shape again (the nodelet's
  // deviation loop, a server's fixed batch size) skips the analysis, the uploads and the device allocations
  mrs_tg_plan* cached_plan = nullptr;
  // mrs_tg_solve_batch's transfer arenas, kept across calls: one device block holding every input and output array of a
  // call, and a pinned host block (hipHostMalloc) through which the small host arrays travel packed, one copy each way
  void* d_arena = nullptr;
  size_t d_arena_bytes = 0;
  void* h_arena = nullptr;
  size_t h_arena_bytes = 0;
  // pinned host scratch of mrs_tg_optimize_paths (the arrays of its rounds), kept across calls
  void* h_scratch = nullptr;
  size_t h_scratch_bytes = 0;
  // what the most recent mrs_tg_find_trajectory decided (mrs_tg_find_trajectory_info)
  int32_t find_rejection = 0;
  double find_baca_total = 0.0;
};

struct mrs_tg_plan {
  mrs_tg_ctx* ctx = nullptr;
  mrs_tg::BatchView view{};
  std::vector<int32_t> order_host;
  std::vector<int32_t> seg_offsets_host;
  int32_t* d_seg_offsets = nullptr;
  int32_t* d_order = nullptr;
  int32_t* d_slot_start = nullptr;
  // lazily allocated scratch
  double* d_ws = nullptr;
  size_t ws_doubles = 0;
  double* d_H = nullptr;
  double* d_Ainv = nullptr;
  mrs_tg::NonlinearPlan nl;
};

namespace {

int fail(mrs_tg_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->last_error = buf;
  set_global_error(buf);
  return code;
}

long long budget_ticks(const mrs_tg_ctx* ctx, double seconds_left) {
  // a budget that is already spent still lets every path evaluate its start point once (nlopt checks after an evaluation)
  const double t = seconds_left * ctx->wall_clock_hz;
  return t < 1.0 ? 1ll : (t > 9.0e18 ? (long long)9.0e18 : (long long)t);
}

#define HIP_TRY(ctx, expr)                                                                        \
  do {                                                                                            \
    hipError_t e__ = (expr);                                                                      \
    if (e__ != hipSuccess) return fail((ctx), MRS_TG_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
  } while (0)

// Arms the per-dispatch timer for the next timed launch of this thread (the main kernel of the family `kernel_id`):
// the events ride on the kernel launch itself, so their difference is that dispatch's own duration.
struct ProfileScope {
  mrs_tg_ctx* ctx;
  int id;
  bool armed = false;
  ProfileScope(mrs_tg_ctx* c, int kernel_id) : ctx(c), id(kernel_id) {
    if (!ctx->profiling) return;
    if (ctx->ev_start[id].empty()) {
      ctx->ev_start[id].assign(mrs_tg_ctx::kTimerRing, nullptr);
      ctx->ev_stop[id].assign(mrs_tg_ctx::kTimerRing, nullptr);
    }
    const int slot = (int)(ctx->ev_count[id] % mrs_tg_ctx::kTimerRing);
    if (!ctx->ev_start[id][slot]) {
      if (hipEventCreate(&ctx->ev_start[id][slot]) != hipSuccess || hipEventCreate(&ctx->ev_stop[id][slot]) != hipSuccess) return;
    }
    mrs_tg::set_kernel_timer(ctx->ev_start[id][slot], ctx->ev_stop[id][slot]);
    armed = true;
  }
  ~ProfileScope() {
    if (!armed) return;
    const mrs_tg::KernelTimer left = mrs_tg::take_kernel_timer();  // consumed <=> a timed kernel was launched
    if (left.start == nullptr) ++ctx->ev_count[id];
  }
};

// hipSetDevice on every call costs more than asking which device is current
hipError_t use_device(int device) {
  int current = -1;
  if (hipGetDevice(&current) == hipSuccess && current == device) return hipSuccess;  // (another library on this thread may
  return hipSetDevice(device);                                                        // have switched: ask, do not remember)
}

int ensure_ws(mrs_tg_plan* plan, size_t doubles) {
  if (plan->ws_doubles >= doubles) return MRS_TG_OK;
  if (plan->d_ws) {
    (void)hipStreamSynchronize(plan->ctx->stream);  // pool contract: no work in flight on a block that is given back
    mrs_tg::pool_free(plan->d_ws);
  }
  plan->d_ws = nullptr;
  plan->ws_doubles = 0;
  HIP_TRY(plan->ctx, mrs_tg::pool_alloc(&plan->d_ws, doubles * sizeof(double)));
  plan->ws_doubles = doubles;
  return MRS_TG_OK;
}

int ensure_blocks(mrs_tg_plan* plan) {
  if (plan->d_H) return MRS_TG_OK;
  const size_t bytes = mrs_tg_plan_block_bytes(plan);
  HIP_TRY(plan->ctx, mrs_tg::pool_alloc(&plan->d_H, bytes));
  HIP_TRY(plan->ctx, mrs_tg::pool_alloc(&plan->d_Ainv, bytes));
  return MRS_TG_OK;
}

int check_options(mrs_tg_ctx* ctx, const mrs_tg_options* opt) {
  if (!opt) return fail(ctx, MRS_TG_ERR_INVALID_ARG, "options is NULL");
  if (opt->derivative_to_optimize < 0 || opt->derivative_to_optimize > 4)
    return fail(ctx, MRS_TG_ERR_INVALID_ARG, "derivative_to_optimize %d outside [0, 4]", opt->derivative_to_optimize);
  if (opt->time_alloc_method < MRS_TG_TIME_ALLOC_NONE || opt->time_alloc_method > MRS_TG_TIME_ALLOC_RICHTER_TIME_AND_CONSTRAINTS)
    return fail(ctx, MRS_TG_ERR_INVALID_ARG,
                "time_alloc_method %d is not one of -1 (fixed times), 0 / 1 (time only, gradient-free), 2 (Mellinger), "
                "3 / 4 (time and free constraints, gradient-free)",
                opt->time_alloc_method);
  if (opt->sampling_dt > 0 && opt->sample_capacity < 0)
    return fail(ctx, MRS_TG_ERR_INVALID_ARG, "negative sample_capacity");
  return MRS_TG_OK;
}

}  // namespace

namespace mrs_tg {
int report_error(mrs_tg_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return fail(ctx, code, "%s", buf);
}
}  // namespace mrs_tg

namespace {
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)mrs_tg::pool_free(p);
  }
  hipError_t alloc(size_t bytes) { return mrs_tg::pool_alloc(&p, bytes ? bytes : 8); }
  template <class T>
  T* as() {
    return static_cast<T*>(p);
  }
};
}  // namespace


extern "C" {

int mrs_tg_abi_version(void) { return MRS_TG_ABI_VERSION; }

int mrs_tg_capabilities(void) { return mrs_tg::careful_rerun_built() ? MRS_TG_CAP_CAREFUL_COST : 0; }

void mrs_tg_kernel_trace_reset(void) { mrs_tg::kernel_trace_reset(); }

int mrs_tg_kernel_trace(const char** names_out, int capacity) {
  if (!names_out || capacity <= 0) return 0;
  return mrs_tg::kernel_trace(names_out, capacity);
}

void mrs_tg_default_options(mrs_tg_options* opt) {
  if (!opt) return;
  mrs_tg::policy::default_solver_options(opt);
}

const char* mrs_tg_last_error(const mrs_tg_ctx* ctx) {
  if (ctx) return ctx->last_error.c_str();
  static thread_local std::string copy;
  std::lock_guard<std::mutex> lk(g_err_mutex);
  copy = g_last_error;
  return copy.c_str();
}

int mrs_tg_create(int device_ordinal, mrs_tg_ctx** ctx_out) {
  if (!ctx_out) return fail(nullptr, MRS_TG_ERR_INVALID_ARG, "ctx_out is NULL");
  *ctx_out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, MRS_TG_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  if (device_ordinal < 0 || device_ordinal >= n)
    return fail(nullptr, MRS_TG_ERR_INVALID_ARG, "device ordinal %d outside [0, %d)", device_ordinal, n);
  mrs_tg_ctx* ctx = new (std::nothrow) mrs_tg_ctx();
  if (!ctx) return fail(nullptr, MRS_TG_ERR_NOMEM, "out of host memory");
  ctx->device = device_ordinal;
  if ((e = use_device(device_ordinal)) != hipSuccess || (e = hipGetDeviceProperties(&ctx->prop, device_ordinal)) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) {
    delete ctx;
    return fail(nullptr, MRS_TG_ERR_HIP, "device setup failed: %s", hipGetErrorString(e));
  }
  ctx->stream = ctx->own_stream;
  {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device_ordinal) == hipSuccess && khz > 0)
      ctx->wall_clock_hz = 1.0e3 * (double)khz;
  }
  ++g_live_contexts;
  *ctx_out = ctx;
  return MRS_TG_OK;
}

void mrs_tg_destroy(mrs_tg_ctx* ctx) {
  if (!ctx) return;
  (void)use_device(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->cached_plan) {
    mrs_tg_plan* p = ctx->cached_plan;
    ctx->cached_plan = nullptr;
    mrs_tg_plan_destroy(p);
  }
  for (int i = 0; i < 3; ++i) {
    for (hipEvent_t e : ctx->ev_start[i])
      if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_stop[i])
      if (e) (void)hipEventDestroy(e);
  }
  if (ctx->d_arena) (void)mrs_tg::pool_free(ctx->d_arena);
  if (ctx->h_arena) (void)hipHostFree(ctx->h_arena);
  if (ctx->h_scratch) (void)hipHostFree(ctx->h_scratch);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
  if (--g_live_contexts == 0) {  // the sampling tables and the cached device blocks go with the last context
    mrs_tg::sample_tables_release();
    mrs_tg::pool_release_cached();
  }
}

int mrs_tg_set_stream(mrs_tg_ctx* ctx, void* hip_stream) {
  if (!ctx) return fail(nullptr, MRS_TG_ERR_INVALID_ARG, "ctx is NULL");
  ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
  return MRS_TG_OK;
}

int mrs_tg_reset_stream(mrs_tg_ctx* ctx) {
  if (!ctx) return fail(nullptr, MRS_TG_ERR_INVALID_ARG, "ctx is NULL");
  ctx->stream = ctx->own_stream;
  return MRS_TG_OK;
}

int mrs_tg_synchronize(mrs_tg_ctx* ctx) {
  if (!ctx) return fail(nullptr, MRS_TG_ERR_INVALID_ARG, "ctx is NULL");
  HIP_TRY(ctx, use_device(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MRS_TG_OK;
}

int mrs_tg_set_profiling(mrs_tg_ctx* ctx, int enabled) {
  if (!ctx) return fail(nullptr, MRS_TG_ERR_INVALID_ARG, "ctx is NULL");
  ctx->profiling = enabled != 0;
  if (ctx->profiling)
    for (int i = 0; i < 3; ++i) ctx->ev_count[i] = 0;  // a new series
  return MRS_TG_OK;
}

int mrs_tg_kernel_ms_history(mrs_tg_ctx* ctx, int kernel_id, float* ms_out, int capacity) {
  if (!ctx || (capacity > 0 && !ms_out)) return fail(ctx, MRS_TG_ERR_INVALID_ARG, "NULL argument");
  if (kernel_id < 0 || kernel_id > 2) return fail(ctx, MRS_TG_ERR_INVALID_ARG, "kernel id %d outside [0, 2]", kernel_id);
  const long long total = ctx->ev_count[kernel_id];
  long long n = total < mrs_tg_ctx::kTimerRing ? total : mrs_tg_ctx::kTimerRing;
  if (n > capacity) n = capacity;
  HIP_TRY(ctx, use_device(ctx->device));
  for (long long i = 0; i < n; ++i) {  // the newest n launches, oldest first
    const int slot = (int)((total - n + i) % mrs_tg_ctx::kTimerRing);
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_stop[kernel_id][slot]));
    HIP_TRY(ctx, hipEventElapsedTime(&ms_out[i], ctx->ev_start[kernel_id][slot], ctx->ev_stop[kernel_id][slot]));
  }
  return (int)n;
}

int mrs_tg_last_kernel_ms(mrs_tg_ctx* ctx, int kernel_id, float* ms_out) {
  if (!ctx || !ms_out) return fail(ctx, MRS_TG_ERR_INVALID_ARG, "NULL argument");
  if (kernel_id < 0 || kernel_id > 2 || ctx->ev_count[kernel_id] == 0)
    return fail(ctx, MRS_TG_ERR_INVALID_ARG, "no timed launch recorded for kernel id %d (mrs_tg_set_profiling, then a call that runs that kernel)", kernel_id);
  const int n = mrs_tg_kernel_ms_history(ctx, kernel_id, ms_out, 1);
  return n == 1 ? MRS_TG_OK : n;
}

// ---- plan ------------------------------------------------------------------------------------

int mrs_tg_plan_create(mrs_tg_ctx* ctx, int32_t n_paths, const int32_t* so, mrs_tg_plan** plan_out) {
  if (!ctx || !plan_out || !so) return fail(ctx, MRS_TG_ERR_INVALID_ARG, "NULL argument");
  *plan_out = nullptr;
  if (n_paths < 0) return fail(ctx, MRS_TG_ERR_INVALID_ARG, "negative n_paths");
  if (so[0] != 0) return fail(ctx, MRS_TG_ERR_INVALID_ARG, "seg_offsets[0] must be 0");
  int max_S = 0, min_S = n_paths ? INT32_MAX : 0;
  for (int p = 0; p < n_paths; ++p) {
    const int S = so[p + 1] - so[p];
    if (S < 1) return fail(ctx, MRS_TG_ERR_INVALID_ARG, "path %d has %d segments (need >= 1)", p, S);
    if (S > MRS_TG_MAX_SEGMENTS)
      return fail(ctx, MRS_TG_ERR_INVALID_ARG, "path %d has %d segments (at most %d)", p, S, MRS_TG_MAX_SEGMENTS);
    max_S = std::max(max_S, S);
    min_S = std::min(min_S, S);
  }
  HIP_TRY(ctx, use_device(ctx->device));
  mrs_tg_plan* plan = new (std::nothrow) mrs_tg_plan();
  if (!plan) return fail(ctx, MRS_TG_ERR_NOMEM, "out of host memory");
  plan->ctx = ctx;
  plan->seg_offsets_host.assign(so, so + n_paths + 1);
  plan->order_host.resize(n_paths);
  std::iota(plan->order_host.begin(), plan->order_host.end(), 0);
  std::stable_sort(plan->order_host.begin(), plan->order_host.end(),
                   [so](int a, int b) { return (so[a + 1] - so[a]) > (so[b + 1] - so[b]); });
  // slot_start[j] = number of (q, j') pairs with j' < j; paths sorted longest first => slot j is a prefix
  std::vector<int32_t> slot_start(max_S + 1, 0);
  {
    std::vector<int32_t> cnt(max_S + 1, 0);
    for (int p = 0; p < n_paths; ++p) cnt[so[p + 1] - so[p]]++;  // paths with exactly S segments
    int longer = 0;                                              // paths with S > j
    std::vector<int32_t> per_slot(max_S, 0);
    for (int j = max_S - 1; j >= 0; --j) {
      longer += cnt[j + 1];
      per_slot[j] = longer;
    }
    for (int j = 0; j < max_S; ++j) slot_start[j + 1] = slot_start[j] + per_slot[j];
  }
  auto cleanup = [&](int code) {
    mrs_tg_plan_destroy(plan);
    return code;
  };
  // the three structure arrays in ONE device block, uploaded by one copy (a blocking copy costs the host 10-20 us whatever its
  // size, and the nodelet's deviation loop makes a plan per round: three of them were a tenth of a one-request call)
  hipError_t e;
  std::vector<int32_t> packed;
  packed.reserve((size_t)n_paths * 2 + 2 + (size_t)max_S + 1);
  packed.insert(packed.end(), so, so + n_paths + 1);
  const size_t off_order = packed.size();
  packed.insert(packed.end(), plan->order_host.begin(), plan->order_host.end());
  if (n_paths == 0) packed.push_back(0);
  const size_t off_slot = packed.size();
  packed.insert(packed.end(), slot_start.begin(), slot_start.end());
  if ((e = mrs_tg::pool_alloc(&plan->d_seg_offsets, sizeof(int32_t) * packed.size())) != hipSuccess ||
      (e = hipMemcpy(plan->d_seg_offsets, packed.data(), sizeof(int32_t) * packed.size(), hipMemcpyHostToDevice)) != hipSuccess)
    return cleanup(fail(ctx, MRS_TG_ERR_HIP, "plan allocation failed: %s", hipGetErrorString(e)));
  plan->d_order = plan->d_seg_offsets + off_order;        // (parts of the one block: mrs_tg_plan_destroy frees d_seg_offsets)
  plan->d_slot_start = plan->d_seg_offsets + off_slot;
  plan->view.n_paths = n_paths;
  plan->view.n_segments = so[n_paths];
  plan->view.max_segments = max_S;
  plan->view.uniform_S = (n_paths > 0 && min_S == max_S) ? max_S : 0;
  plan->view.seg_offsets = plan->d_seg_offsets;
  plan->view.order = plan->d_order;
  plan->view.slot_start = plan->d_slot_start;
  const int rc = mrs_tg::nonlinear_plan_build(plan->nl, plan->seg_offsets_host, plan->order_host);
  if (rc != 0) return cleanup(fail(ctx, MRS_TG_ERR_HIP, "nonlinear plan setup failed"));
  *plan_out = plan;
  return MRS_TG_OK;
}

void mrs_tg_plan_destroy(mrs_tg_plan* plan) {
  if (!plan) return;
  (void)use_device(plan->ctx->device);
  (void)hipStreamSynchronize(plan->ctx->stream);
  mrs_tg::nonlinear_plan_free(plan->nl);
  if (plan->d_seg_offsets) (void)mrs_tg::pool_free(plan->d_seg_offsets);
  // (d_order and d_slot_start are parts of the block d_seg_offsets heads)
  if (plan->d_ws) (void)mrs_tg::pool_free(plan->d_ws);
  if (plan->d_H) (void)mrs_tg::pool_free(plan->d_H);
  if (plan->d_Ainv) (void)mrs_tg::pool_free(plan->d_Ainv);
  delete plan;
}

int32_t mrs_tg_plan_n_paths(const mrs_tg_plan* plan) { return plan ? plan->view.n_paths : 0; }
int32_t mrs_tg_plan_n_segments(const mrs_tg_plan* plan) { return plan ? plan->view.n_segments : 0; }
int32_t mrs_tg_plan_max_segments(const mrs_tg_plan* plan) { return plan ? plan->view.max_segments : 0; }

int mrs_tg_plan_get_order(const mrs_tg_plan* plan, int32_t* order_out) {
  if (!plan || !order_out) return fail(plan ? plan->ctx : nullptr, MRS_TG_ERR_INVALID_ARG, "NULL argument");
  std::copy(plan->order_host.begin(), plan->order_host.end(), order_out);
  return MRS_TG_OK;
}

size_t mrs_tg_plan_block_bytes(const mrs_tg_plan* plan) {
  if (!plan) return 0;
  return (size_t)plan->view.max_segments * 100 * (size_t)plan->view.n_paths * sizeof(double);
}

int mrs_tg_plan_assemble(mrs_tg_plan* plan, int32_t d, const double* seg_times_dev, double* H_dev, double* Ainv_dev) {
  if (!plan || !seg_times_dev || !H_dev || !Ainv_dev)
    return fail(plan ? plan->ctx : nullptr, MRS_TG_ERR_INVALID_ARG, "NULL argument");
  mrs_tg_ctx* ctx = plan->ctx;
  if (d < 0 || d > 4) return fail(ctx, MRS_TG_ERR_INVALID_ARG, "derivative_to_optimize %d outside [0, 4]", d);
  HIP_TRY(ctx, use_device(ctx->device));
  ProfileScope ps(ctx, 0);
  HIP_TRY(ctx, mrs_tg::launch_assemble(plan->view, d, seg_times_dev, H_dev, Ainv_dev, ctx->stream));
  return MRS_TG_OK;
}

int mrs_tg_plan_solve(mrs_tg_plan* plan, const double* wp, const uint8_t* mask, const double* vals, const double* limits,
                      const mrs_tg_options* opt, double* seg_times, double* coeffs, int32_t* status, double* cost,
                      int32_t* n_samples, double* samples) {
  if (!plan) return fail(nullptr, MRS_TG_ERR_INVALID_ARG, "plan is NULL");
  mrs_tg_ctx* ctx = plan->ctx;
  int rc = check_options(ctx, opt);
  if (rc != MRS_TG_OK) return rc;
  if (!mask || !vals || !seg_times || !coeffs || !status)
    return fail(ctx, MRS_TG_ERR_INVALID_ARG, "fixed_mask, fixed_values, seg_times, coeffs_out and status_out are required");
  if ((opt->estimate_times || opt->time_alloc_method != MRS_TG_TIME_ALLOC_NONE) && !limits)
    return fail(ctx, MRS_TG_ERR_INVALID_ARG, "limits are required for time estimation and for every time-allocation mode");
  if (opt->estimate_times && !wp) return fail(ctx, MRS_TG_ERR_INVALID_ARG, "waypoints are required when estimate_times is set");
  if ((opt->flags & MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS) && !wp)
    return fail(ctx, MRS_TG_ERR_INVALID_ARG, "MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS needs the waypoints array");
  if (opt->sampling_dt > 0 && !n_samples) return fail(ctx, MRS_TG_ERR_INVALID_ARG, "n_samples_out is required when sampling");
  HIP_TRY(ctx, use_device(ctx->device));
  const mrs_tg::BatchView& b = plan->view;
  // MRS_TG_VERIFY_FLAGS=1 (debug / test knob, read once): the statement of MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS is checked on
  // every solve, not only at bind time -- a blocking check; a false statement fails the call instead of solving the
  // waypoints' problem on launches that take the saturated-device kernel and the caller's on the others
  static const bool verify_flags = [] {
    const char* e = std::getenv("MRS_TG_VERIFY_FLAGS");
    return e != nullptr && std::atoi(e) != 0;
  }();
  if (verify_flags && !mrs_tg::dry_run() && (opt->flags & MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS)) {
    long long bad = 0;
    HIP_TRY(ctx, mrs_tg::count_position_mismatches(b, wp, mask, vals, ctx->stream, &bad));
    if (bad != 0)
      return fail(ctx, MRS_TG_ERR_INVALID_ARG,
                  "MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS: %lld vertices whose position constraint is absent or differs from their waypoint",
                  bad);
  }
  const double* pos_wp = (opt->flags & MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS) ? wp : nullptr;
  const int d = opt->derivative_to_optimize;
  struct SharedDeviceScope {
    explicit SharedDeviceScope(bool on) { mrs_tg::set_shared_device_hint(on); }
    ~SharedDeviceScope() { mrs_tg::set_shared_device_hint(false); }
  } shared_scope((opt->flags & MRS_TG_FLAG_SHARED_DEVICE) != 0);
  struct ConstrainedSlotsScope {
    explicit ConstrainedSlotsScope(bool on) { mrs_tg::set_constrained_slots_hint(on); }
    ~ConstrainedSlotsScope() { mrs_tg::set_constrained_slots_hint(false); }
  } slots_scope((opt->flags & MRS_TG_FLAG_CONSTRAINED_SLOTS) != 0);
  bool sampled = false;  // the sampling rode on the final solve's launch
  // (the Mellinger pipeline takes the estimate as its start point itself: mrs_tg::NonlinearParams::estimate_wp)
  if (opt->estimate_times && opt->time_alloc_method != MRS_TG_TIME_ALLOC_MELLINGER)
    HIP_TRY(ctx, mrs_tg::launch_estimate_times(b, wp, limits, seg_times, ctx->stream));

  if (opt->time_alloc_method != MRS_TG_TIME_ALLOC_NONE && opt->time_alloc_method != MRS_TG_TIME_ALLOC_MELLINGER) {
    mrs_tg::DfoParams prm;
    prm.derivative = d;
    prm.mode = opt->time_alloc_method;
    prm.max_iterations = opt->max_iterations;
    prm.f_rel = opt->f_rel;
    prm.f_abs = opt->f_abs;
    prm.x_rel = opt->x_rel;
    prm.x_abs = opt->x_abs;
    prm.time_penalty = opt->time_penalty;
    prm.soft_weight = opt->soft_constraint_weight;
    prm.use_soft = opt->use_soft_constraints;
    prm.initial_stepsize_rel = opt->initial_stepsize_rel;
    prm.time_budget_ticks = opt->max_time_s > 0 ? budget_ticks(ctx, opt->max_time_s) : 0;
    ProfileScope ps(ctx, 2);
    HIP_TRY(ctx, mrs_tg::launch_dfo(plan->nl, b, prm, mask, vals, limits, seg_times, coeffs, status, cost, ctx->stream,
                                    (opt->flags & MRS_TG_FLAG_GENERAL_PATTERNS) != 0));
  } else if (opt->time_alloc_method == MRS_TG_TIME_ALLOC_MELLINGER) {
    mrs_tg::NonlinearParams prm;
    prm.derivative = d;
    prm.max_iterations = opt->max_iterations;
    prm.f_rel = opt->f_rel;
    prm.f_abs = opt->f_abs;
    prm.x_rel = opt->x_rel;
    prm.x_abs = opt->x_abs;
    prm.time_budget_ticks = opt->max_time_s > 0 ? budget_ticks(ctx, opt->max_time_s) : 0;
    if ((opt->flags & MRS_TG_FLAG_CAREFUL_COST) && !mrs_tg::careful_rerun_built())
      return fail(ctx, MRS_TG_ERR_UNSUPPORTED,
                  "MRS_TG_FLAG_CAREFUL_COST: this library was built without the careful re-run (rebuild with -DMRS_TG_WITH_CAREFUL=1)");
    prm.careful_cap = (opt->flags & MRS_TG_FLAG_CAREFUL_COST) ? 1 : 0;
    if (opt->estimate_times) {
      prm.estimate_wp = wp;
      prm.estimate_limits = limits;
    }
    prm.pos_wp = pos_wp;
    prm.reference_status = (opt->flags & MRS_TG_FLAG_REFERENCE_STATUS) != 0;
    ProfileScope ps(ctx, 2);
    HIP_TRY(ctx, mrs_tg::launch_nonlinear(plan->nl, b, prm, mask, vals, limits, seg_times, coeffs, status, cost, ctx->stream,
                                          opt->sampling_dt, opt->sample_capacity, n_samples, samples, &sampled,
                                          (opt->flags & MRS_TG_FLAG_GENERAL_PATTERNS) != 0));
  } else {
    const bool fused = (opt->flags & MRS_TG_FLAG_MATERIALIZED_BLOCKS) == 0;  // the default since ABI 2
    const bool general = (opt->flags & MRS_TG_FLAG_GENERAL_PATTERNS) != 0;   // position-free vertices may occur
    if ((rc = ensure_ws(plan, general ? std::max(mrs_tg::linear_workspace_doubles(b), mrs_tg::general_workspace_doubles(b))
                                      : mrs_tg::linear_workspace_doubles(b))) != MRS_TG_OK)
      return rc;
    if (!fused) {
      if ((rc = ensure_blocks(plan)) != MRS_TG_OK) return rc;
      ProfileScope ps(ctx, 0);
      HIP_TRY(ctx, mrs_tg::launch_assemble(b, d, seg_times, plan->d_H, plan->d_Ainv, ctx->stream));
    }
    ProfileScope ps(ctx, 1);
    if (fused && !general && opt->sampling_dt > 0 && mrs_tg::rows_tail_sampling_pays(b)) {  // solve and sample in one launch
      mrs_tg::RowsTail tail;
      tail.sampling_dt = opt->sampling_dt;
      tail.sample_capacity = opt->sample_capacity;
      tail.n_samples = n_samples;
      tail.samples = samples;
      HIP_TRY(ctx, mrs_tg::launch_solve_rows(b, d, mask, vals, seg_times, coeffs, status, cost, nullptr, ctx->stream, tail));
      sampled = true;
    } else {
      HIP_TRY(ctx, mrs_tg::launch_solve_linear(b, d, fused, mask, vals, seg_times, plan->d_H, plan->d_Ainv, plan->d_ws,
                                               coeffs, status, cost, nullptr, ctx->stream, general ? nullptr : pos_wp));
    }
    // the paths the fast kernels sent back with status -2 (a vertex without a position constraint): 5 x 5 vertex blocks
    if (general)
      HIP_TRY(ctx, mrs_tg::launch_solve_general(b, d, mask, vals, seg_times, plan->d_ws, coeffs, status, cost, ctx->stream));
  }
  if (opt->sampling_dt > 0 && !sampled)
    HIP_TRY(ctx, mrs_tg::launch_sample(b, coeffs, seg_times, opt->sampling_dt, opt->sample_capacity, n_samples, samples,
                                       ctx->stream));
  return MRS_TG_OK;
}

static constexpr size_t kGroupWsPresizeBytes = (size_t)256 << 20;  // bind-time reservation of a grouped launch's factor stores

struct mrs_tg_bound_solve {
  mrs_tg_plan* plan;
  const double* wp;
  const uint8_t* mask;
  const double* vals;
  const double* limits;
  mrs_tg_options opt;
  double* seg_times;
  double* coeffs;
  int32_t* status;
  double* cost;
  int32_t* n_samples;
  double* samples;
};

int mrs_tg_plan_bind_solve(mrs_tg_plan* plan, const double* wp, const uint8_t* mask, const double* vals, const double* limits,
                           const mrs_tg_options* opt, double* seg_times, double* coeffs, int32_t* status, double* cost,
                           int32_t* n_samples, double* samples, mrs_tg_bound_solve** bound_out) {
  if (!plan || !bound_out) return fail(plan ? plan->ctx : nullptr, MRS_TG_ERR_INVALID_ARG, "NULL argument");
  *bound_out = nullptr;
  const int rc = check_options(plan->ctx, opt);
  if (rc != MRS_TG_OK) return rc;
  if (opt->flags & MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS) {
    // the caller's statement, checked once on the arrays as they are now (a blocking check: binding is not a hot path)
    if (!wp || !mask || !vals) return fail(plan->ctx, MRS_TG_ERR_INVALID_ARG, "MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS needs waypoints, fixed_mask and fixed_values");
    mrs_tg_ctx* ctx = plan->ctx;
    HIP_TRY(ctx, use_device(ctx->device));
    long long bad = 0;
    HIP_TRY(ctx, mrs_tg::count_position_mismatches(plan->view, wp, mask, vals, ctx->stream, &bad));
    if (bad != 0)
      return fail(ctx, MRS_TG_ERR_INVALID_ARG,
                  "MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS: %lld vertices whose position constraint is absent or differs from their waypoint",
                  bad);
  }
  mrs_tg_bound_solve* b = new (std::nothrow) mrs_tg_bound_solve{plan, wp, mask, vals, limits, *opt, seg_times, coeffs, status, cost,
                                                                n_samples, samples};
  if (!b) return fail(plan->ctx, MRS_TG_ERR_NOMEM, "out of host memory");
  // A fixed-times default solve can go out in grouped launches (mrs_tg_bound_solve_launch_group), whose saturated-device kernel
  // wants one factor store per batch of the group for paths that need the general step.  SMALL plans reserve it HERE, once,
  // so that the first large group does not pay an allocation and a stream synchronisation inside somebody's timed region;
  // a plan whose full group would hold more than kGroupWsPresizeBytes (1024 x 10: 157 MB for 16 batches; 8192 x 10 would
  // be 1.26 GB, 65536 x 10 ten) reserves nothing at bind time -- most bound solves are never launched in a group -- and
  // mrs_tg_bound_solve_launch_group sizes the store for the group it actually launches.
  if (opt->time_alloc_method == MRS_TG_TIME_ALLOC_NONE && !opt->estimate_times && opt->sampling_dt <= 0 &&
      !(opt->flags & (MRS_TG_FLAG_MATERIALIZED_BLOCKS | MRS_TG_FLAG_GENERAL_PATTERNS)) &&
      mrs_tg::quad_kernel_applies(plan->view, (long long)plan->view.n_paths * mrs_tg::kRowsGroupMax, false) &&
      (size_t)mrs_tg::kRowsGroupMax * mrs_tg::linear_workspace_doubles(plan->view) * sizeof(double) <= kGroupWsPresizeBytes) {
    const int rcw = ensure_ws(plan, (size_t)mrs_tg::kRowsGroupMax * mrs_tg::linear_workspace_doubles(plan->view));
    if (rcw != MRS_TG_OK) {
      delete b;
      return rcw;
    }
  }
  *bound_out = b;
  return MRS_TG_OK;
}

int mrs_tg_bound_solve_launch(mrs_tg_bound_solve* b) {
  if (!b) return fail(nullptr, MRS_TG_ERR_INVALID_ARG, "bound solve is NULL");
  return mrs_tg_plan_solve(b->plan, b->wp, b->mask, b->vals, b->limits, &b->opt, b->seg_times, b->coeffs, b->status, b->cost,
                           b->n_samples, b->samples);
}

void mrs_tg_bound_solve_destroy(mrs_tg_bound_solve* b) { delete b; }

// ---- the issue loop on several host threads ------------------------------------------------------
// One runtime launch costs the host 3.5-4.5 us; four 10 us kernels in flight retire one every 2.6 us, so a single issuing
// thread is what bounds a host that keeps four batches in flight.  Helper threads (created on first use, one per extra
// issuing thread) take the launches of the bound solves k = j, j + T, ... -- the order within a bound solve's stream is kept,
// the streams are independent anyway.  A helper spins for work for kSpinUs after its last job (a server that issues runs
// back to back never pays a wake-up), then sleeps on a condition variable.
namespace {

struct IssueJob {
  mrs_tg_bound_solve* const* bound = nullptr;
  int n_bound = 0, n_launches = 0, me = 0;
  const int* owner = nullptr;  // [n_bound] the thread that issues bound solve i: solves that share a context share a thread
  int rc = MRS_TG_OK;
};

class IssuePool {
 public:
  static IssuePool& instance() {
    static IssuePool pool;
    return pool;
  }
  // runs jobs[1..] on helpers and jobs[0] on the caller; returns when all are done
  void run(std::vector<IssueJob>& jobs) {
    std::lock_guard<std::mutex> one_caller(run_mutex_);  // the workers and their job slots serve one run at a time
    const int helpers = (int)jobs.size() - 1;
    ensure(helpers);
    for (int j = 0; j < helpers; ++j) workers_[j]->post(&jobs[j + 1]);
    issue(jobs[0]);
    for (int j = 0; j < helpers; ++j) workers_[j]->wait();
  }
  static void issue(IssueJob& job) {
    for (int k = 0, i = 0; k < job.n_launches; ++k, i = (i + 1 == job.n_bound) ? 0 : i + 1) {
      if (job.owner[i] != job.me) continue;
      const int rc = mrs_tg_bound_solve_launch(job.bound[i]);
      if (rc != MRS_TG_OK) {
        job.rc = rc;
        return;
      }
    }
  }

 private:
  static constexpr int kSpinUs = 2000;
  struct Worker {
    std::atomic<IssueJob*> job{nullptr};
    std::atomic<bool> done{true}, quit{false};
    std::mutex m;
    std::condition_variable cv;
    std::thread th;
    Worker() : th([this] { loop(); }) {}
    ~Worker() {
      quit.store(true);
      {
        std::lock_guard<std::mutex> lk(m);
      }
      cv.notify_one();
      th.join();
    }
    void post(IssueJob* j) {
      done.store(false, std::memory_order_relaxed);
      job.store(j, std::memory_order_release);
      {
        std::lock_guard<std::mutex> lk(m);  // pairs with the sleeper's predicate check
      }
      cv.notify_one();
    }
    void wait() {
      while (!done.load(std::memory_order_acquire)) __builtin_ia32_pause();
    }
    void loop() {
      while (!quit.load()) {
        IssueJob* j = nullptr;
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(kSpinUs);
        while (!(j = job.exchange(nullptr, std::memory_order_acquire))) {
          if (quit.load()) return;
          if (std::chrono::steady_clock::now() > t_end) {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [this] { return job.load(std::memory_order_acquire) != nullptr || quit.load(); });
          } else {
            __builtin_ia32_pause();
          }
        }
        IssuePool::issue(*j);
        done.store(true, std::memory_order_release);
      }
    }
  };
  void ensure(int helpers) {
    while ((int)workers_.size() < helpers) workers_.emplace_back(new Worker());
  }
  std::vector<std::unique_ptr<Worker>> workers_;
  std::mutex run_mutex_;
};

}  // namespace

int mrs_tg_bound_solve_launch_many_mt(mrs_tg_bound_solve* const* bound, int32_t n_bound, int32_t n_launches,
                                      int32_t n_threads) {
  if (!bound || n_bound < 1 || n_launches < 0) return fail(nullptr, MRS_TG_ERR_INVALID_ARG, "bound solves are required");
  for (int32_t i = 0; i < n_bound; ++i)
    if (!bound[i]) return fail(nullptr, MRS_TG_ERR_INVALID_ARG, "bound solve %d is NULL", i);
  try {
    // A context (its plans' workspaces, its timers, its error string) is driven by one thread at a time: the bound solves
    // are partitioned by CONTEXT, not by index -- every solve of a context goes to the same issuing thread, and the
    // number of threads is at most the number of distinct contexts.
    std::vector<mrs_tg_ctx*> ctxs;
    std::vector<int> owner((size_t)n_bound);
    for (int32_t i = 0; i < n_bound; ++i) {
      mrs_tg_ctx* c = bound[i]->plan->ctx;
      size_t j = 0;
      while (j < ctxs.size() && ctxs[j] != c) ++j;
      if (j == ctxs.size()) ctxs.push_back(c);
      owner[(size_t)i] = (int)j;
    }
    int T = n_threads < 1 ? 1 : n_threads;
    if (T > (int)ctxs.size()) T = (int)ctxs.size();
    if (T > 8) T = 8;
    if (T == 1) return mrs_tg_bound_solve_launch_many(bound, n_bound, n_launches);
    for (int& o : owner) o %= T;
    std::vector<IssueJob> jobs((size_t)T);
    for (int j = 0; j < T; ++j) {
      jobs[j].bound = bound;
      jobs[j].n_bound = n_bound;
      jobs[j].n_launches = n_launches;
      jobs[j].me = j;
      jobs[j].owner = owner.data();
    }
    IssuePool::instance().run(jobs);
    for (const IssueJob& j : jobs)
      if (j.rc != MRS_TG_OK) return j.rc;
    return MRS_TG_OK;
  } catch (const std::bad_alloc&) {
    return fail(nullptr, MRS_TG_ERR_NOMEM, "out of host memory");
  } catch (const std::exception& ex) {  // std::system_error of a thread that could not be started: nothing crosses the C boundary
    return fail(nullptr, MRS_TG_ERR_UNSUPPORTED, "issue threads: %s", ex.what());
  }
}

int mrs_tg_bound_solve_launch_many(mrs_tg_bound_solve* const* bound, int32_t n_bound, int32_t n_launches) {
  if (!bound || n_bound < 1 || n_launches < 0) return fail(nullptr, MRS_TG_ERR_INVALID_ARG, "bound solves are required");
  for (int32_t i = 0; i < n_bound; ++i)
    if (!bound[i]) return fail(nullptr, MRS_TG_ERR_INVALID_ARG, "bound solve %d is NULL", i);
  for (int32_t k = 0, i = 0; k < n_launches; ++k, i = (i + 1 == n_bound) ? 0 : i + 1) {
    const int rc = mrs_tg_bound_solve_launch(bound[i]);
    if (rc != MRS_TG_OK) return rc;
  }
  return MRS_TG_OK;
}

// The issue loop with consecutive launches packed into one dispatch: launch k still solves bound[k % n_bound], but a run of
// consecutive launches whose bound solves share a PLAN (one batch structure, one context and stream; the solves differ in
// their input / output arrays) goes out as a single kernel whose workgroups are divided among the batches, on that plan's
// stream.  bound = [A0, A1, B0, B1] with A*, B* on two plans therefore issues (A0 A1), (B0 B1), (A0 A1), ... alternately on
// the two streams.  Fixed times, the default solve, no sampling.  Every path runs the single-batch kernel's instructions.
int mrs_tg_bound_solve_launch_group(mrs_tg_bound_solve* const* bound, int32_t n_bound, int32_t n_launches) {
  if (!bound || n_bound < 1 || n_launches < 0) return fail(nullptr, MRS_TG_ERR_INVALID_ARG, "bound solves are required");
  for (int32_t i = 0; i < n_bound; ++i)
    if (!bound[i]) return fail(nullptr, MRS_TG_ERR_INVALID_ARG, "bound solve %d is NULL", i);
  for (int32_t i = 0; i < n_bound; ++i) {
    const mrs_tg_bound_solve* b = bound[i];
    mrs_tg_ctx* ctx = b->plan->ctx;
    const mrs_tg_options& o = b->opt;
    if (o.time_alloc_method != MRS_TG_TIME_ALLOC_NONE || o.estimate_times || o.sampling_dt > 0 ||
        (o.flags & (MRS_TG_FLAG_MATERIALIZED_BLOCKS | MRS_TG_FLAG_GENERAL_PATTERNS)))
      return fail(ctx, MRS_TG_ERR_UNSUPPORTED, "grouped launches are for the fixed-times default solve without sampling (solve %d differs)", i);
    if (!b->mask || !b->vals || !b->seg_times || !b->coeffs || !b->status)
      return fail(ctx, MRS_TG_ERR_INVALID_ARG, "fixed_mask, fixed_values, seg_times, coeffs_out and status_out are required (solve %d)", i);
    if (!mrs_tg::rows_kernel_applies(b->plan->view, false))
      return fail(ctx, MRS_TG_ERR_UNSUPPORTED, "grouped launches need a batch the one-lane-per-unknown solve takes (solve %d: paths too long)", i);
  }
  for (int32_t k = 0; k < n_launches;) {
    const mrs_tg_bound_solve* first = bound[k % n_bound];
    mrs_tg_plan* plan = first->plan;
    mrs_tg_ctx* ctx = plan->ctx;
    mrs_tg::RowsGroup g;
    bool group_constrained_slots = false;
    for (; g.n < mrs_tg::kRowsGroupMax && k < n_launches; ++k, ++g.n) {
      const mrs_tg_bound_solve* b = bound[k % n_bound];
      if (b->plan != plan || b->opt.derivative_to_optimize != first->opt.derivative_to_optimize) break;
      if (g.n > 0 && b == first) break;  // the round is complete: the same arrays twice in one launch would be written twice
      g.mask[g.n] = b->mask;
      g.vals[g.n] = b->vals;
      g.seg_times[g.n] = b->seg_times;
      g.coeffs[g.n] = b->coeffs;
      g.status[g.n] = b->status;
      g.cost[g.n] = b->cost;
      g.pos_wp[g.n] = (b->opt.flags & MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS) ? b->wp : nullptr;
      group_constrained_slots = group_constrained_slots || (b->opt.flags & MRS_TG_FLAG_CONSTRAINED_SLOTS) != 0;
    }
    if (plan->view.n_paths == 0) continue;
    HIP_TRY(ctx, use_device(ctx->device));
    ProfileScope ps(ctx, 1);
    // MRS_TG_FLAG_CONSTRAINED_SLOTS of any batch of the group: the dispatch runs the instantiations that take stop_at vertices
    // inside the specialised sweeps (the hint was only set by mrs_tg_plan_solve)
    struct HintScope {
      explicit HintScope(bool on) { mrs_tg::set_constrained_slots_hint(on); }
      ~HintScope() { mrs_tg::set_constrained_slots_hint(false); }
    } hint_scope(group_constrained_slots);
    if (mrs_tg::quad_kernel_applies(plan->view, (long long)plan->view.n_paths * g.n, false)) {
      // the dispatch carries more paths than the rows kernel has wavefront slots for: four lanes per path, factors in LDS
      // (one factor store per batch of THIS group; grows when a larger group comes, never shrinks)
      int rc = ensure_ws(plan, (size_t)g.n * mrs_tg::linear_workspace_doubles(plan->view));
      if (rc != MRS_TG_OK) return rc;
      HIP_TRY(ctx, mrs_tg::launch_solve_quad_group(plan->view, first->opt.derivative_to_optimize, g, plan->d_ws, ctx->stream));
    } else {
      HIP_TRY(ctx, mrs_tg::launch_solve_rows_group(plan->view, first->opt.derivative_to_optimize, g, ctx->stream));
    }
  }
  return MRS_TG_OK;
}

// The routing table, from the routers themselves: the calling thread runs the very launch functions a solve would run, in dry
// mode -- every size rule, environment knob and hint takes effect, kernels are noted instead of enqueued.
int mrs_tg_plan_explain(mrs_tg_plan* plan, const mrs_tg_options* opt, int32_t group_size, const char** names_out, int32_t capacity) {
  if (!plan || !names_out || capacity <= 0) return fail(plan ? plan->ctx : nullptr, MRS_TG_ERR_INVALID_ARG, "plan, names_out and a positive capacity are required");
  mrs_tg_ctx* ctx = plan->ctx;
  int rc = check_options(ctx, opt);
  if (rc != MRS_TG_OK) return rc;
  if (group_size < 0 || group_size > mrs_tg::kRowsGroupMax)
    return fail(ctx, MRS_TG_ERR_INVALID_ARG, "group_size %d: a dispatch carries 1 .. %d batches (0 = a single solve)", group_size, mrs_tg::kRowsGroupMax);
  // stand-ins for the caller's device arrays: never dereferenced (nothing is enqueued), only tested against NULL
  static double dummy_d[4];
  static uint8_t dummy_b[4];
  static int32_t dummy_i[4];
  struct DryScope {
    mrs_tg_ctx* c;
    bool was_profiling;
    explicit DryScope(mrs_tg_ctx* ctx_) : c(ctx_), was_profiling(ctx_->profiling) {
      c->profiling = false;  // (no events are armed for launches that do not happen)
      mrs_tg::set_dry_run(true);
      mrs_tg::kernel_trace_reset();
    }
    ~DryScope() {
      mrs_tg::set_dry_run(false);
      c->profiling = was_profiling;
    }
  };
  {
    DryScope scope(ctx);
    if (group_size == 0) {
      rc = mrs_tg_plan_solve(plan, dummy_d, dummy_b, dummy_d, dummy_d, opt, dummy_d, dummy_d, dummy_i, dummy_d,
                             opt->sampling_dt > 0 ? dummy_i : nullptr, opt->sampling_dt > 0 ? dummy_d : nullptr);
    } else {
      std::vector<mrs_tg_bound_solve> solves((size_t)group_size,
                                             mrs_tg_bound_solve{plan, dummy_d, dummy_b, dummy_d, dummy_d, *opt, dummy_d, dummy_d, dummy_i, dummy_d, nullptr, nullptr});
      std::vector<mrs_tg_bound_solve*> ptrs;
      for (mrs_tg_bound_solve& b : solves) ptrs.push_back(&b);
      rc = mrs_tg_bound_solve_launch_group(ptrs.data(), group_size, group_size);
    }
  }
  if (rc != MRS_TG_OK) return rc;
  return mrs_tg::kernel_trace(names_out, capacity);
}

int mrs_tg_plan_cost_gradient(mrs_tg_plan* plan, int32_t d, const uint8_t* mask, const double* vals,
                              const double* seg_times, double* cost, double* grad) {
  if (!plan || !mask || !vals || !seg_times || !cost || !grad)
    return fail(plan ? plan->ctx : nullptr, MRS_TG_ERR_INVALID_ARG, "NULL argument");
  mrs_tg_ctx* ctx = plan->ctx;
  if (d < 0 || d > 4) return fail(ctx, MRS_TG_ERR_INVALID_ARG, "derivative_to_optimize %d outside [0, 4]", d);
  HIP_TRY(ctx, use_device(ctx->device));
  HIP_TRY(ctx, mrs_tg::launch_cost_gradient(plan->nl, plan->view, d, mask, vals, seg_times, cost, grad, ctx->stream));
  return MRS_TG_OK;
}

int mrs_tg_plan_segment_maxima(mrs_tg_plan* plan, const double* coeffs, const double* seg_times, double* maxima) {
  if (!plan || !coeffs || !seg_times || !maxima)
    return fail(plan ? plan->ctx : nullptr, MRS_TG_ERR_INVALID_ARG, "NULL argument");
  mrs_tg_ctx* ctx = plan->ctx;
  HIP_TRY(ctx, use_device(ctx->device));
  HIP_TRY(ctx, mrs_tg::launch_segment_maxima(plan->view, coeffs, seg_times, maxima, ctx->stream));
  return MRS_TG_OK;
}

int mrs_tg_plan_careful_count(mrs_tg_plan* plan, int32_t* count_out) {
  if (!plan || !count_out) return fail(plan ? plan->ctx : nullptr, MRS_TG_ERR_INVALID_ARG, "NULL argument");
  mrs_tg_ctx* ctx = plan->ctx;
  *count_out = 0;
  HIP_TRY(ctx, use_device(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (!plan->nl.d_careful) return MRS_TG_OK;  // no outer loop has run on this plan
  HIP_TRY(ctx, hipMemcpy(count_out, plan->nl.d_careful + 2, sizeof(int32_t), hipMemcpyDeviceToHost));
  return MRS_TG_OK;
}

int mrs_tg_plan_sample_states(mrs_tg_plan* plan, const double* coeffs, const double* seg_times, double sampling_dt,
                              int32_t sample_capacity, int32_t* n_samples, double* states) {
  if (!plan || !coeffs || !seg_times || !n_samples)
    return fail(plan ? plan->ctx : nullptr, MRS_TG_ERR_INVALID_ARG, "NULL argument");
  mrs_tg_ctx* ctx = plan->ctx;
  if (!(sampling_dt > 0.0) || sample_capacity < 0 || (sample_capacity > 0 && !states))
    return fail(ctx, MRS_TG_ERR_INVALID_ARG, "sampling_dt must be positive and states_out_dev given for a positive capacity");
  HIP_TRY(ctx, use_device(ctx->device));
  HIP_TRY(ctx, mrs_tg::launch_sample_states(plan->view, coeffs, seg_times, sampling_dt, sample_capacity, n_samples,
                                            sample_capacity > 0 ? states : nullptr, ctx->stream));
  return MRS_TG_OK;
}

// ---- one-call host interface ------------------------------------------------------------------


extern "C++" {
namespace mrs_tg {

void* ctx_host_scratch(mrs_tg_ctx* ctx, size_t bytes) {
  if (!ctx) return nullptr;
  if (ctx->h_scratch_bytes >= bytes && ctx->h_scratch) return ctx->h_scratch;
  if (ctx->h_scratch) (void)hipHostFree(ctx->h_scratch);
  ctx->h_scratch = nullptr;
  ctx->h_scratch_bytes = 0;
  const size_t want = bytes + bytes / 4;  // (a later round's arrays are a little larger: its paths have more waypoints)
  if (hipHostMalloc(&ctx->h_scratch, want, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    ctx->h_scratch = nullptr;
    return nullptr;
  }
  ctx->h_scratch_bytes = want;
  return ctx->h_scratch;
}

}  // namespace mrs_tg
}  // extern "C++"

static int solve_batch_impl(mrs_tg_ctx* ctx, int32_t n_paths, const int32_t* so, const double* wp, const uint8_t* mask,
                            const double* vals, const double* limits, const mrs_tg_options* opt, double* seg_times,
                            double* coeffs, int32_t* status, double* cost, int32_t* n_samples, double* samples,
                            bool coeffs_required);

// the plan of the previous call is kept: the same batch shape again (a server's fixed batch, the re-solves of the nodelet's
// deviation loop) costs no analysis, no structure upload and no workspace allocation
static int cached_plan_for(mrs_tg_ctx* ctx, int32_t n_paths, const int32_t* so, mrs_tg_plan** plan_out) {
  mrs_tg_plan* plan = ctx->cached_plan;
  if (plan && (plan->view.n_paths != n_paths ||
               std::memcmp(plan->seg_offsets_host.data(), so, sizeof(int32_t) * ((size_t)n_paths + 1)) != 0)) {
    ctx->cached_plan = nullptr;
    mrs_tg_plan_destroy(plan);
    plan = nullptr;
  }
  if (!plan) {
    const int rc = mrs_tg_plan_create(ctx, n_paths, so, &plan);
    if (rc != MRS_TG_OK) return rc;
    ctx->cached_plan = plan;
  }
  *plan_out = plan;
  return MRS_TG_OK;
}

extern "C++" {
namespace mrs_tg {

// One round of optimize() for the `n_paths` requests still active: vertex expansion, findTrajectory's solve (one batched
// pipeline on the cached plan of this shape), both gates and validateTrajectorySpatial on the device.  `in.block` is the host
// block of policy_round_layout (pinned: read and written by the GPU in place through ONE copy kernel each way; pageable: by the
// runtime's copies); when the call returns its result region holds ok / n_samples / status / max_deviation / is_safe per
// path, the safe flag of every segment, and the sample rows of the paths that are finished.
int policy_round_device(mrs_tg_ctx* ctx, const PolicyRoundIn& in) {
  if (!ctx || !in.block || in.n_paths <= 0) return fail(ctx, MRS_TG_ERR_INVALID_ARG, "policy round: nothing to solve");
  const size_t A = (size_t)in.n_paths, nS = in.n_segments, nV = in.n_vertices;
  const int cap = in.sample_capacity;
  const PolicyRoundLayout L = policy_round_layout(A, nS, cap);
  HIP_TRY(ctx, use_device(ctx->device));
  hipStream_t s = ctx->stream;
  mrs_tg_plan* plan = nullptr;
  int rc = cached_plan_for(ctx, in.n_paths, in.seg_offsets, &plan);
  if (rc != MRS_TG_OK) return rc;
  // device arena: [the block's input region] | results (small) | mask | values | times | coefficients | cost | status | n | rows | samples
  auto up = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
  size_t off = L.in_bytes;
  const size_t o_res = off;
  off += L.samples - L.in_bytes;  // the small results, laid out as in the block
  const size_t o_mask = off;
  off += up(nV * 5);
  const size_t o_vals = off;
  off += up(nV * 20 * sizeof(double));
  const size_t o_t = off;
  off += up(nS * sizeof(double));
  const size_t o_c = off;
  off += up(nS * 40 * sizeof(double));
  const size_t o_cost = off;
  off += up(A * sizeof(double));
  const size_t o_st = off;
  off += up(A * sizeof(int32_t));
  const size_t o_ns = off;
  off += up(A * sizeof(int32_t));
  const size_t o_rows = off;
  off += up(A * sizeof(int32_t));
  const size_t o_smp = off;
  off += up(A * (size_t)cap * 4 * sizeof(double));
  if (ctx->d_arena_bytes < off) {
    if (ctx->d_arena) {
      (void)hipStreamSynchronize(s);  // pool contract: no work in flight on a block that is given back
      (void)mrs_tg::pool_free(ctx->d_arena);
    }
    ctx->d_arena = nullptr;
    ctx->d_arena_bytes = 0;
    const size_t want = off + off / 2;  // (the next round's paths have more waypoints: one allocation for a request's rounds)
    HIP_TRY(ctx, mrs_tg::pool_alloc(&ctx->d_arena, want));
    ctx->d_arena_bytes = want;
  }
  char* d = static_cast<char*>(ctx->d_arena);
  struct SyncOnExit {  // nothing of this call is in flight when it returns, on error paths as well (the arena is reused)
    hipStream_t st;
    ~SyncOnExit() { (void)hipStreamSynchronize(st); }
  } sync_on_exit{s};
  // is the block memory the GPU addresses (hipHostMalloc: ctx_host_scratch)?  Then one copy kernel moves the inputs, and the
  // results are written into it directly
  hipPointerAttribute_t at;
  bool pinned = hipPointerGetAttributes(&at, in.block) == hipSuccess && at.type == hipMemoryTypeHost;
  if (!pinned) (void)hipGetLastError();
  char* blk_dev = pinned ? static_cast<char*>(at.devicePointer ? at.devicePointer : (void*)in.block) : nullptr;
  if (pinned) {
    mrs_tg::CopyList upl;
    upl.add(blk_dev, d, L.in_bytes);
    HIP_TRY(ctx, mrs_tg::launch_copy_many(upl, s));
  } else {
    HIP_TRY(ctx, hipMemcpyAsync(d, in.block, L.in_bytes, hipMemcpyHostToDevice, s));
  }
  const double* wp_d = reinterpret_cast<const double*>(d + L.wp);
  uint8_t* mask_d = reinterpret_cast<uint8_t*>(d + o_mask);
  double* vals_d = reinterpret_cast<double*>(d + o_vals);
  HIP_TRY(ctx, mrs_tg::launch_policy_expand((int)nV, in.opt.derivative_to_optimize, wp_d, reinterpret_cast<const int32_t*>(d + L.vinfo),
                                            reinterpret_cast<const double*>(d + L.init), mask_d, vals_d, s));
  mrs_tg_options opt = in.opt;
  opt.estimate_times = 1;
  opt.sample_capacity = cap;
  opt.flags |= MRS_TG_FLAG_REFERENCE_STATUS;  // the length check below is the reference's answer to a runaway (:1178-1199)
  int32_t* st_d = reinterpret_cast<int32_t*>(d + o_st);
  int32_t* ns_d = reinterpret_cast<int32_t*>(d + o_ns);
  double* smp_d = reinterpret_cast<double*>(d + o_smp);
  rc = mrs_tg_plan_solve(plan, wp_d, mask_d, vals_d, reinterpret_cast<const double*>(d + L.lim), &opt, reinterpret_cast<double*>(d + o_t),
                         reinterpret_cast<double*>(d + o_c), st_d, reinterpret_cast<double*>(d + o_cost), ns_d, smp_d);
  if (rc != MRS_TG_OK) return rc;
  // gates + validateTrajectorySpatial where the samples are; the small results go where the host reads them
  char* res = pinned ? blk_dev : d;  // (res + L.<field> addresses the field in either place: the arena mirrors the block)
  (void)o_res;
  mrs_tg::PolicyValidateArgs va{};
  va.n_paths = in.n_paths;
  va.seg_offsets = reinterpret_cast<const int32_t*>(d + L.so);
  va.wp = wp_d;
  va.samples = smp_d;
  va.n_samples = ns_d;
  va.status = st_d;
  va.baca_total = reinterpret_cast<const double*>(d + L.baca);
  va.dt = opt.sampling_dt;
  va.max_len_factor = in.max_len_factor;
  va.min_len_factor = in.min_len_factor;
  va.max_deviation = in.max_deviation;
  va.capacity = cap;
  va.first_segment = in.first_segment;
  va.check_enabled = in.check_enabled;
  va.last_round = in.last_round;
  va.ok_out = reinterpret_cast<int32_t*>(res + L.ok);
  va.ns_out = reinterpret_cast<int32_t*>(res + L.ns);
  va.status_out = reinterpret_cast<int32_t*>(res + L.status);
  va.max_dev_out = reinterpret_cast<double*>(res + L.max_dev);
  va.is_safe_out = reinterpret_cast<uint8_t*>(res + L.is_safe);
  va.safe_out = reinterpret_cast<uint8_t*>(res + L.safe);
  va.ns_copy = reinterpret_cast<int32_t*>(d + o_rows);
  HIP_TRY(ctx, mrs_tg::launch_policy_validate(va, s));
  if (pinned) {  // the finished paths' rows only
    HIP_TRY(ctx, mrs_tg::launch_copy_samples(smp_d, reinterpret_cast<double*>(blk_dev + L.samples), va.ns_copy, in.n_paths, cap, s));
  } else {
    HIP_TRY(ctx, hipMemcpyAsync(in.block + L.ok, d + o_res, L.samples - L.ok, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(in.block + L.samples, smp_d, A * (size_t)cap * 4 * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return MRS_TG_OK;
}

}  // namespace mrs_tg
}  // extern "C++"

int mrs_tg_solve_batch(mrs_tg_ctx* ctx, int32_t n_paths, const int32_t* so, const double* wp, const uint8_t* mask,
                       const double* vals, const double* limits, const mrs_tg_options* opt, double* seg_times,
                       double* coeffs, int32_t* status, double* cost, int32_t* n_samples, double* samples) {
  return solve_batch_impl(ctx, n_paths, so, wp, mask, vals, limits, opt, seg_times, coeffs, status, cost, n_samples, samples, true);
}

extern "C++" {
namespace mrs_tg {
// mrs_tg_solve_batch for a caller that reads only the samples (the policy layer): the coefficients stay on the device
int solve_batch_samples_only(mrs_tg_ctx* ctx, int32_t n_paths, const int32_t* so, const double* wp, const uint8_t* mask,
                             const double* vals, const double* limits, const mrs_tg_options* opt, double* seg_times,
                             int32_t* status, int32_t* n_samples, double* samples) {
  return solve_batch_impl(ctx, n_paths, so, wp, mask, vals, limits, opt, seg_times, nullptr, status, nullptr, n_samples, samples,
                          false);
}
}  // namespace mrs_tg
}  // extern "C++"

static int solve_batch_impl(mrs_tg_ctx* ctx, int32_t n_paths, const int32_t* so, const double* wp, const uint8_t* mask,
                            const double* vals, const double* limits, const mrs_tg_options* opt, double* seg_times,
                            double* coeffs, int32_t* status, double* cost, int32_t* n_samples, double* samples,
                            bool coeffs_required) {
  if (!ctx) return fail(nullptr, MRS_TG_ERR_INVALID_ARG, "ctx is NULL");
  int rc = check_options(ctx, opt);
  if (rc != MRS_TG_OK) return rc;
  if (!so || !mask || !vals || !seg_times || (!coeffs && coeffs_required) || !status)
    return fail(ctx, MRS_TG_ERR_INVALID_ARG, "seg_offsets, fixed_mask, fixed_values, seg_times, coeffs_out, status_out are required");
  if (n_paths == 0) return MRS_TG_OK;
  const auto t_call = std::chrono::steady_clock::now();
  mrs_tg_plan* plan = nullptr;
  if ((rc = cached_plan_for(ctx, n_paths, so, &plan)) != MRS_TG_OK) return rc;
  const size_t nS = (size_t)so[n_paths], nV = nS + (size_t)n_paths;
  const bool sampling = opt->sampling_dt > 0;
  if (sampling && !n_samples) return fail(ctx, MRS_TG_ERR_INVALID_ARG, "n_samples_out is required when sampling");
  const size_t samp_doubles = sampling && samples ? (size_t)n_paths * (size_t)opt->sample_capacity * 4 : 0;
  HIP_TRY(ctx, use_device(ctx->device));
  hipStream_t s = ctx->stream;

  // ---- the arrays of the call.  Every one gets its place in ONE device block kept by the context (no allocation per call
  // once a batch shape has been seen).  How an array travels depends on where the caller keeps it:
  //   * pinned host memory (mrs_tg_host_alloc / mrs_tg_host_register, or any hipHostMalloc'ed block), which the GPU
  //     addresses directly: all pinned inputs are gathered by ONE copy kernel, all pinned outputs scattered by one (a
  //     kernel launch costs the host ~3 us, a hipMemcpyAsync 10-25 us, and seven of those were half of a 1024-path call).
  //     Fixed-times mode with every array pinned needs no copy at all: the solve kernel reads the caller's inputs once and
  //     writes the caller's outputs once, over PCIe, while it computes;
  //   * pageable memory, small (<= stage_max bytes): packed into the context's pinned staging block, which travels with the
  //     pinned arrays in the same copy kernel -- one transfer each way, no synchronisation in between;
  //   * pageable memory, large: hipMemcpyAsync on the caller's buffer (the runtime pins the pages in place; staging 3 MB
  //     of coefficients through another host copy costs more than that).
  // Waypoints are only read by the time estimator: not uploaded when estimate_times is off.
  static const size_t stage_max = [] {
    const char* e = std::getenv("MRS_TG_STAGE_MAX_BYTES");
    return e ? (size_t)std::atoll(e) : (size_t)256 * 1024;
  }();
  static const bool zero_copy_allowed = [] {
    const char* e = std::getenv("MRS_TG_ZERO_COPY");
    return e == nullptr || std::atoi(e) != 0;
  }();
  struct Arr {
    const void* src;  // host source (inputs)
    void* dst;        // host destination (outputs)
    size_t bytes;
    bool staged;
    void* pinned;     // device-side address of the caller's array when it lives in pinned memory
    size_t off;       // offset in the device arena
  };
  // device-side address of a host range the GPU can address as a whole: first AND last byte must be pinned and lie in the
  // same mapping (a range that mrs_tg_host_register covers only in part, or an allocation shorter than the batch implies,
  // takes the copying route -- where a short array is a host-side fault of the caller's, not a GPU page fault)
  auto pinned_address = [](const void* ptr, size_t bytes) -> void* {
    auto query = [](const void* q, hipPointerAttribute_t* at) {
      if (hipPointerGetAttributes(at, q) != hipSuccess) {
        (void)hipGetLastError();  // an ordinary (pageable) pointer is reported as an error by some runtimes
        return false;
      }
      return at->type == hipMemoryTypeHost;
    };
    hipPointerAttribute_t first, last;
    if (!query(ptr, &first)) return nullptr;
    void* dev_first = first.devicePointer ? first.devicePointer : const_cast<void*>(ptr);
    if (bytes > 1) {
      const char* end = static_cast<const char*>(ptr) + (bytes - 1);
      if (!query(end, &last)) return nullptr;
      const char* dev_last = static_cast<const char*>(last.devicePointer ? last.devicePointer : (void*)end);
      if (dev_last - static_cast<const char*>(dev_first) != (ptrdiff_t)(bytes - 1)) return nullptr;  // two mappings
    }
    return dev_first;
  };
  auto make = [&](const void* src, void* dst, size_t bytes) {
    const void* host = src ? src : dst;
    Arr a{src, dst, bytes, false, nullptr, 0};
    if (host != nullptr && bytes > 0) {
      a.pinned = pinned_address(host, bytes);
      a.staged = a.pinned == nullptr && bytes <= stage_max;
    }
    return a;
  };
  // (the waypoints travel when something reads them: the time estimate, or kernels told that positions are the waypoints)
  const bool want_wp = wp != nullptr && (opt->estimate_times != 0 || (opt->flags & MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS) != 0);
  enum { A_WP, A_MASK, A_VALS, A_LIM, A_T, A_C, A_ST, A_COST, A_NS, A_SMP, A_COUNT };
  Arr arr[A_COUNT] = {
      make(want_wp ? wp : nullptr, nullptr, want_wp ? nV * 4 * sizeof(double) : 0),
      make(mask, nullptr, nV * 5),
      make(vals, nullptr, nV * 20 * sizeof(double)),
      make(limits, nullptr, limits ? (size_t)n_paths * 9 * sizeof(double) : 0),
      make(seg_times, seg_times, nS * sizeof(double)),
      make(nullptr, coeffs, nS * 40 * sizeof(double)),
      make(nullptr, status, (size_t)n_paths * sizeof(int32_t)),
      make(nullptr, cost, (size_t)n_paths * sizeof(double)),   // the kernels want a cost buffer even when the caller does not
      make(nullptr, sampling ? n_samples : nullptr, (size_t)n_paths * sizeof(int32_t)),
      make(nullptr, samp_doubles ? samples : nullptr, samp_doubles * sizeof(double)),
  };
  // device layout: unstaged inputs | staged inputs | seg_times (in and out) | staged outputs | unstaged outputs: the staged
  // arrays of each direction are one contiguous span, and the host arena mirrors [staged inputs | seg_times | staged outputs]
  const int in_ids[4] = {A_WP, A_MASK, A_VALS, A_LIM};
  const int out_ids[5] = {A_C, A_ST, A_COST, A_NS, A_SMP};
  auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t off = 0;
  for (int id : in_ids)
    if (!arr[id].staged) { arr[id].off = off; off += align(arr[id].bytes); }
  const size_t span_begin = off;
  for (int id : in_ids)
    if (arr[id].staged) { arr[id].off = off; off += align(arr[id].bytes); }
  const size_t t_off = off;
  arr[A_T].off = off;
  off += align(arr[A_T].bytes);
  const size_t in_span_end = arr[A_T].staged ? off : t_off;
  const size_t out_span_begin = arr[A_T].staged ? t_off : off;
  for (int id : out_ids)
    if (arr[id].staged) { arr[id].off = off; off += align(arr[id].bytes); }
  const size_t span_end = off;
  for (int id : out_ids)
    if (!arr[id].staged) { arr[id].off = off; off += align(arr[id].bytes ? arr[id].bytes : 8); }
  const size_t d_need = off ? off : 256, h_need = span_end - span_begin;
  if (ctx->d_arena_bytes < d_need) {
    if (ctx->d_arena) (void)mrs_tg::pool_free(ctx->d_arena);
    ctx->d_arena = nullptr;
    ctx->d_arena_bytes = 0;
    HIP_TRY(ctx, mrs_tg::pool_alloc(&ctx->d_arena, d_need));
    ctx->d_arena_bytes = d_need;
  }
  if (ctx->h_arena_bytes < h_need) {
    if (ctx->h_arena) (void)hipHostFree(ctx->h_arena);
    ctx->h_arena = nullptr;
    ctx->h_arena_bytes = 0;
    HIP_TRY(ctx, hipHostMalloc(&ctx->h_arena, h_need, hipHostMallocDefault));
    ctx->h_arena_bytes = h_need;
  }
  char* dbase = static_cast<char*>(ctx->d_arena);
  char* hbase = static_cast<char*>(ctx->h_arena) - span_begin;  // hbase + device offset = the array's place in the host arena
  struct SyncOnExit {  // nothing of this call is in flight when it returns, on error paths as well (the arenas are reused)
    hipStream_t st;
    ~SyncOnExit() { (void)hipStreamSynchronize(st); }
  } sync_on_exit{s};

  mrs_tg_options local = *opt;
  if (!(local.flags & MRS_TG_FLAG_GENERAL_PATTERNS)) {
    // the masks are in host memory here: a vertex without a position constraint switches the general solver on
    for (size_t v = 0; v < nV; ++v)
      if (mask[v * 5] == 0) {
        local.flags |= MRS_TG_FLAG_GENERAL_PATTERNS;
        break;
      }
  }
  if (!(local.flags & MRS_TG_FLAG_CONSTRAINED_SLOTS) && local.derivative_to_optimize == 4) {
    // ... and an interior vertex with a constrained derivative slot (a stop_at waypoint) the instantiations that take it
    for (int32_t p = 0; p < n_paths && !(local.flags & MRS_TG_FLAG_CONSTRAINED_SLOTS); ++p)
      for (size_t v = (size_t)so[p] + p + 1; v < (size_t)so[p + 1] + p; ++v)
        if (mask[v * 5 + 1] | mask[v * 5 + 2] | mask[v * 5 + 3] | mask[v * 5 + 4]) {
          local.flags |= MRS_TG_FLAG_CONSTRAINED_SLOTS;
          break;
        }
  }
  // ... and a path that starts from a moving state (non-zero constrained derivatives at its first vertex): a hint for the
  // outer loop's launch shape only (small batches of 13-15 segments; launch_nonlinear)
  bool moving_starts = false;
  if (local.time_alloc_method == MRS_TG_TIME_ALLOC_MELLINGER && n_paths <= 1536 && plan->view.max_segments >= 13 &&
      plan->view.max_segments <= 15) {
    for (int32_t p = 0; p < n_paths && !moving_starts; ++p) {
      const size_t v = (size_t)so[p] + p;
      for (int k = 1; k < 5 && !moving_starts; ++k)
        if (mask[v * 5 + k])
          for (int q = 0; q < 4; ++q) moving_starts = moving_starts || vals[(v * 5 + k) * 4 + q] != 0.0;
    }
  }
  struct MovingScope {
    explicit MovingScope(bool on) { mrs_tg::set_moving_starts_hint(on); }
    ~MovingScope() { mrs_tg::set_moving_starts_hint(false); }
  } moving_scope(moving_starts);
  // zero copy: one pass over every array (fixed times, default solve) and every array the caller passed is pinned
  bool zero_copy = zero_copy_allowed && local.time_alloc_method == MRS_TG_TIME_ALLOC_NONE && !local.estimate_times &&
                   (local.flags & (MRS_TG_FLAG_GENERAL_PATTERNS | MRS_TG_FLAG_MATERIALIZED_BLOCKS)) == 0 &&
                   mrs_tg::rows_kernel_applies(plan->view, sampling) && (!sampling || mrs_tg::rows_tail_sampling_pays(plan->view));
  for (int id = 0; id < A_COUNT && zero_copy; ++id) {
    const Arr& a = arr[id];
    if (id == A_LIM) continue;  // the fixed-times solve never reads the limits: a pageable limits array does not decide this
    if ((a.src || a.dst) && a.bytes && !a.pinned) zero_copy = false;
  }
  // where the kernels find array `id`: the caller's own (pinned) memory under zero copy, its slot of the arena otherwise
  auto dev = [&](int id) -> void* {
    const Arr& a = arr[id];
    return (zero_copy && a.pinned && a.bytes) ? a.pinned : static_cast<void*>(dbase + a.off);
  };

  // ---- host to device
  if (!zero_copy) {
    mrs_tg::CopyList up;
    for (int id : {A_WP, A_MASK, A_VALS, A_LIM, A_T}) {
      const Arr& a = arr[id];
      if (!a.src || !a.bytes) continue;
      if (a.staged) std::memcpy(hbase + a.off, a.src, a.bytes);
      else if (a.pinned) up.add(a.pinned, dbase + a.off, a.bytes);
      else HIP_TRY(ctx, hipMemcpyAsync(dbase + a.off, a.src, a.bytes, hipMemcpyHostToDevice, s));
    }
    if (in_span_end > span_begin) up.add(hbase + span_begin, dbase + span_begin, in_span_end - span_begin);
    HIP_TRY(ctx, mrs_tg::launch_copy_many(up, s));
  }
  if (local.max_time_s > 0) {  // what is left of the caller's budget when the kernels start
    const double spent = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_call).count();
    local.max_time_s = local.max_time_s - spent > 1.0e-9 ? local.max_time_s - spent : 1.0e-9;
  }
  // (estimate_times without waypoints is refused by mrs_tg_plan_solve, as before)
  rc = mrs_tg_plan_solve(plan, want_wp ? static_cast<double*>(dev(A_WP)) : nullptr, static_cast<uint8_t*>(dev(A_MASK)),
                         static_cast<double*>(dev(A_VALS)), limits ? static_cast<double*>(dev(A_LIM)) : nullptr, &local,
                         static_cast<double*>(dev(A_T)), static_cast<double*>(dev(A_C)), static_cast<int32_t*>(dev(A_ST)),
                         static_cast<double*>(dev(A_COST)), sampling ? static_cast<int32_t*>(dev(A_NS)) : nullptr,
                         samp_doubles ? static_cast<double*>(dev(A_SMP)) : nullptr);
  if (rc != MRS_TG_OK) return rc;

  // ---- device to host: pinned arrays and the staged span in one copy kernel, large pageable arrays by the runtime; one
  // synchronisation
  if (!zero_copy) {
    mrs_tg::CopyList down;
    if (span_end > out_span_begin) down.add(dbase + out_span_begin, hbase + out_span_begin, span_end - out_span_begin);
    for (int id : {A_T, A_C, A_ST, A_COST, A_NS, A_SMP}) {
      const Arr& a = arr[id];
      if (!a.dst || !a.bytes || a.staged) continue;
      if (a.pinned && id == A_SMP)  // only the rows every path has produced (its capacity is sized for the longest acceptable one)
        HIP_TRY(ctx, mrs_tg::launch_copy_samples(static_cast<const double*>(dev(A_SMP)), static_cast<double*>(a.pinned),
                                                 static_cast<const int32_t*>(dev(A_NS)), n_paths, opt->sample_capacity, s));
      else if (a.pinned) down.add(dbase + a.off, a.pinned, a.bytes);
      else HIP_TRY(ctx, hipMemcpyAsync(a.dst, dbase + a.off, a.bytes, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(ctx, mrs_tg::launch_copy_many(down, s));
  }
  HIP_TRY(ctx, hipStreamSynchronize(s));
  if (!zero_copy)
    for (int id : {A_T, A_C, A_ST, A_COST, A_NS, A_SMP}) {
      const Arr& a = arr[id];
      if (a.dst && a.bytes && a.staged) std::memcpy(a.dst, hbase + a.off, a.bytes);
    }
  return MRS_TG_OK;
}

// Pinned host memory for the arrays of mrs_tg_solve_batch / mrs_tg_optimize_paths: buffers allocated (or registered) here
// are read and written by the GPU's DMA engines directly, with no staging copy on either side.
int mrs_tg_host_alloc(size_t bytes, void** ptr_out) {
  if (!ptr_out) return fail(nullptr, MRS_TG_ERR_INVALID_ARG, "ptr_out is NULL");
  *ptr_out = nullptr;
  const hipError_t e = hipHostMalloc(ptr_out, bytes ? bytes : 8, hipHostMallocDefault);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(nullptr, e == hipErrorOutOfMemory ? MRS_TG_ERR_NOMEM : MRS_TG_ERR_HIP, "hipHostMalloc(%zu) failed: %s", bytes,
                hipGetErrorString(e));
  }
  return MRS_TG_OK;
}

void mrs_tg_host_free(void* ptr) {
  if (ptr) (void)hipHostFree(ptr);
}

int mrs_tg_host_register(void* ptr, size_t bytes) {
  if (!ptr || !bytes) return fail(nullptr, MRS_TG_ERR_INVALID_ARG, "nothing to register");
  const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(nullptr, MRS_TG_ERR_HIP, "hipHostRegister failed: %s", hipGetErrorString(e));
  }
  return MRS_TG_OK;
}

int mrs_tg_host_unregister(void* ptr) {
  if (!ptr) return MRS_TG_OK;
  const hipError_t e = hipHostUnregister(ptr);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(nullptr, MRS_TG_ERR_HIP, "hipHostUnregister failed: %s", hipGetErrorString(e));
  }
  return MRS_TG_OK;
}

// ---- single path, findTrajectory()'s signature -------------------------------------------------


int mrs_tg_find_trajectory(mrs_tg_ctx* ctx, const mrs_tg_waypoint* wps, int32_t n_wp, const mrs_tg_initial_state* init,
                           const double* limits9, const mrs_tg_options* opt_in, int32_t relax_heading,
                           double* seg_times_out, double* coeffs_out, int32_t* status_out, int32_t* n_samples_out,
                           double* samples_out) {
  if (!ctx) return fail(nullptr, MRS_TG_ERR_INVALID_ARG, "ctx is NULL");
  if (!wps || !limits9 || !opt_in || !seg_times_out || !coeffs_out || !status_out || !n_samples_out)
    return fail(ctx, MRS_TG_ERR_INVALID_ARG, "NULL argument");
  if (n_wp < 2) return fail(ctx, MRS_TG_ERR_INVALID_ARG, "need at least 2 waypoints, got %d", n_wp);
  const int d = opt_in->derivative_to_optimize;
  if (d < 2 || d > 4) return fail(ctx, MRS_TG_ERR_INVALID_ARG, "derivative_to_optimize must be 2, 3 or 4");
  namespace pol = mrs_tg::policy;
  const int V = n_wp, S = n_wp - 1;
  ctx->find_rejection = MRS_TG_FIND_ACCEPTED;
  ctx->find_baca_total = 0.0;
  *n_samples_out = 0;
  try {
    std::vector<double> raw(V * 4), wp(V * 4), vals(V * 20), baca;
    std::vector<uint8_t> mask(V * 5), stop(V);
    for (int i = 0; i < V; ++i) {
      for (int k = 0; k < 4; ++k) raw[i * 4 + k] = wps[i].coords[k];
      stop[i] = wps[i].stop_at;
    }
    // vertices: src/mrs_trajectory_generation.cpp:923-977; limits: :985-1038
    pol::build_vertices(raw.data(), stop.data(), V, init, d, wp.data(), mask.data(), vals.data());
    double lim[9];
    pol::effective_limits(limits9, relax_heading != 0, lim);
    // initial_total_time_baca (:1048-1056), from the same vertices and limits the optimiser sees
    ctx->find_baca_total = pol::baca_total_time(S, wp.data(), lim, baca);
    mrs_tg_options opt = *opt_in;
    opt.estimate_times = 1;
    // the reference's own status rule: a runaway of the feasibility scaling keeps the outer loop's code and is discarded by
    // the length check below, as at :1178-1199 (a 5 cm path whose estimate is 0.025 s is no runaway: it is never checked)
    opt.flags |= MRS_TG_FLAG_REFERENCE_STATUS;
    const int32_t so[2] = {0, S};
    double cost = 0.0;
    int rc = mrs_tg_solve_batch(ctx, 1, so, wp.data(), mask.data(), vals.data(), lim, &opt, seg_times_out, coeffs_out,
                                status_out, &cost, n_samples_out, samples_out);
    if (rc != MRS_TG_OK) return rc;
  } catch (const std::bad_alloc&) {
    return fail(ctx, MRS_TG_ERR_NOMEM, "out of host memory for a path of %d waypoints", n_wp);
  }
  // accept >= 1 except MAXTIME(6), and -1 (src/mrs_trajectory_generation.cpp:1138-1149)
  if (!pol::code_accepted(*status_out)) {
    ctx->find_rejection = MRS_TG_FIND_REJECTED_CODE;
    ctx->last_error = "optimization failed with code " + std::to_string(*status_out);
    *n_samples_out = 0;
    return MRS_TG_OK;
  }
  // "validate the temporal sampling of the trajectory" (:1178-1199): states.size() * sampling_dt against the Baca total.  A
  // trajectory with more samples than samples_out holds is reported as capacity + 1: the count the check sees is then a
  // LOWER bound of the real one, enough to reject "too long" whenever capacity * dt exceeds the allowed length
  if (opt_in->sampling_dt > 0 && samples_out) {
    const int verdict = pol::length_check(*n_samples_out, opt_in->sampling_dt, ctx->find_baca_total,
                                          opt_in->max_trajectory_len_factor, opt_in->min_trajectory_len_factor);
    if (verdict != 0) {
      char msg[256];
      std::snprintf(msg, sizeof(msg), "trajectory sampling failed: the final trajectory sampling is too %s = %.2f, initial 'baca' "
                    "estimate = %.2f, allowed factor %.2f", verdict > 0 ? "long" : "short", *n_samples_out * opt_in->sampling_dt,
                    ctx->find_baca_total, verdict > 0 ? opt_in->max_trajectory_len_factor : opt_in->min_trajectory_len_factor);
      ctx->last_error = msg;
      ctx->find_rejection = verdict > 0 ? MRS_TG_FIND_REJECTED_TOO_LONG : MRS_TG_FIND_REJECTED_TOO_SHORT;
      *n_samples_out = 0;
    }
  }
  return MRS_TG_OK;
}

int mrs_tg_find_trajectory_info(const mrs_tg_ctx* ctx, int32_t* rejection_out, double* baca_total_time_out) {
  if (!ctx) return fail(nullptr, MRS_TG_ERR_INVALID_ARG, "ctx is NULL");
  if (rejection_out) *rejection_out = ctx->find_rejection;
  if (baca_total_time_out) *baca_total_time_out = ctx->find_baca_total;
  return MRS_TG_OK;
}

}  // extern "C"
