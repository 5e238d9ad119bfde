// mrs_tg_launch.h -- host-visible view of a batch and the kernel launchers (internal to the library).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstddef>
#include <cstdint>

#include "../../include/mrs_tg.h"

struct mrs_tg_ctx;

namespace mrs_tg {

// Per-dispatch timing (mrs_tg_set_profiling): the next launch of the kernel family a ProfileScope names carries these
// events on the launch itself (hipExtLaunchKernelGGL), so their difference is the dispatch's own start-to-end time.
struct KernelTimer {
  hipEvent_t start = nullptr, stop = nullptr;
};
// Hint of the calling thread's current ABI call (MRS_TG_FLAG_SHARED_DEVICE): other batches are in flight on the device, so
// launch shapes that leave wavefront slots free are preferred over the lowest latency of this launch alone.
void set_shared_device_hint(bool on);
bool shared_device_hint();
// Hint of the calling thread's current ABI call (MRS_TG_FLAG_CONSTRAINED_SLOTS): the batch may hold vertices with constrained
// derivative slots beside its paths' ends; min-snap launches then use the instantiations that take such vertices
void set_constrained_slots_hint(bool on);
bool constrained_slots_hint();
// Hint of the calling thread's current ABI call: some paths start from a moving state (non-zero constrained derivatives at
// their first vertex).  Set by mrs_tg_solve_batch from the values it holds in host memory (a device-resident caller has no
// flag for it); read by launch_nonlinear's choice between the plan's dimension split and lane groups
void set_moving_starts_hint(bool on);
bool moving_starts_hint();
KernelTimer take_kernel_timer();  // the pending pair (null events when nothing is pending); consumed by the call
void set_kernel_timer(hipEvent_t start, hipEvent_t stop);  // arms the next timed launch of this thread
// Kernel trace (mrs_tg_kernel_trace): every launch of the library notes its kernel's name in a small per-thread ring, so
// that a test or the benchmark can SAY which kernels a call ran instead of inferring it from batch sizes (one pointer store)
void note_kernel(const char* name);
void kernel_trace_reset();
int kernel_trace(const char** names_out, int capacity);  // oldest first; at most the newest 32 since the reset
// Dry run (mrs_tg_plan_explain): the calling thread's launchers run their routing -- every size rule, environment knob and
// hint exactly as in a real call -- and NOTE the kernels they would launch, but enqueue nothing
bool dry_run();
void set_dry_run(bool on);
// launch with the pending timer, if any
#define MRS_TG_LAUNCH_TIMED(kernel, grid, block, lds, stream, ...)                                           \
  do {                                                                                                       \
    const ::mrs_tg::KernelTimer kt__ = ::mrs_tg::take_kernel_timer();                                        \
    ::mrs_tg::note_kernel(#kernel);                                                                          \
    if (!::mrs_tg::dry_run())                                                                                \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, kt__.start, kt__.stop, 0, __VA_ARGS__);        \
  } while (0)
// the same for a kernel template of two arguments (a comma inside a macro argument needs parentheses, which would end up in the
// noted name): MRS_TG_LAUNCH_TIMED_T2(kernel, A, B, grid, ...) launches kernel<A, B> and notes "kernel<A, B>"
#define MRS_TG_LAUNCH_TIMED_T2(kernel, A, B, grid, block, lds, stream, ...)                                        \
  do {                                                                                                             \
    const ::mrs_tg::KernelTimer kt__ = ::mrs_tg::take_kernel_timer();                                              \
    ::mrs_tg::note_kernel(#kernel "<" #A ", " #B ">");                                                             \
    if (!::mrs_tg::dry_run())                                                                                      \
      hipExtLaunchKernelGGL((kernel<A, B>), grid, block, lds, stream, kt__.start, kt__.stop, 0, __VA_ARGS__);      \
  } while (0)
// plain launches, and launches that carry the events of a multi-kernel timing themselves
#define MRS_TG_LAUNCH(kernel, ...)                \
  do {                                            \
    ::mrs_tg::note_kernel(#kernel);               \
    if (!::mrs_tg::dry_run()) hipLaunchKernelGGL(kernel, __VA_ARGS__);      \
  } while (0)
#define MRS_TG_LAUNCH_EXT(kernel, ...)            \
  do {                                            \
    ::mrs_tg::note_kernel(#kernel);               \
    if (!::mrs_tg::dry_run()) hipExtLaunchKernelGGL(kernel, __VA_ARGS__);   \
  } while (0)

// records `message` as the context's (and the global) last error and returns `code`
int report_error(mrs_tg_ctx* ctx, int code, const char* fmt, ...);
// pinned host scratch owned by the context (grown on demand, kept across calls, freed with the context); null when the
// runtime refuses the allocation
void* ctx_host_scratch(mrs_tg_ctx* ctx, size_t bytes);
// mrs_tg_solve_batch for a caller that reads only times, statuses and samples: the coefficients are not downloaded
int solve_batch_samples_only(mrs_tg_ctx* ctx, int32_t n_paths, const int32_t* seg_offsets, const double* waypoints,
                             const uint8_t* fixed_mask, const double* fixed_values, const double* limits,
                             const mrs_tg_options* opt, double* seg_times_inout, int32_t* status_out, int32_t* n_samples_out,
                             double* samples_out);

// Device-resident structure of a batch (built once per plan).
struct BatchView {
  int n_paths;
  int n_segments;            // sum of S over the batch
  int max_segments;          // largest S
  int uniform_S;             // S if every path has the same segment count, else 0
  const int32_t* seg_offsets;  // [n_paths + 1] CSR over segments (caller's path order)
  const int32_t* order;        // [n_paths] position q -> path index, sorted by S descending (stable)
  const int32_t* slot_start;   // [max_segments + 1] slot_start[j] = number of (q, j') pairs with j' < j
};

hipError_t launch_assemble(const BatchView& b, int d, const double* seg_times, double* H, double* Ainv,
                           hipStream_t stream);
hipError_t launch_solve_linear(const BatchView& b, int d, bool fused, const uint8_t* mask, const double* vals,
                               const double* seg_times, const double* H, const double* Ainv, double* ws,
                               double* coeffs, int32_t* status, double* cost, const int32_t* status_in,
                               hipStream_t stream, const double* pos_wp = nullptr);
// Up to kCopyMax flat copies in ONE launch: how mrs_tg_solve_batch moves arrays between pinned host memory (which the
// GPU addresses directly) and the device -- a kernel launch costs the host ~3 us, a hipMemcpyAsync 10-25 us.
constexpr int kCopyMax = 8;
struct CopyList {
  const void* src[kCopyMax];
  void* dst[kCopyMax];
  unsigned long long bytes[kCopyMax];
  int n = 0;
  void add(const void* s, void* d, size_t b) {
    if (b == 0 || s == nullptr || d == nullptr) return;
    src[n] = s;
    dst[n] = d;
    bytes[n] = b;
    ++n;
  }
};
hipError_t launch_copy_many(const CopyList& cl, hipStream_t stream);
// samples [n_paths][capacity][4]: only the min(n_samples[p], capacity) rows a path has produced are copied (a sample buffer
// is sized for the longest trajectory the caller would accept; a typical one uses a fraction of it)
hipError_t launch_copy_samples(const double* src, double* dst, const int32_t* n_samples, int n_paths, int capacity,
                               hipStream_t stream);

hipError_t launch_estimate_times(const BatchView& b, const double* wp, const double* limits, double* seg_times,
                                 hipStream_t stream);
// (large launches: 8 or 16 lanes per path, 64 / G paths per wavefront -- sample_group_kernel; else one wavefront per path)
hipError_t launch_sample(const BatchView& b, const double* coeffs, const double* seg_times, double dt, int capacity,
                         int32_t* n_samples, double* samples, hipStream_t stream);
int sample_group_lanes(const BatchView& b);  // 0 | 8 | 16: which sampler a launch of this batch takes
// the same walk, every sample with its derivative orders 0..4: states [n_paths][capacity][kSampleStateOrders][4]
constexpr int kSampleStateOrders = 5;
hipError_t launch_sample_states(const BatchView& b, const double* coeffs, const double* seg_times, double dt, int capacity,
                                int32_t* n_samples, double* states, hipStream_t stream);
// ---- the policy layer's per-round device work (mrs_tg_policy_dev.hip) ----
// constraint mask [n_vertices][5] and values [n_vertices][5][4] from the unwrapped waypoints [n_vertices][4], vinfo[v] = position of
// the vertex's path in the round's batch << 4 | flags, and the initial states [n_paths][12] (velocity, acceleration, jerk)
hipError_t launch_policy_expand(int n_vertices, int d, const double* wp, const int32_t* vinfo, const double* init, uint8_t* mask,
                                double* vals, hipStream_t stream);
struct PolicyValidateArgs {
  int n_paths;
  const int32_t* seg_offsets;   // [n_paths + 1] (device)
  const double* wp;             // [sum V][4]
  const double* samples;        // [n_paths][capacity][4]
  const int32_t* n_samples;     // [n_paths] as the solve reported them
  const int32_t* status;        // [n_paths]
  const double* baca_total;     // [n_paths] initial_total_time_baca
  double dt, max_len_factor, min_len_factor, max_deviation;
  int capacity, first_segment, check_enabled, last_round;
  // results (any address the device can write: pinned host memory)
  int32_t* ok_out;
  int32_t* ns_out;
  int32_t* status_out;
  double* max_dev_out;
  uint8_t* is_safe_out;
  uint8_t* safe_out;            // [sum S]
  int32_t* ns_copy;             // [n_paths] (device) rows of the finished paths' samples that travel
};
hipError_t launch_policy_validate(const PolicyValidateArgs& args, hipStream_t stream);

size_t linear_workspace_doubles(const BatchView& b);
// MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS, checked: the number of vertices whose position is unconstrained or whose constrained
// position differs (bitwise) from its waypoint.  Blocks until the count is on the host.
hipError_t count_position_mismatches(const BatchView& b, const double* wp, const uint8_t* mask, const double* vals,
                                     hipStream_t stream, long long* count_out);
// any fixed / free pattern, position-free vertices included (mrs_tg_general.hip): solves the paths flagged in `only` (by
// path, non-zero) or, without it, the paths whose status is -2; status out = 1 or opt_status' stopping reason
size_t general_workspace_doubles(const BatchView& b);
hipError_t launch_solve_general(const BatchView& b, int d, const uint8_t* mask, const double* vals, const double* seg_times,
                                double* ws, double* coeffs, int32_t* status, double* cost, hipStream_t stream,
                                const int32_t* only = nullptr, const int32_t* opt_status = nullptr);
// phase-split tile kernel (mrs_tg_tile.hip): small and medium batches whose per-path state fits in LDS
bool tile_kernel_applies(const BatchView& b, bool fused);
hipError_t launch_solve_tile(const BatchView& b, int d, bool fused, const uint8_t* mask, const double* vals,
                             const double* seg_times, const double* H, const double* Ainv, double* coeffs,
                             int32_t* status, double* cost, const int32_t* status_in, hipStream_t stream);

// What the rows kernel can do around its solve for the same path, in the same launch:
//   before: the feasibility scaling of the segment times (scaleSegmentTimesWithViolation's T_i <- s_i T_i from the
//           per-segment maxima, written back to seg_times) -- the last stage before the final solve of the Mellinger pipeline;
//   after:  the sampling of the solved trajectory (sampleWholeTrajectory).
// All-null / zero = the plain solve.
struct RowsTail {
  const double* maxima = nullptr;        // [n_segments][9]; with limits and opt_status: scale the times first
  bool maxima_in_launch = false;         // instead: solve at the incoming times, take the maxima of THAT trajectory, scale, solve
                                         // again (the closing stages of a pipeline in one launch; rows_pipeline_applies)
  const double* limits = nullptr;        // [n_paths][9]
  const int32_t* opt_status = nullptr;   // paths whose search was refused (-2) keep their times
  const double* sum_t0 = nullptr;        // [n_paths] total time the outer loop started from: the runaway test (mrs_tg.h)
  double* seg_times_out = nullptr;       // the (scaled) times are written here (the caller's seg_times)
  double sampling_dt = 0.0;              // > 0: sample
  int sample_capacity = 0;
  int32_t* n_samples = nullptr;
  double* samples = nullptr;
  const double* sample_acc = nullptr;    // the walk's accumulated times (sample_acc_table): filled in by launch_solve_rows
  int sample_acc_n = 0;
  const double* pos_wp = nullptr;        // MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS: the compact [vertex][4] array the saturated-device
                                         // solve reads vertex positions from (the other kernels read the value array)
};

// A[k] = k additions of dt to 0, the accumulated time of the reference's sampling walk, on the current device: at least
// capacity + 80 entries, built by a kernel on `stream` the first time a (device, dt) pair is seen and ordered behind that
// build for launches on other streams; a bounded, least-recently-used cache (mrs_tg_kernels.hip)
// `pin` keeps the table from being recycled between this call's return and the enqueue of the kernel that reads it (another
// host thread may retire the entry in that window -- a new dt evicting the least recently used one, a larger capacity
// outgrowing it -- and a drain of the retired list would hand the block to someone else): the caller declares an AccPin
// before the call and lets it go out of scope behind its launch.
struct AccPin {
  void* block = nullptr;
  AccPin() = default;
  AccPin(const AccPin&) = delete;
  AccPin& operator=(const AccPin&) = delete;
  ~AccPin();
};
hipError_t sample_acc_table(double dt, int capacity, hipStream_t stream, const double** table_out, int* n_out, AccPin* pin);
void sample_tables_release();  // frees every table (with the last context of the process)

// one lane per unknown, no materialised blocks (mrs_tg_rows.hip): the fused linear solve of every path that fits its LDS record
bool rows_kernel_applies(const BatchView& b, bool with_sampling = false);
// sampling on the solve's launch pays while a wavefront holds one path (small batches: one launch and one staging pass
// less, 1024 x 10 nonlinear 138 -> 132 us); with two paths per wavefront the walks of the two run one after the other and
// the separate sampler (one wavefront per path) is faster (8192 x 10: 483 vs 548 us)
bool rows_tail_sampling_pays(const BatchView& b);
// solve -> maxima -> scaling -> solve -> sampling of a pipeline in one launch of the rows kernel (small batches)
bool rows_pipeline_applies(const BatchView& b);
hipError_t launch_solve_rows(const BatchView& b, int d, const uint8_t* mask, const double* vals, const double* seg_times,
                             double* coeffs, int32_t* status, double* cost, const int32_t* status_in, hipStream_t stream,
                             const RowsTail& tail = RowsTail());

// up to kRowsGroupMax batches of one plan solved by one launch (fixed-times default solve): per batch its arrays
constexpr int kRowsGroupMax = 16;
struct RowsGroup {
  const uint8_t* mask[kRowsGroupMax];
  const double* vals[kRowsGroupMax];
  const double* seg_times[kRowsGroupMax];
  double* coeffs[kRowsGroupMax];
  int32_t* status[kRowsGroupMax];
  double* cost[kRowsGroupMax];
  const double* pos_wp[kRowsGroupMax];  // per batch: waypoints under MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS, else nullptr
  int n = 0;
};
hipError_t launch_solve_rows_group(const BatchView& b, int d, const RowsGroup& g, hipStream_t stream);

// four lanes per path, factors in LDS (mrs_tg_quad.hip): the fixed-times solve of launches that carry more paths than the rows
// kernel has wavefront slots for (paths_in_launch: of all batches a grouped launch carries).  ws: the plan's global factor
// store (linear_workspace_doubles per batch), used by wavefronts whose paths need the general masked step.  tail: the
// feasibility scaling of a Mellinger pipeline's last solve (no sampling on this launch).
bool quad_kernel_applies(const BatchView& b, long long paths_in_launch, bool with_sampling);
hipError_t launch_solve_quad(const BatchView& b, int d, const uint8_t* mask, const double* vals, const double* seg_times,
                             double* coeffs, int32_t* status, double* cost, const int32_t* status_in, double* ws,
                             hipStream_t stream, const RowsTail& tail = RowsTail());
hipError_t launch_solve_quad_group(const BatchView& b, int d, const RowsGroup& g, double* ws, hipStream_t stream);

}  // namespace mrs_tg
