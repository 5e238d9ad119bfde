/*
 * mrs_tg.h -- C ABI of the MI355X-native batched polynomial trajectory optimiser.
 *
 * Drop-in boundary for the numerical core of ctu-mrs/mrs_uav_trajectory_generation:
 * everything MrsTrajectoryGeneration::findTrajectory() does between building its vertices and
 * receiving the sampled states (/root/reference/src/mrs_trajectory_generation.cpp:1046-1169),
 * for a whole batch of independent paths at once.  The reference has no FFI for this path; the
 * seam is a C++ call sequence, so each entry point names the reference calls it replaces
 * (paths relative to /root/reference/, "linear_impl.h" / "nonlinear_impl.h" are
 * include/eth_trajectory_generation/impl/polynomial_optimization_{linear,nonlinear}_impl.h).
 *
 * Conventions
 *   - plain C types, caller-owned buffers, no exceptions cross the boundary;
 *   - every function returns MRS_TG_OK (0) or a negative MRS_TG_ERR_* and records a message
 *     retrievable with mrs_tg_last_error();
 *   - 4 dimensions (x, y, z, heading), 10 coefficients per polynomial in ascending powers
 *     (include/eth_trajectory_generation/polynomial.h:35-37), IEEE double throughout;
 *   - a batch is CSR over segments: path p owns segments [seg_offsets[p], seg_offsets[p+1]) and
 *     vertices [seg_offsets[p] + p, seg_offsets[p+1] + p + 1);
 *   - per-path results carry an nlopt-style status (src/mrs_trajectory_generation.cpp:1138-1149
 *     accepts >= 1 except 6, and -1).
 *   - there is NO CPU fallback: without a usable HIP device every call fails with
 *     MRS_TG_ERR_NO_DEVICE.
 */
#ifndef MRS_TG_H_
#define MRS_TG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRS_TG_ABI_VERSION 5
#define MRS_TG_N_COEFF 10
#define MRS_TG_N_DIM 4
#define MRS_TG_N_SLOT 5 /* derivative slots per vertex: position .. snap */
#define MRS_TG_MAX_SEGMENTS 256 /* longest path a plan accepts (per-path optimiser state lives in the 160 KB LDS of a CU;
                                   the reference's deviation loop ends near 30 segments) */

enum {
  MRS_TG_OK = 0,
  MRS_TG_ERR_INVALID_ARG = -1,
  MRS_TG_ERR_NO_DEVICE = -2,
  MRS_TG_ERR_HIP = -3,
  MRS_TG_ERR_UNSUPPORTED = -4,
  MRS_TG_ERR_NOMEM = -5
};

/* per-path status values (nlopt.h result codes, as consumed by the nodelet) */
enum {
  MRS_TG_STATUS_FAILURE = -1,
  MRS_TG_STATUS_INVALID_ARGS = -2, /* also: a vertex of the path leaves its POSITION unconstrained and the general solver
                                      was not asked for.  Every caller of the reference constrains the position of every
                                      vertex (src/...cpp:944, 963, 967) and the fast kernels rely on it; the general
                                      fixed / free patterns of setupConstraintReorderingMatrix (linear_impl.h:184-257)
                                      are solved under MRS_TG_FLAG_GENERAL_PATTERNS (every mode) */
  MRS_TG_STATUS_ROUNDOFF_LIMITED = -4, /* nlopt::ROUNDOFF_LIMITED, which the nodelet rejects (:1103-1106, 1146-1149).  Mellinger
                                      mode: the feasibility scaling that follows the outer loop has multiplied the path's
                                      total time by more than MRS_TG_RUNAWAY_TIME_FACTOR -- the outer loop ended on a point
                                      with a segment on the 0.01 s bound next to seconds-long neighbours, where the linear
                                      solve has a condition number of (T_max / T_min)^7 and its maxima are rounding noise
                                      (DESIGN.md section 5).  The arrays hold what was computed; it is not a trajectory
                                      to fly.  The reference returns such a path with MAXEVAL_REACHED and leaves it to the
                                      nodelet's length check (:1178-1199) to discard it */
  MRS_TG_STATUS_SUCCESS = 1,
  MRS_TG_STATUS_FTOL_REACHED = 3,
  MRS_TG_STATUS_XTOL_REACHED = 4,
  MRS_TG_STATUS_MAXEVAL_REACHED = 5,
  MRS_TG_STATUS_MAXTIME_REACHED = 6
};

/* sum of the final segment times / sum of the times the outer loop started from, above which a Mellinger result is
 * reported as MRS_TG_STATUS_ROUNDOFF_LIMITED.  Healthy paths: median 1.6, 99.9 % below 3, the largest of 8192 paths with
 * limits scaled by 0.3 ... 3 and mixed constraint patterns 18.6; runaways: 30 ... 1e10 (oracle, tests/test_oracle_runaway.py) */
#define MRS_TG_RUNAWAY_TIME_FACTOR 25.0

/* time_alloc_method (NonlinearOptimizationParameters::TimeAllocMethod,
 * include/eth_trajectory_generation/polynomial_optimization_nonlinear.h:92-100) */
enum {
  MRS_TG_TIME_ALLOC_NONE = -1,         /* fixed segment times: PolynomialOptimization::solveLinear only */
  MRS_TG_TIME_ALLOC_SQUARED_TIME = 0,  /* kSquaredTime: J_d + time_penalty (sum T)^2 + soft constraints, gradient-free */
  MRS_TG_TIME_ALLOC_RICHTER_TIME = 1,  /* kRichterTime: J_d + time_penalty sum T + soft constraints, gradient-free */
  MRS_TG_TIME_ALLOC_MELLINGER = 2,     /* kMellingerOuterLoop, the shipping default
                                          (config/private/trajectory_generation.yaml:7) */
  MRS_TG_TIME_ALLOC_SQUARED_TIME_AND_CONSTRAINTS = 3, /* kSquaredTimeAndConstraints: as 0, segment times and free
                                          end-point derivatives are the variables (nonlinear_impl.h:429-536) */
  MRS_TG_TIME_ALLOC_RICHTER_TIME_AND_CONSTRAINTS = 4  /* kRichterTimeAndConstraints: as 1, same variables */
};

enum {
  MRS_TG_FLAG_FUSED_ASSEMBLY = 1,     /* the default since ABI 2 (kept so that ABI-1 callers still say what they mean): every
                                         lane of the solve kernel forms its column of the reduced system straight from the
                                         segment times; no block is written to memory */
  MRS_TG_FLAG_MATERIALIZED_BLOCKS = 2,/* linear mode: run the assembly kernel (mrs_tg_plan_assemble: full H_i and A_i^-1 of
                                         every segment in HBM, the reference's updateSegmentTimes + constructR products) and
                                         solve from the materialised blocks */
  MRS_TG_FLAG_SHARED_DEVICE = 4,      /* a hint, results are unaffected: the caller keeps several batches in flight on this
                                         device (one context + stream each), so small batches are launched in shapes that
                                         leave wavefront slots to the other streams instead of minimising the latency of
                                         this one launch */
  MRS_TG_FLAG_GENERAL_PATTERNS = 16,  /* some vertices may leave their POSITION free (setupFromVertices takes any fixed / free
                                         pattern, linear_impl.h:184-257; the nodelet never builds such a vertex).  The fast
                                         kernels return those paths with status -2; with this flag they take a general route
                                         behind the fast kernels, in every time_alloc_method: 5 x 5 vertex blocks
                                         (mrs_tg_general.hip) for every linear solve of the pipeline, and in Mellinger mode
                                         the outer loop with that solve as its evaluation (optimize_general_kernel).  The
                                         other paths of the batch are the fast kernels' results, bit for bit.
                                         mrs_tg_solve_batch sets the flag by itself when its host copy of fixed_mask shows
                                         such a vertex; callers of the device-pointer interface say so */
  MRS_TG_FLAG_CAREFUL_COST = 8        /* Mellinger mode: paths on which a trial point's cost lost its digits in the fast
                                         evaluation (a segment on the 0.01 s bound next to long neighbours; about 0.3 % of
                                         random 10-segment paths) are run again with the cost the reference computes,
                                         0.5 c^T Q c from the coefficients (computeCost, linear_impl.h:128-141), in every
                                         evaluation.  One more kernel per call, about the duration of the outer loop
                                         itself; without the flag such a trial point is rejected where the reference may
                                         accept it.  The re-run kernel is a compile-time option
                                         (MRS_TG_WITH_CAREFUL, ON in the shipped library since ABI 4; it moves 65536 x 10
                                         from 99.9435 % to 99.9481 % agreement with the oracle): a library built without it
                                         (-DMRS_TG_WITH_CAREFUL=0) refuses the flag with MRS_TG_ERR_UNSUPPORTED, and
                                         mrs_tg_capabilities() says which one is loaded */
};

enum {
  MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS = 32 /* (ABI 4) the caller states that the position constraint of EVERY vertex is its
                                         waypoint: fixed_mask[v][0] != 0 and fixed_values[v][0][:] == waypoints[v][:], bit for
                                         bit -- what every vertex findTrajectory builds looks like
                                         (src/mrs_trajectory_generation.cpp:944, 963, 967: addConstraint(POSITION,
                                         waypoint.coords)).  `waypoints` must then be given, and kernels may read vertex
                                         positions from that compact [vertex][4] array instead of from 8 bytes out of every
                                         160 of fixed_values: the saturated-device solve (launches of >= 6144 paths) moves
                                         1.03 instead of 1.34 times its compulsory bytes.  Results are bit-identical.
                                         mrs_tg_plan_bind_solve CHECKS the statement once, on the arrays as they are at bind
                                         time, and refuses the bind with MRS_TG_ERR_INVALID_ARG if it does not hold;
                                         mrs_tg_plan_solve trusts it: the flag only changes launches that take the
                                         saturated-device kernel (>= 6144 paths per dispatch), so a statement that is false -- or
                                         has become false because the arrays of a bound solve were rewritten in place -- gives the
                                         solution of the waypoints' problem there and of the caller's problem on smaller launches.
                                         With MRS_TG_VERIFY_FLAGS=1 in the environment every mrs_tg_plan_solve re-checks the
                                         statement (a blocking check, for debugging and tests) and fails with
                                         MRS_TG_ERR_INVALID_ARG when it does not hold.  Other kernels ignore the flag */
  ,
  MRS_TG_FLAG_CONSTRAINED_SLOTS = 64  /* a HINT (results agree to rounding with and without it): beside the ends of its paths the
                                         batch may hold vertices with derivative slots constrained to zero -- stop_at waypoints --
                                         and the objective order is snap.  The min-snap launches of the large batches' kernels are
                                         compiled for position-only interior vertices (they are the benchmark configs' kernels) and
                                         hand other paths to the general steps; with the hint they run the instantiations that
                                         eliminate such vertices inside the specialised sweeps (always the case below snap).
                                         mrs_tg_solve_batch, mrs_tg_find_trajectory and mrs_tg_optimize_paths set it themselves
                                         from the masks they hold in host memory */
  ,
  MRS_TG_FLAG_REFERENCE_STATUS = 128  /* (ABI 5) Mellinger mode: the path's status is the outer loop's own stopping reason, as in
                                         the reference -- the product's runaway rule (MRS_TG_STATUS_ROUNDOFF_LIMITED when the
                                         feasibility scaling multiplied the total time by more than MRS_TG_RUNAWAY_TIME_FACTOR)
                                         is switched off.  For callers that apply the reference's own answer to a runaway, the
                                         length check against the Baca estimate (src/...cpp:1178-1199): mrs_tg_find_trajectory and
                                         mrs_tg_optimize_paths set the flag themselves.  The rule measures against the Euclidean
                                         estimate, which is 0.01-0.03 s for waypoints a few centimetres apart: such a path is
                                         stretched 25-fold by a perfectly healthy scaling, and the reference never checks a
                                         trajectory shorter than one second.  Coefficients, times and samples are the same bits
                                         with and without the flag */
};

/* mrs_tg_capabilities(): what this build of the library contains beyond the mandatory surface */
enum {
  MRS_TG_CAP_CAREFUL_COST = 1 /* MRS_TG_FLAG_CAREFUL_COST is honoured (optimize_careful_kernel is built in) */
};

typedef struct mrs_tg_options {
  int32_t derivative_to_optimize; /* 2 acceleration, 3 jerk, 4 snap (src/...cpp:904-919) */
  int32_t time_alloc_method;      /* MRS_TG_TIME_ALLOC_* */
  int32_t estimate_times;         /* != 0: initial times from estimateSegmentTimes (src/...cpp:1046,
                                     vertex.cpp:491-565); 0: use seg_times_inout as given */
  int32_t max_iterations;         /* nlopt maxeval (nonlinear_impl.h:73; param max_iterations) */
  double f_rel, f_abs;            /* nlopt ftol (src/...cpp:884; nonlinear.h:42-46) */
  double x_rel, x_abs;            /* nlopt xtol (src/...cpp:885; nonlinear.h:48-54) */
  double sampling_dt;             /* > 0: sample the result (sampleWholeTrajectory, src/...cpp:1169) */
  int32_t sample_capacity;        /* samples_out holds this many samples per path */
  int32_t flags;                  /* MRS_TG_FLAG_* */
  /* gradient-free modes 0 / 1 / 3 / 4 only (objectiveFunctionTime[AndConstraints], nonlinear_impl.h:568-614, 651-722) */
  double time_penalty;            /* param time_penalty (config/private/trajectory_generation.yaml:4) */
  double soft_constraint_weight;  /* param soft_constraints_weight (:6) */
  int32_t use_soft_constraints;   /* param soft_constraints_enabled (:5) */
  int32_t reserved_;
  double initial_stepsize_rel;    /* 0.1 (src/...cpp:893) */
  double max_time_s;              /* nlopt maxtime (src/...cpp:899: 2 * 0.95 * timeLeft()); <= 0: none.  The time-allocation
                                     search of a path that is still running when the budget has passed stops at its last
                                     evaluated point with MRS_TG_STATUS_MAXTIME_REACHED (checked once per objective
                                     evaluation against the device's constant-rate clock; the budget starts when the
                                     search kernel starts, for mrs_tg_solve_batch minus the host time already spent in
                                     the call) */
  /* (ABI 5) mrs_tg_find_trajectory only -- the temporal sanity check of findTrajectory (src/...cpp:1178-1199): a sampled
   * trajectory longer than one second whose length n_samples * sampling_dt exceeds max_trajectory_len_factor times, or falls
   * below min_trajectory_len_factor times, the path's Baca estimate (estimateSegmentTimesBaca summed, :1048-1056) is
   * discarded.  Defaults 3.0 / 0.33 (config/public/trajectory_generation.yaml:35-36); <= 0 switches that side off.  The
   * batched solve calls ignore both (they are handed vertices, not a path); mrs_tg_optimize_paths uses the pair in
   * mrs_tg_policy_options */
  double max_trajectory_len_factor, min_trajectory_len_factor;
} mrs_tg_options;

typedef struct mrs_tg_ctx mrs_tg_ctx;
typedef struct mrs_tg_plan mrs_tg_plan;

/* ---- context ------------------------------------------------------------------------------- */

/* Bind a context to HIP device `device_ordinal`.  One context per thread/stream; re-entrant
 * across contexts.  Replaces the construction of PolynomialOptimizationNonLinear<10>
 * (src/mrs_trajectory_generation.cpp:1064). */
int mrs_tg_create(int device_ordinal, mrs_tg_ctx** ctx_out);
void mrs_tg_destroy(mrs_tg_ctx* ctx);
const char* mrs_tg_last_error(const mrs_tg_ctx* ctx); /* ctx may be NULL: last global error */
int mrs_tg_abi_version(void);
int mrs_tg_capabilities(void); /* MRS_TG_CAP_* bits (ABI 4) */
void mrs_tg_default_options(mrs_tg_options* opt);

/* Launch on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) instead of
 * the context's own stream.  NULL means HIP's null (default) stream, which is what torch uses unless a
 * side stream is current.  mrs_tg_reset_stream() goes back to the context's own stream. */
int mrs_tg_set_stream(mrs_tg_ctx* ctx, void* hip_stream);
int mrs_tg_reset_stream(mrs_tg_ctx* ctx);
int mrs_tg_synchronize(mrs_tg_ctx* ctx);

/* ---- several devices -------------------------------------------------------------------------- */

/* A set of contexts, one per entry of `device_ordinals` (an ordinal may repeat: two contexts on one GPU), for
 * mrs_tg_multi_solve_batch.  The reference handles one path per request on one worker thread
 * (src/mrs_trajectory_generation.cpp:1064-1083, 1513); paths are independent, so a batch shards over the devices with no
 * exchange between them. */
typedef struct mrs_tg_multi mrs_tg_multi;
int mrs_tg_create_multi(const int* device_ordinals, int n_devices, mrs_tg_multi** multi_out);
void mrs_tg_destroy_multi(mrs_tg_multi* multi);
int mrs_tg_multi_n_devices(const mrs_tg_multi* multi);
mrs_tg_ctx* mrs_tg_multi_context(mrs_tg_multi* multi, int index); /* the context of device `index` (owned by `multi`) */
/* Which device solves which path: shard_out[p] in [0, n_devices).  Uniform batches are cut into contiguous ranges whose
 * sizes differ by at most one; ragged batches are balanced on the segment count (longest path first onto the least
 * loaded device). */
int mrs_tg_multi_shard(const mrs_tg_multi* multi, int32_t n_paths, const int32_t* seg_offsets, int32_t* shard_out);
/* mrs_tg_solve_batch over all devices of `multi`: same arguments, same results (every path is solved by exactly the
 * kernels a single-device call would run on it).  One host thread per device solves its shard from / into the caller's
 * buffers; returns the first error of any shard (mrs_tg_multi_last_error). */
int mrs_tg_multi_solve_batch(mrs_tg_multi* multi, int32_t n_paths, const int32_t* seg_offsets, const double* waypoints,
                             const uint8_t* fixed_mask, const double* fixed_values, const double* limits,
                             const mrs_tg_options* opt, double* seg_times_inout, double* coeffs_out, int32_t* status_out,
                             double* cost_out, int32_t* n_samples_out, double* samples_out);
const char* mrs_tg_multi_last_error(const mrs_tg_multi* multi);

/* ---- one-call host interface ---------------------------------------------------------------- */

/* Host buffers in, host buffers out; blocking.  A batch of one path reproduces the nodelet's call.
 * Replaces: estimateSegmentTimes (src/...cpp:1046), setupFromVertices (:1065; linear_impl.h:62-106),
 * the 12 addMaximumMagnitudeConstraint calls (:1067-1081, folded into `limits`), optimize() (:1083;
 * nonlinear_impl.h:90-234,336-408), getTrajectory (:1163) and sampleWholeTrajectory (:1169).
 *
 *   waypoints      [sum V][4]      x, y, z, heading (already unwrapped as at src/...cpp:935)
 *   fixed_mask     [sum V][5]      != 0: derivative k of the vertex is constrained (Vertex::addConstraint)
 *   fixed_values   [sum V][5][4]   the constrained values (ignored where the mask is 0)
 *   limits         [n_paths][9]    {v,a,j} x {horizontal, vertical, heading}: index 3*(k-1)+group;
 *                                  a heading limit >= FLT_MAX means relax_heading (src/...cpp:1030-1038)
 *   seg_times_inout[sum S]         in: segment times (unless estimate_times); out: final times
 *   coeffs_out     [sum S][4][10]
 *   status_out     [n_paths]       nlopt-style code;  cost_out [n_paths] J_d (computeCost) -- may be NULL
 *   n_samples_out  [n_paths], samples_out [n_paths][sample_capacity][4] (x, y, z, heading wrapped to
 *                  (-pi, pi] as the nodelet reads it, src/...cpp:1582-1599) -- may be NULL when
 *                  sampling_dt <= 0.  n_samples_out reports the count the reference would produce when it
 *                  fits; a longer trajectory is reported as sample_capacity + 1 (only the first
 *                  sample_capacity samples are written) -- size the capacity from the length the caller
 *                  would still accept (the nodelet's max_trajectory_len_factor check, src/...cpp:1178-1186).
 */
int mrs_tg_solve_batch(mrs_tg_ctx* ctx, int32_t n_paths, const int32_t* seg_offsets, const double* waypoints,
                       const uint8_t* fixed_mask, const double* fixed_values, const double* limits,
                       const mrs_tg_options* opt, double* seg_times_inout, double* coeffs_out, int32_t* status_out,
                       double* cost_out, int32_t* n_samples_out, double* samples_out);

/* Pinned host memory for the arrays of mrs_tg_solve_batch, mrs_tg_multi_solve_batch and mrs_tg_find_trajectory.  The call
 * keeps one device block and one pinned staging block per context and moves each array the cheapest way its location
 * allows: an array in pinned memory (from mrs_tg_host_alloc, registered with mrs_tg_host_register, or any block of
 * hipHostMalloc) is read / written by the DMA engines in place; small pageable arrays are packed into the staging block
 * and travel as ONE copy per direction; large pageable arrays go through hipMemcpyAsync on the caller's pages.  A host that
 * calls in a loop (the nodelet's worker: std::vector outputs it re-uses) gets the PCIe rate by allocating its in / out
 * arrays here once.  mrs_tg_host_register pins an existing allocation until mrs_tg_host_unregister: the caller must
 * unregister before it frees or reallocates the block. */
int mrs_tg_host_alloc(size_t bytes, void** ptr_out);
void mrs_tg_host_free(void* ptr);
int mrs_tg_host_register(void* ptr, size_t bytes);
int mrs_tg_host_unregister(void* ptr);

/* ---- plan interface: analysis once, device-resident data, asynchronous ----------------------- */

/* Analyse the batch structure (host seg_offsets): sorts paths by segment count, sizes the
 * workspace, uploads the CSR structure.  Corresponds to the structural half of setupFromVertices
 * (setupConstraintReorderingMatrix, linear_impl.h:184-257). */
int mrs_tg_plan_create(mrs_tg_ctx* ctx, int32_t n_paths, const int32_t* seg_offsets_host, mrs_tg_plan** plan_out);
void mrs_tg_plan_destroy(mrs_tg_plan* plan);
int32_t mrs_tg_plan_n_paths(const mrs_tg_plan* plan);
int32_t mrs_tg_plan_n_segments(const mrs_tg_plan* plan);
int32_t mrs_tg_plan_max_segments(const mrs_tg_plan* plan);
/* order_out[q] = index of the path processed in position q (paths sorted by segment count, longest
 * first, stable).  The materialised blocks below are laid out by position q. */
int mrs_tg_plan_get_order(const mrs_tg_plan* plan, int32_t* order_out);

/* The Hessian / mapping-block assembly kernel on its own: for every segment of every path
 * H_i = A_i^-T Q_i A_i^-1 and A_i^-1, both full 10x10 f64 (updateSegmentTimes linear_impl.h:289-304
 * + the block products of constructR :317-320).  seg_times_dev [sum S] (CSR order).
 * Output layout ("slot-major SoA", stated in DESIGN.md): element (r, c) of the block of segment j of
 * the path at position q lives at  ((j * 100 + r * 10 + c) * n_paths + q);  each output holds
 * max_segments * 100 * n_paths doubles; slots j >= S_q are left untouched. Asynchronous. */
int mrs_tg_plan_assemble(mrs_tg_plan* plan, int32_t derivative_to_optimize, const double* seg_times_dev,
                         double* H_dev, double* Ainv_dev);
/* Bytes needed for each of H_dev / Ainv_dev. */
size_t mrs_tg_plan_block_bytes(const mrs_tg_plan* plan);

/* Same contract as mrs_tg_solve_batch but every pointer is DEVICE memory (inputs already resident in
 * HBM) and the call is asynchronous on the context's stream.  samples/cost pointers may be NULL. */
int mrs_tg_plan_solve(mrs_tg_plan* plan, const double* waypoints_dev, const uint8_t* fixed_mask_dev,
                      const double* fixed_values_dev, const double* limits_dev, const mrs_tg_options* opt,
                      double* seg_times_inout_dev, double* coeffs_out_dev, int32_t* status_out_dev,
                      double* cost_out_dev, int32_t* n_samples_out_dev, double* samples_out_dev);

/* A solve with its arguments fixed once: mrs_tg_bound_solve_launch(b) enqueues what mrs_tg_plan_solve would with the
 * arguments given here (options copied; the device buffers must stay where they are).  For a server that keeps several
 * batches in flight and re-issues the same solve step after step: the per-step host cost is one pointer. */
typedef struct mrs_tg_bound_solve mrs_tg_bound_solve;
int mrs_tg_plan_bind_solve(mrs_tg_plan* plan, const double* waypoints_dev, const uint8_t* fixed_mask_dev,
                           const double* fixed_values_dev, const double* limits_dev, const mrs_tg_options* opt,
                           double* seg_times_inout_dev, double* coeffs_out_dev, int32_t* status_out_dev, double* cost_out_dev,
                           int32_t* n_samples_out_dev, double* samples_out_dev, mrs_tg_bound_solve** bound_out);
int mrs_tg_bound_solve_launch(mrs_tg_bound_solve* bound);
void mrs_tg_bound_solve_destroy(mrs_tg_bound_solve* bound);
/* The issue loop of a host that keeps several batches in flight: launch k = 0 .. n_launches-1 goes to bound[k % n_bound]
 * (one bound solve per context + stream).  Stops at the first error and returns its code (mrs_tg_last_error of that
 * solve's context has the text). */
int mrs_tg_bound_solve_launch_many(mrs_tg_bound_solve* const* bound, int32_t n_bound, int32_t n_launches);
/* The same loop on n_threads host threads (the caller + helper threads of the library, created on first use) -- a runtime
 * launch costs the host 3.5-4.5 us, more than four concurrent 10 us kernels take to retire one, so one issuing thread
 * bounds a host with four batches in flight.  The bound solves are partitioned by CONTEXT: all solves of one context are
 * issued by the same thread (a context is driven by one thread at a time), in their order within the run; n_threads is
 * lowered to the number of distinct contexts.  Concurrent callers are served one run after the other.  The helpers spin
 * for 2 ms after a run before they go to sleep.  Returns when every launch has been issued (not finished). */
int mrs_tg_bound_solve_launch_many_mt(mrs_tg_bound_solve* const* bound, int32_t n_bound, int32_t n_launches,
                                      int32_t n_threads);
/* The same loop with consecutive launches packed into one dispatch: launch k still solves bound[k % n_bound], but a run of
 * consecutive launches whose bound solves share a PLAN (one batch structure, one context and stream; the solves differ in
 * their input / output arrays) goes out as a single kernel, on that plan's stream, whose workgroups are divided among the
 * batches (at most 16, and never the same bound solve twice).  bound = [A0, A1, B0, B1] with A*, B* bound to two plans issues
 * (A0 A1), (B0 B1), (A0 A1), ... alternately on the two plans' streams.  A launch costs the host 3 us and a 1024-path solve
 * occupies a quarter of an MI355X for 9 us: a host that issues them one by one is the bottleneck of a short run.
 * Requirements (else MRS_TG_ERR_UNSUPPORTED with a message): fixed segment times, the default solve (no
 * MRS_TG_FLAG_MATERIALIZED_BLOCKS / _GENERAL_PATTERNS), no sampling, batches the default solve takes.  The results are those
 * of mrs_tg_bound_solve_launch_many, bit for bit. */
int mrs_tg_bound_solve_launch_group(mrs_tg_bound_solve* const* bound, int32_t n_bound, int32_t n_launches);
/* Building blocks of the outer loop, exposed for parity tests (device pointers, asynchronous):
 * J_d and the h = 0.1 forward-difference gradient at the given times
 * (getCostAndGradientMellinger, nonlinear_impl.h:257-333): cost_out_dev [n_paths], grad_out_dev [sum S]. */
int mrs_tg_plan_cost_gradient(mrs_tg_plan* plan, int32_t derivative_to_optimize, const uint8_t* fixed_mask_dev,
                              const double* fixed_values_dev, const double* seg_times_dev, double* cost_out_dev,
                              double* grad_out_dev);
/* Per-segment maxima of |p^(k)| for k = 1..3 and the groups {x,y}, {z}, {heading}
 * (Trajectory::computeMaxDerivatives*, trajectory.cpp:422-565): maxima_out_dev [sum S][3][3] indexed
 * [segment][k-1][group]. */
int mrs_tg_plan_segment_maxima(mrs_tg_plan* plan, const double* coeffs_dev, const double* seg_times_dev,
                               double* maxima_out_dev);
/* sampleWholeTrajectory with every field of the sampled state (sampleTrajectoryInRange, trajectory_sampling.cpp:49-104:
 * five evaluateRange passes over the same accumulate-and-carry walk, trajectory.cpp:93-151): for the trajectories given by
 * coeffs_dev [sum S][4][10] and seg_times_dev [sum S], states_out_dev [n_paths][sample_capacity][MRS_TG_STATE_ORDERS][4]
 * holds per sample the derivative orders 0..4 of (x, y, z, heading) -- position_W / velocity_W / acceleration_W / jerk_W /
 * snap_W in the first three columns, yaw (wrapped to (-pi, pi] as setFromYaw's quaternion round trip does), yaw rate and
 * yaw acceleration in the fourth; time_from_start of sample i is i * sampling_dt.  n_samples_out_dev [n_paths] as for the
 * solve calls (capacity + 1 = more samples than fit).  The solve calls' own samples_out (positions and heading, all the
 * nodelet reads: src/mrs_trajectory_generation.cpp:1582-1599) are order 0 of this, bit for bit.  Device pointers,
 * asynchronous on the context's stream. */
/* How many paths of the plan's most recent Mellinger solve with MRS_TG_FLAG_CAREFUL_COST had a trial point whose
 * by-product cost lost its digits and were run again.  Blocks until that solve has finished. */
int mrs_tg_plan_careful_count(mrs_tg_plan* plan, int32_t* count_out);
#define MRS_TG_STATE_ORDERS 5
int mrs_tg_plan_sample_states(mrs_tg_plan* plan, const double* coeffs_dev, const double* seg_times_dev, double sampling_dt,
                              int32_t sample_capacity, int32_t* n_samples_out_dev, double* states_out_dev);

/* Duration in milliseconds of the most recent launch of a kernel, from the start and end time stamps of that very dispatch
 * (the events are attached to the kernel launch itself, hipExtLaunchKernelGGL: what rocprofv3 --kernel-trace reports for
 * it) -- requires mrs_tg_set_profiling(ctx, 1).  kernel_id: 0 block assembly, 1 linear solve, 2 nonlinear outer loop.
 * Blocks until that launch has finished. */
int mrs_tg_set_profiling(mrs_tg_ctx* ctx, int enabled); /* switching it on starts a new series */
int mrs_tg_last_kernel_ms(mrs_tg_ctx* ctx, int kernel_id, float* ms_out);
/* The durations of the newest timed launches of the series (at most 512 are kept, oldest first; launches may be queued
 * back to back, every one carries its own pair of events).  Returns the number written (<= capacity) or a negative
 * MRS_TG_ERR_*.  Blocks until those launches have finished. */
int mrs_tg_kernel_ms_history(mrs_tg_ctx* ctx, int kernel_id, float* ms_out, int capacity);

/* Diagnostics (ABI 4): the names of the kernels the CALLING THREAD has launched through this library since its last
 * mrs_tg_kernel_trace_reset(), oldest first; at most the newest 32 are kept.  names_out receives pointers to static
 * strings ("solve_quad_group_kernel", ...).  Returns the number written.  A test or a benchmark can state which kernels a
 * call ran instead of inferring them from the batch size. */
void mrs_tg_kernel_trace_reset(void);
int mrs_tg_kernel_trace(const char** names_out, int capacity);
/* (ABI 5) The routing table, asked of the routers: the names of the kernels a call WOULD launch for this plan under these
 * options, in launch order -- the calling thread runs the same launch functions in a dry mode in which every size rule,
 * environment knob and hint takes effect and nothing is enqueued (no device work, no argument is dereferenced).
 * group_size 0: what mrs_tg_plan_solve(plan, ..., opt, ...) launches; 1 .. 16: what one dispatch of
 * mrs_tg_bound_solve_launch_group carrying that many batches of this plan launches.  names_out receives pointers to static
 * strings; returns the number written (<= capacity) or a negative MRS_TG_ERR_*.  Resets the calling thread's kernel trace.
 * tests/test_gpu_routing.py pins the route of every BASELINE config and of the nodelet's defaults with it; DESIGN.md section 4's
 * table is its output (scripts/routing_table.py). */
int mrs_tg_plan_explain(mrs_tg_plan* plan, const mrs_tg_options* opt, int32_t group_size, const char** names_out, int32_t capacity);

/* ---- single-path convenience mirroring findTrajectory()'s signature ------------------------- */

typedef struct mrs_tg_waypoint {
  double coords[4]; /* x, y, z, heading  (Waypoint_t, src/mrs_trajectory_generation.cpp:66-70) */
  uint8_t stop_at;
} mrs_tg_waypoint;

typedef struct mrs_tg_initial_state { /* the TrackerCommand fields read at src/...cpp:925-957 */
  double heading;
  double velocity[4], acceleration[4], jerk[4]; /* xyz + heading rate / acceleration / jerk */
} mrs_tg_initial_state;

/* findTrajectory(waypoints, initial_state, sampling_dt, relax_heading) for one path
 * (src/mrs_trajectory_generation.cpp:857-1209), the WHOLE function: builds the vertices (:923-977), estimates the segment
 * times and the Baca total (:1046-1056), optimises, samples, and applies BOTH of the reference's gates -- the accept / reject
 * rule on the nlopt code (:1138-1149) and the temporal sanity check of the sampled trajectory against the Baca estimate
 * (:1178-1199, opt->max_trajectory_len_factor / min_trajectory_len_factor).  limits9 as above.  Returns MRS_TG_OK and
 * *n_samples_out > 0 exactly where the reference returns the states; *n_samples_out = 0 where it returns {} (status_out,
 * seg_times_out and coeffs_out still hold what was computed; mrs_tg_last_error(ctx) has the reference's message and
 * mrs_tg_find_trajectory_info says which gate).  samples_out [sample_capacity][4]; a trajectory with more samples than that
 * which passes the gates is reported as sample_capacity + 1.  initial_state may be NULL. */
int mrs_tg_find_trajectory(mrs_tg_ctx* ctx, const mrs_tg_waypoint* waypoints, int32_t n_waypoints,
                           const mrs_tg_initial_state* initial_state, const double* limits9,
                           const mrs_tg_options* opt, int32_t relax_heading, double* seg_times_out,
                           double* coeffs_out, int32_t* status_out, int32_t* n_samples_out, double* samples_out);

/* (ABI 5) What the context's most recent mrs_tg_find_trajectory decided: *rejection_out = one of MRS_TG_FIND_*, and
 * *baca_total_time_out = initial_total_time_baca (:1048-1056) of that path -- the figure the reference prints beside its
 * "estimated/final trajectory length ratio" (:1201-1203).  Either pointer may be NULL. */
enum {
  MRS_TG_FIND_ACCEPTED = 0,
  MRS_TG_FIND_REJECTED_CODE = 1,      /* the optimiser's code is one the nodelet rejects (:1146-1149) */
  MRS_TG_FIND_REJECTED_TOO_LONG = 2,  /* "the final trajectory sampling is too long" (:1178-1186) */
  MRS_TG_FIND_REJECTED_TOO_SHORT = 3  /* "the final trajectory sampling is too short" (:1188-1196) */
};
int mrs_tg_find_trajectory_info(const mrs_tg_ctx* ctx, int32_t* rejection_out, double* baca_total_time_out);

/* (ABI 5) estimateSegmentTimesBaca (src/eth_trajectory_generation/vertex.cpp:301-485) for one path, as findTrajectory calls it
 * (:1048-1049; also the clock of the fallback sampler, :1309): waypoints [n_waypoints][4] with headings already unwrapped
 * along the path (:935), limits9 as above (after relax_heading), seg_times_out [n_waypoints - 1].  Host arithmetic, no device. */
int mrs_tg_estimate_times_baca(const double* waypoints, int32_t n_waypoints, const double* limits9, double* seg_times_out);

/* ---- path-policy layer: optimize() around findTrajectory(), for a batch of paths ------------- */

/* Parameters of the reference's policy layer (config/public/trajectory_generation.yaml; defaults by
 * mrs_tg_default_policy_options). */
typedef struct mrs_tg_policy_options {
  mrs_tg_options solver;               /* derivative, time allocation, sampling_dt ... of findTrajectory */
  int32_t check_deviation_enabled;     /* check_trajectory_deviation/enabled */
  double max_deviation;                /* check_trajectory_deviation/max_deviation [m] */
  int32_t max_deviation_iterations;    /* check_trajectory_deviation/max_iterations */
  int32_t max_deviation_first_segment; /* max_deviation_first_segment_ (src/...cpp:874-878) */
  double min_waypoint_distance;        /* preprocessPath (:484) */
  int32_t path_straightener_enabled;   /* path_straightener/... (:448-477) */
  double path_straightener_max_deviation, path_straightener_max_hdg_deviation;
  double max_trajectory_len_factor, min_trajectory_len_factor; /* length sanity check (:1178-1199) */
  int32_t fallback_sampling;           /* use findTrajectoryFallback (:1215-1395) instead of the optimiser */
  double fallback_speed_factor, fallback_accel_factor, fallback_stopping_time;
  int32_t override_heading_atan2;      /* getTrajectoryReference (:1582-1597) */
  int32_t reserved_;
  double max_execution_time_s;         /* max_execution_time (:2008-2033); <= 0: none.  As optimize() does (:702-716, :754-768):
                                          a round that starts while overtime() holds runs the fallback sampler for the paths
                                          still active (they succeed, "executing fallback sampling, we are running over time");
                                          otherwise the solver's max_time_s becomes 2 * 0.95 * time left (:899), and paths whose
                                          solve comes back after the deadline fail, as findTrajectory's own checks make the
                                          nodelet give up (:1085, 1156, 1171, 1516-1522).  fallback_sampling = 1 never looks
                                          at the clock */
} mrs_tg_policy_options;

void mrs_tg_default_policy_options(mrs_tg_policy_options* opt);

/* MrsTrajectoryGeneration::optimize() (src/mrs_trajectory_generation.cpp:620-851) for n_paths independent
 * paths: preprocessPath, solve (all still-active paths of a round in ONE batched GPU call), Baca length
 * sanity check, validateTrajectorySpatial, mid-point insertion into unsafe segments, re-solve -- up to
 * max_deviation_iterations rounds.  ROS-only branches (tf, stamps, prediction splicing, overtime) are absent.
 *   wp_offsets [n_paths+1] CSR over `waypoints`; the first waypoint of a path is its initial condition when
 *   has_initial_state[p] != 0 (then initial_states[p] supplies the derivatives, :946-957).
 *   limits [n_paths][9]; relax_heading [n_paths] or NULL.
 *   samples_out [n_paths][sample_capacity][4] (x, y, z, heading); a path needing more samples fails.
 *   success_out [n_paths] 1/0; max_deviation_out, n_waypoints_out (after subdivision), iterations_out may be NULL.
 * Batches of requests: the arrays of a round live in one block of PINNED host memory kept by the context until it is
 * destroyed (about n_paths x sample_capacity x 32 bytes x 1.25 for the largest batch seen; ordinary memory when the runtime
 * refuses it), and the per-path host work runs on up to 16 threads from a few hundred requests on (MRS_TG_POLICY_THREADS). */
int mrs_tg_optimize_paths(mrs_tg_ctx* ctx, int32_t n_paths, const int32_t* wp_offsets, const mrs_tg_waypoint* waypoints,
                          const mrs_tg_initial_state* initial_states, const uint8_t* has_initial_state, const double* limits,
                          const uint8_t* relax_heading, const mrs_tg_policy_options* opt, int32_t sample_capacity,
                          int32_t* success_out, int32_t* n_samples_out, double* samples_out, double* max_deviation_out,
                          int32_t* n_waypoints_out, int32_t* iterations_out);

/* getWaypointInTrajectoryIdxs (src/...cpp:1461-1499) for one path; returns the number of indices written. */
int32_t mrs_tg_waypoint_trajectory_idxs(const double* samples, int32_t n_samples, const mrs_tg_waypoint* waypoints,
                                        int32_t n_waypoints, int32_t* idxs_out);

#ifdef __cplusplus
}
#endif
#endif /* MRS_TG_H_ */
