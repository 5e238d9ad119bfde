/*
 * mrs_tg_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the hot path of ctu-mrs/mrs_uav_trajectory_generation
 * (MrsTrajectoryGeneration::findTrajectory, src/mrs_trajectory_generation.cpp:857-1209, and
 * the vendored eth_trajectory_generation library below it).  Each function cites the
 * reference file:line it follows.  It deliberately uses the reference's own arithmetic route
 * (per-segment A(T), Schur-complement inverse, H = A^-T Q A^-1, dense R = C^T H C, QR solve,
 * Jenkins-Traub roots) and shares NO code or constants with the HIP product path.
 * Two further arithmetic routes of the SAME algorithm can be switched on for the linear solve (mto_set_arithmetic below: exactly
 * rounded unit-time tables; the whole solve in 113-bit arithmetic) -- tests use them to tell the rounding noise of the
 * reference's route from a difference in the algorithm.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * PARITY UNPINNED: the reference ships no golden vectors / known-answer tests for this path
 *  and cannot be compiled here (Eigen3, NLopt, mrs_lib, ROS absent), so
 * this oracle is pinned only against closed-form known answers and 60-digit mpmath ground
 * truth of the same formulas (tests/golden, oracle/gen_golden.py), not against reference output.
 *
 * All file:line citations are relative to /root/reference/.
 */
#ifndef MRS_TG_ORACLE_H_
#define MRS_TG_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTO_N 10    /* coefficients per polynomial   (src/mrs_trajectory_generation.cpp:1063) */
#define MTO_D 4     /* dimensions x,y,z,heading      (src/mrs_trajectory_generation.cpp:902)  */
#define MTO_HALF 5  /* derivative slots per segment end (N/2) */
#define MTO_MAX_SEG 256 /* = MRS_TG_MAX_SEGMENTS of the product (include/mrs_tg.h) */
#define MTO_RUNAWAY_TIME_FACTOR 25.0 /* see solve_one in mto_nonlinear.c */

/* nlopt-style result codes (nlopt.h; gate at src/mrs_trajectory_generation.cpp:1138-1149) */
enum {
  MTO_FAILURE = -1, MTO_INVALID_ARGS = -2, MTO_ROUNDOFF_LIMITED = -4,
  MTO_SUCCESS = 1, MTO_STOPVAL_REACHED = 2, MTO_FTOL_REACHED = 3, MTO_XTOL_REACHED = 4,
  MTO_MAXEVAL_REACHED = 5, MTO_MAXTIME_REACHED = 6
};

/* One path.  Vertex v has 5 derivative slots k (0=position..4=snap); fixed_mask[v*5+k] != 0
 * means the vertex carries a constraint on derivative k (Vertex::addConstraint,
 * src/eth_trajectory_generation/vertex.cpp:134-137) with value fixed_values[(v*5+k)*4+dim]. */
typedef struct {
  int n_seg;                    /* S; vertices = S+1 */
  int derivative_to_optimize;   /* 2,3,4 (acc, jerk, snap) src/mrs_trajectory_generation.cpp:904-919 */
  const uint8_t* fixed_mask;    /* [(S+1)*5] */
  const double* fixed_values;   /* [(S+1)*5*4] */
} mto_path;

typedef struct {
  int max_iterations;           /* nlopt maxeval            nonlinear_impl.h:73  */
  double f_rel, f_abs;          /* nlopt ftol               nonlinear_impl.h:69-70 */
  double x_rel, x_abs;          /* nlopt xtol               nonlinear_impl.h:71-72 */
} mto_nlopt_params;

/* ---- polynomial primitives ---------------------------------------------------------- */
/* B[r][k] = k!/(k-r)!  polynomial.cpp:155-170 */
double mto_base_coeff(int r, int k);
/* Horner with derivative table, polynomial.h:150-163 */
double mto_poly_eval(const double* c, int n, double t, int derivative);
/* derivative coefficients zero-padded to n, polynomial.h:108-119 */
void mto_poly_derivative(const double* c, int n, int derivative, double* out);
/* discrete convolution, polynomial.cpp:176-192; out has nd+nk-1 entries */
void mto_convolve(const double* data, int nd, const double* kernel, int nk, double* out);
/* all complex roots of a real polynomial given in INCREASING powers
 * (findRootsJenkinsTraub, rpoly/rpoly_ak1.cpp:76-120 + rpoly_ak1 :148-932).
 * returns number of roots written (0 if none), -1 on failure. */
int mto_find_roots_jenkins_traub(const double* coeffs_increasing, int n_coeffs, double* re, double* im);

/* ---- per-segment matrices (row-major 10x10) ----------------------------------------- */
void mto_mapping_matrix(double T, double* A);                          /* linear_impl.h:113-121 */
void mto_invert_mapping_matrix(const double* A, double* Ainv);         /* linear_impl.h:148-177 */
void mto_cost_matrix(int derivative, double T, double* Q);             /* linear_impl.h:606-618 */
void mto_segment_hessian(int derivative, double T, double* Hout, double* Ainv_out); /* linear_impl.h:320 */
/* Arithmetic route of the per-segment matrices H(T), A^-1(T) (process-wide; set it before a batch, not during one):
 * 0 (default) = the reference's route (A inverted numerically, Q from pow(), dense products);
 * 1 = the same matrices from exactly rounded unit-time tables (mto_linear.c) -- the reference's algorithm without the
 *     rounding noise of its route, for tests that must tell the two apart;
 * 2 = the whole linear solve (tables, R, QR, coefficients, cost) in 113-bit arithmetic, rounded to double once: the value
 *     both other routes and the HIP path approximate.  ~50x slower; the optimisers around it stay in double. */
void mto_set_arithmetic(int mode);
/* 1: mto_solve_batch reports a Mellinger result whose feasibility scaling multiplied the path's total time by more than
 * MTO_RUNAWAY_TIME_FACTOR as MTO_ROUNDOFF_LIMITED, as the PRODUCT does (a documented deviation of include/mrs_tg.h from the
 * reference, which returns the outer loop's code and discards the trajectory by the nodelet's length check); 0 (default): the
 * reference's behaviour. */
void mto_set_runaway_rule(int on);
/* 1: the QR solve of R_pp runs every loop over its full dense range; 0 (default): the same loops limited to the band of
 * R_pp outside of which every operand is an exact zero -- bit-identical results (mto_linear.c qr_solve), O(n) instead of
 * O(n^3) in the path length */
void mto_set_dense_qr(int on);
int mto_get_arithmetic(void);
/* the unit-time tables of route 1: ABAR^-1 [10][10], HBAR_d [5][10][10] (113-bit arithmetic, rounded once) */
void mto_unit_tables(double* abar_inv_out, double* hbar_out);

/* ---- linear QP ------------------------------------------------------------------------ */
/* setupFromVertices + solveLinear (linear_impl.h:62-106,184-257,311-373).
 * coeffs_out [S][4][10] ascending powers. Returns 0 on success. */
int mto_solve_linear(const mto_path* path, const double* seg_times, double* coeffs_out);
/* number of free constraints n_free (every unconstrained (vertex, derivative 0..4), linear_impl.h:191-254) */
int mto_count_free_constraints(const mto_path* path);
/* as mto_solve_linear; free_out (may be NULL) receives getFreeConstraints: [4][n_free], ordered by (vertex, derivative) */
int mto_solve_linear_free(const mto_path* path, const double* seg_times, double* coeffs_out, double* free_out);
/* setFreeConstraints + updateSegmentsFromCompactConstraints (linear_impl.h:515-522, 264-282): coefficients
 * from given free constraints without a solve */
int mto_coeffs_from_free_constraints(const mto_path* path, const double* seg_times, const double* free_in, double* coeffs_out);
/* computeCost linear_impl.h:128-141 */
double mto_compute_cost(int n_seg, int derivative, const double* seg_times, const double* coeffs);

/* ---- nonlinear time allocation (mode 2, Mellinger) ------------------------------------ */
/* getCostAndGradientMellinger nonlinear_impl.h:257-333; grad may be NULL */
double mto_cost_and_gradient_mellinger(const mto_path* path, const double* seg_times, double* grad);
/* Self-defined deterministic projected L-BFGS standing in for NLopt LD_LBFGS
 * (optimizeTimeMellingerOuterLoop nonlinear_impl.h:160-234).  See DESIGN.md "outer loop".
 * seg_times is in/out: on return it holds the LAST EVALUATED point (that is what the
 * reference's poly_opt_ holds when scaleSegmentTimesWithViolation runs, nonlinear_impl.h:213).
 * n_eval_out may be NULL. */
int mto_optimize_times_mellinger(const mto_path* path, const mto_nlopt_params* prm, double* seg_times,
                                 int* n_eval_out, double* final_cost_out);

/* ---- feasibility ---------------------------------------------------------------------- */
/* Trajectory::computeMinMaxMagnitude for one segment (trajectory.cpp:211-243,
 * segment.cpp:113-212): maximum over candidates of the norm of derivative `derivative`
 * restricted to dims[0..n_dims). coeffs = that segment's [4][10]. */
double mto_segment_max_magnitude(const double* seg_coeffs, double T, int derivative, const int* dims, int n_dims);
/* Trajectory::scaleSegmentTimesToMeetConstraints trajectory.cpp:598-692.
 * limits[9] = {v_h, v_v, v_hdg, a_h, a_v, a_hdg, j_h, j_v, j_hdg}. Scales coeffs in place and
 * seg_times in place; returns within_range (0/1); n_sweeps_out may be NULL. */
int mto_scale_segment_times_to_meet_constraints(int n_seg, double* coeffs, double* seg_times,
                                                const double* limits, int* n_sweeps_out);

/* ---- sampling --------------------------------------------------------------------------- */
/* sampleWholeTrajectory -> Trajectory::evaluateRange (trajectory_sampling.cpp:49-124,
 * trajectory.cpp:93-151). Writes up to capacity samples of [x,y,z,heading_raw] for derivative
 * `derivative`; returns the number the reference would produce, or capacity + 1 if that is more than capacity
 * (capacity 0: unbounded count). */
int mto_sample_trajectory(int n_seg, const double* coeffs, const double* seg_times, double dt,
                          int derivative, double* out, int capacity);
/* yaw after the quaternion round trip (eth_mav_msgs/common.h:130-140, eigen_mav_msgs.h setFromYaw/getYaw) */
double mto_wrap_yaw(double yaw);

/* ---- input side -------------------------------------------------------------------------- */
/* estimateSegmentTimesEuclidean vertex.cpp:491-565. waypoints [S+1][4].
 * limits as above (heading limits >= FLT_MAX disable the heading term). */
void mto_estimate_segment_times_euclidean(int n_seg, const double* waypoints, const double* limits, double* times_out);
/* estimateSegmentTimesBaca vertex.cpp:301-485 */
void mto_estimate_segment_times_baca(int n_seg, const double* waypoints, const double* limits, double* times_out);
/* sradians::unwrap as used at src/mrs_trajectory_generation.cpp:935 */
double mto_unwrap_heading(double what, double from);

/* ---- whole path: findTrajectory core (src/mrs_trajectory_generation.cpp:1046-1169) ------ */
typedef struct {
  int derivative_to_optimize;
  int time_alloc_method;      /* -1: fixed times, linear only;  2: Mellinger outer loop + scaling */
  int estimate_times;         /* 1: seg_times from the Euclidean estimator, 0: as given */
  mto_nlopt_params nlopt;
  double sampling_dt;         /* <= 0: no sampling */
  double time_penalty;        /* modes 0 / 1 */
  int use_soft_constraints;
  double soft_constraint_weight;
  double initial_stepsize_rel;
} mto_options;

/* Batch driver in the C-ABI's CSR layout (include/mrs_tg.h). Single thread when n_threads<=1,
 * otherwise a static partition over pthreads. samples_out [n_paths][sample_capacity][4]
 * (x,y,z,wrapped heading) or NULL. */
int mto_solve_batch(int n_paths, const int32_t* seg_offsets, const double* waypoints,
                    const uint8_t* fixed_mask, const double* fixed_values, const double* limits,
                    const mto_options* opt, double* seg_times_inout, double* coeffs_out,
                    int32_t* status_out, double* cost_out, int32_t* n_samples_out, double* samples_out,
                    int sample_capacity, int n_threads);


/* ---- gradient-free time-allocation modes 0 / 1 (mto_dfo.c) ---------------------------------- */
typedef struct {
  int time_alloc_method;        /* 0 kSquaredTime, 1 kRichterTime, 3 / 4 the same + free constraints (nonlinear.h:92-100) */
  mto_nlopt_params nlopt;
  double time_penalty;          /* nonlinear.h:70; param time_penalty */
  int use_soft_constraints;     /* param soft_constraints_enabled */
  double soft_constraint_weight;
  double initial_stepsize_rel;  /* 0.1, src/mrs_trajectory_generation.cpp:893 */
} mto_dfo_params;

/* computeMaximumOfMagnitude over all four dimensions (linear_impl.h:478-508) */
double mto_max_of_magnitude(int n_seg, const double* coeffs, const double* seg_times, int derivative);
/* evaluateMaximumMagnitudeAsSoftConstraint over the 12 registered constraints (nonlinear_impl.h:740-762,
 * src/mrs_trajectory_generation.cpp:1067-1081); maxima[k-1] = 4-D maximum of derivative k */
double mto_soft_constraint_cost(const double maxima[3], const double* limits9, double weight);
/* objectiveFunctionTime nonlinear_impl.h:568-614; parts_out [3] = trajectory, time, soft (may be NULL) */
double mto_objective_time(const mto_path* path, const double* seg_times, const double* limits9, const mto_dfo_params* prm,
                          double* parts_out);
/* own derivative-free search standing in for NLopt LN_BOBYQA; x in/out = last evaluated point */
int mto_optimize_time_dfo(const mto_path* path, const double* limits9, const mto_dfo_params* prm, double* x,
                          int* n_eval_out, double* f_last_out);

/* modes 3 / 4: x = [segment times, free constraints of dimension 0..3]; coeffs_out may be NULL
 * (objectiveFunctionTimeAndConstraints nonlinear_impl.h:651-722) */
double mto_objective_time_and_constraints(const mto_path* path, const double* x, const double* limits9,
                                          const mto_dfo_params* prm, double* coeffs_out, double* parts_out);
/* setFreeEndpointDerivativeHardConstraints nonlinear_impl.h:765-804; lower / upper [4 * n_free] */
void mto_free_derivative_bounds(const mto_path* path, const double* limits9, double* lower, double* upper);
/* optimizeTimeAndFreeConstraints nonlinear_impl.h:429-536 with the same search; seg_times in/out and coeffs_out
 * are those of the last evaluated point */
int mto_optimize_time_and_constraints_dfo(const mto_path* path, const double* limits9, const mto_dfo_params* prm,
                                          double* seg_times, double* coeffs_out, int* n_eval_out, double* f_last_out);

/* ---- path-policy layer (SURVEY.md 8 f: the rows ranked "next") ------------------------------ */
typedef struct {
  int check_deviation_enabled;     /* check_trajectory_deviation/enabled        (config/public/trajectory_generation.yaml) */
  double max_deviation;            /* check_trajectory_deviation/max_deviation  0.05 m */
  int max_deviation_iterations;    /* check_trajectory_deviation/max_iterations 6 */
  int max_deviation_first_segment; /* member flag set at src/mrs_trajectory_generation.cpp:874-878 */
  double min_waypoint_distance;    /* 0.05 m */
  int path_straightener_enabled;
  double path_straightener_max_deviation, path_straightener_max_hdg_deviation;
  double max_trajectory_len_factor, min_trajectory_len_factor; /* 3.0, 0.33 */
  int fallback_sampling;           /* use findTrajectoryFallback instead of the optimiser */
  double fallback_speed_factor, fallback_accel_factor, fallback_stopping_time;
  int override_heading_atan2;
} mto_policy_params;

void mto_default_policy_params(mto_policy_params* p);
double mto_dist_from_segment(const double* point, const double* seg1, const double* seg2);        /* :1533-1554 */
void mto_interpolate_point(const double* a, const double* b, double coeff, double* out);          /* :1612-1625 */
int mto_preprocess_path(const double* wp_in, const uint8_t* stop_in, int n_in, const mto_policy_params* prm,
                        double* wp_out, uint8_t* stop_out);                                        /* :431-500 */
int mto_validate_trajectory_spatial(const double* samples, int n_samples, const double* wps, int n_wp,
                                    const mto_policy_params* prm, uint8_t* segment_safe, double* max_deviation_out); /* :1401-1455 */
int mto_waypoint_trajectory_idxs(const double* samples, int n_samples, const double* wps, int n_wp, int32_t* idxs); /* :1461-1499 */
int mto_fallback_sampling(const double* wps, const uint8_t* stop_at, int n_wp, const double* limits9, int relax_heading,
                          const mto_policy_params* prm, double dt, double* out, int capacity);    /* :1215-1395 */
/* findTrajectory() :857-1209 for one (already preprocessed) path, both gates included: returns 1 where the reference returns
 * the states.  seg_times_out / status_out / baca_total_out / raw_n_samples_out / rejection_out may be NULL
 * (rejection: 0 accepted, 1 optimiser code, 2 too long, 3 too short, 4 more samples than capacity). */
int mto_find_trajectory(const double* wps, const uint8_t* stop_at, int n_wp, const double* initial_state, const double* limits9,
                        int relax_heading, const mto_options* sopt, const mto_policy_params* prm, double* samples_out,
                        int capacity, int* n_samples_out, double* seg_times_out, int32_t* status_out, double* baca_total_out,
                        int* raw_n_samples_out, int* rejection_out);
/* optimize() :620-851 for one path. wps_in [n_in][4] (first = initial condition when initial_state != NULL),
 * initial_state = {heading, velocity[4], acceleration[4], jerk[4]} or NULL. samples_out [capacity][4].
 * Returns success (1/0). */
int mto_optimize_path(const double* wps_in, const uint8_t* stop_in, int n_in, const double* initial_state,
                      const double* limits9, int relax_heading, const mto_options* sopt, const mto_policy_params* prm,
                      double* samples_out, int capacity, int* n_samples_out, double* max_deviation_out,
                      int* n_waypoints_out, int* iterations_out);

#ifdef __cplusplus
}
#endif
/* decision trace of mto_optimize_times_mellinger on the calling thread (see mto_nonlinear.c); buf NULL switches it off */
#define MTO_TRACE_REC 8
void mto_set_optimizer_trace(double* buf, int cap_records);
int mto_optimizer_trace_count(void);

/* per-thread scratch memory for temporaries (mto_scratch.c): mark, allocate (64-byte aligned; zero != 0 clears), release */
typedef struct {
  int block;
  size_t top;
} mto_scratch_state;
mto_scratch_state mto_scratch_mark(void);
void mto_scratch_release(mto_scratch_state s);
void* mto_scratch_alloc(size_t bytes, int zero);

#endif
