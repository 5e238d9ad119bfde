// mrs_tg_nonlinear.h -- launch interface of the segment-time outer loop (internal to the library).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "mrs_tg_launch.h"

namespace mrs_tg {

struct NonlinearParams {
  int derivative;
  int max_iterations;
  double f_rel, f_abs, x_rel, x_abs;
  long long time_budget_ticks = 0;  // nlopt maxtime in ticks of the device's constant wall clock (s_memrealtime); 0 = none
  const long long* deadline = nullptr;  // filled in by launch_nonlinear: the call's absolute deadline (device word), or none
  // Paths on which the by-product cost lost its digits (guarded_cost) are listed for the careful re-run
  // (optimize_careful_kernel): position q of the path appended to careful_list (capacity careful_cap) through
  // careful_count; a listed path keeps its start times.  nullptr: no list (the fast kernel's result stands).  A caller of
  // launch_nonlinear only sets careful_cap != 0 to ask for the re-run; the launcher fills in the rest.
  const int32_t* only_flagged = nullptr;  // sweeping kernels: run the paths whose entry (by position) is non-zero only
  int32_t* careful_count = nullptr;
  int32_t* careful_list = nullptr;
  int careful_cap = 0;
  int32_t* queue_next = nullptr;  // lean kernel of a uniform batch larger than the device holds at once: next unclaimed position of the bin
  int ends_min_segments = 2;  // optimize_lean_shared_ends_kernel: shorter paths are left to the sweeping kernel behind it
  int lean_shared = 0;  // lean kernels: shared half sweeps (evaluate_lean_shared) 1: in batches where every path has its S + 4 lanes (own kernel), 2: also wave by wave inside the mixed kernel
  double* sum_t0 = nullptr;  // [n_paths] by path: sum of the times the search starts from (the runaway test of the final solve)
  // The search starts from estimateSegmentTimesEuclidean of these waypoints ([n_vertices][4]) under estimate_limits
  // ([n_paths][9]) instead of from the incoming seg_times: optimize_wave_kernel computes the estimate itself, in front of
  // every other outer-loop kernel launch_nonlinear launches estimate_times_kernel.  nullptr: start from seg_times.
  const double* estimate_wp = nullptr;
  const double* estimate_limits = nullptr;
  const double* pos_wp = nullptr;  // MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS: where the saturated-device solves read vertex positions
  bool reference_status = false;   // MRS_TG_FLAG_REFERENCE_STATUS: no runaway test, the outer loop's own code is the path's status
};

// Paths are sorted by segment count (longest first), so every lane-group class is a contiguous range
// of positions q.  A path with S segments uses a group of G = min(64, pow2ceil(S + 1)) lanes: lane k of
// the group evaluates the cost at the k-th perturbed time vector (k = 0: unperturbed).
struct DfoParams {
  int derivative;
  int mode;            // 0 squared time, 1 Richter time, 3 / 4 the same with the free constraints as variables
  int max_iterations;
  double f_rel, f_abs, x_rel, x_abs;
  double time_penalty, soft_weight;
  int use_soft;
  double initial_stepsize_rel;
  long long time_budget_ticks = 0;  // as NonlinearParams
};

struct NonlinearBin {
  int group;      // lanes per path: 4, 8, 16, 32 or 64
  int q_begin;    // first position of the bin
  int q_count;    // number of paths in the bin
  int max_S;      // largest segment count in the bin
  int min_S;      // smallest
};

struct NonlinearPlan {
  std::vector<NonlinearBin> bins;
  std::vector<NonlinearBin> bins1;   // the one-lane-per-vector groups of a plan whose own choice is the dimension split (regroup_possible)
  bool regroup_possible = false;     // a call may take the lane-group kernels instead of the split kernel (launch_nonlinear)
  // the plain-path outer loop's bins when EVERY path of four or more segments gets the lanes of the shared half sweeps
  // (G = pow2ceil(S + 4) instead of pow2ceil(S + 1): 5-7, 13-15 and 29-30 segments move to the next group width); chosen
  // by launch_nonlinear while the launch is about as large as the device holds at once (see there)
  std::vector<NonlinearBin> wide_bins;
  int wide_blocks = 0;             // workgroups of a launch over wide_bins
  // objective orders below snap (free slots at the end vertices): every path of four or more segments in a group of at least
  // S + 4 lanes, for the shared half sweeps with free ends (optimize_lean_shared_ends_kernel); empty when a path is too long
  std::vector<NonlinearBin> ends_bins;
  int dim_split = 1;               // lanes per time vector in the outer loop: 1 (compact) or 4 (one per dimension)
  double* d_ws = nullptr;          // factor store of the per-lane linear solve
  size_t ws_doubles = 0;
  int32_t* d_opt_status = nullptr; // stopping reason of the outer loop per path
  double* d_maxima = nullptr;      // [n_segments][9] per-segment maxima
  int32_t* d_queue = nullptr;      // the lean kernel's path queue (one counter)
  double* d_sum_t0 = nullptr;      // [n_paths] sum of the times the outer loop started from (runaway test)
  int32_t* d_fallback = nullptr;   // [n_paths] by position: 1 = the prefix / suffix kernel left the path to the sweeping kernel
  int32_t* d_careful = nullptr;    // [0] count, [2] count of the last completed call, [4..] list of guarded paths
  double* d_careful_ws = nullptr;  // factor store of optimize_careful_kernel's lanes
  size_t careful_ws_doubles = 0;
  // paths with a position-free vertex (MRS_TG_FLAG_GENERAL_PATTERNS), allocated on first use
  int32_t* d_general = nullptr;          // [0] count | [4 + p] flag by path | [4 + n_paths + k] list of positions
  double* d_general_ws = nullptr;        // factor store of optimize_general_kernel's lanes
  double* d_general_solve_ws = nullptr;  // factor store of solve_general_kernel (general_workspace_doubles)
  double* d_general_t0 = nullptr;        // [n_segments] the times the call started from
  // gradient-free modes (0, 1, 3, 4), allocated on first use
  double* d_dfo_vec = nullptr;       // x | x0 | best | h | lb | ub, each 21 n_segments + 20 n_paths doubles
  double* d_dfo_f = nullptr;         // fbest | f_sweep | J_d scratch, 3 * n_paths doubles
  int32_t* d_dfo_state = nullptr;    // phase, i, sg, neval, improved, ret, done, n_var per path
  int32_t* d_dfo_fidx = nullptr;     // modes 3/4: free-constraint index of every (vertex, derivative), -1 if fixed
  double* d_dfo_segcost = nullptr;   // modes 3/4: J_d share of every (segment, dimension)
  int32_t* d_dfo_seg_path = nullptr; // modes 3/4: path of every segment
  long long* d_dfo_deadline = nullptr; // wall-clock deadline of the running search (0 = none), written by its first kernel
};

// shared by the launchers of mrs_tg_nonlinear.hip and mrs_tg_dfo.hip: the plan's lazily allocated buffers, and the flags / list
// of a call's paths with a position-free vertex (`outer_loop`: also what optimize_general_kernel needs; *cap = paths per launch)
hipError_t nonlinear_ensure_buffers(NonlinearPlan& nl, const BatchView& b);
hipError_t nonlinear_prepare_general(NonlinearPlan& nl, const BatchView& b, const uint8_t* mask, const double* seg_times,
                                     bool outer_loop, int* cap, hipStream_t stream);

// whether optimize_careful_kernel (MRS_TG_FLAG_CAREFUL_COST) is in this build (-DMRS_TG_WITH_CAREFUL=1)
bool careful_rerun_built();

int nonlinear_plan_build(NonlinearPlan& nl, const std::vector<int32_t>& seg_offsets, const std::vector<int32_t>& order);
void nonlinear_plan_free(NonlinearPlan& nl);

// general: paths with a position-free vertex may occur (MRS_TG_FLAG_GENERAL_PATTERNS); they take the 5 x 5-block route
// (optimize_general_kernel / solve_general_kernel) behind the fast kernels, the others are untouched by it.
// sampling_dt > 0: the caller wants the result sampled; when the final solve runs on the rows kernel the sampling rides
// on that launch (*sampled_out = true) and the caller must not launch the sampler again
hipError_t launch_nonlinear(NonlinearPlan& nl, const BatchView& b, const NonlinearParams& prm, const uint8_t* mask,
                            const double* vals, const double* limits, double* seg_times, double* coeffs,
                            int32_t* status, double* cost, hipStream_t stream, double sampling_dt = 0.0,
                            int sample_capacity = 0, int32_t* n_samples = nullptr, double* samples = nullptr,
                            bool* sampled_out = nullptr, bool general = false);
hipError_t launch_dfo(NonlinearPlan& nl, const BatchView& b, const DfoParams& prm, const uint8_t* mask, const double* vals,
                      const double* limits, double* seg_times, double* coeffs, int32_t* status, double* cost,
                      hipStream_t stream, bool general = false);
hipError_t launch_cost_gradient(NonlinearPlan& nl, const BatchView& b, int d, const uint8_t* mask, const double* vals,
                                const double* seg_times, double* cost, double* grad, hipStream_t stream);
hipError_t launch_segment_maxima(const BatchView& b, const double* coeffs, const double* seg_times, double* maxima,
                                 hipStream_t stream);

}  // namespace mrs_tg
