// mrs_tg_general.hip -- the linear QP for ANY fixed / free pattern.
//
// PolynomialOptimization::setupFromVertices accepts a vertex without a position constraint
// (/root/reference/include/eth_trajectory_generation/impl/polynomial_optimization_linear_impl.h:184-257 builds the
// reordering for whatever is fixed and free); the nodelet never produces one (every waypoint fixes its position,
// src/mrs_trajectory_generation.cpp:942-972), and the fast kernels are built on that: their vertex blocks are the four
// derivative slots 1..4, a path with a position-free vertex comes back from them with status -2.  This kernel solves those
// paths: the same block-tridiagonal elimination with 5 x 5 vertex blocks (slots 0..4) and masks over all five, none of the
// specialisations, one lane per (path, dimension) with the factors parked in global memory, coefficients and the cost
// 0.5 c^T Q c on the way back (mrs_tg_general.hpp).  Launched only when the caller says such vertices may occur
// (MRS_TG_FLAG_GENERAL_PATTERNS; mrs_tg_solve_batch sees it in its host copy of the masks): behind the fast solve of the
// fixed-times mode, and behind every fast solve of the time-allocation pipelines (mrs_tg_nonlinear.hip).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "mrs_tg_general.hpp"
#include "mrs_tg_launch.h"

namespace mrs_tg {

size_t general_workspace_doubles(const BatchView& b) {
  return (size_t)(b.max_segments + 1) * kGenWs * 4 * (size_t)b.n_paths;
}

// One lane per (path, dimension).  Which paths: those listed in `only` (per path, non-zero = take it) when given, else those
// the fast kernels returned with status -2.  Status out: 1, or the outer loop's stopping reason when `opt_status` is given
// (merge_status); `status` and `cost` may be NULL.
__global__ __launch_bounds__(64) void solve_general_kernel(BatchView b, int d, const uint8_t* __restrict__ mask,
                                                           const double* __restrict__ vals,
                                                           const double* __restrict__ seg_times, double* __restrict__ ws,
                                                           double* __restrict__ coeffs, int32_t* __restrict__ status,
                                                           double* __restrict__ cost, const int32_t* __restrict__ only,
                                                           const int32_t* __restrict__ opt_status) {
  const unsigned t = blockIdx.x * 64u + threadIdx.x;
  const int q = (int)(t >> 2), dim = (int)(t & 3u);
  if (q >= b.n_paths) return;
  const PathRef pr = path_at(b, q);
  if (only ? only[pr.p] == 0 : status[pr.p] != -2) return;  // (the four lanes of a path agree)
  const double* __restrict__ times = seg_times + pr.s0;
  double total = general_solve_lane<true>(mask, vals, pr.v0, pr.S, d, dim, [&](int i) { return times[i]; }, ws,
                                          (size_t)b.n_paths * 4, t, coeffs + (size_t)pr.s0 * kD * kN);
  total += __shfl_xor(total, 1, 64);
  total += __shfl_xor(total, 2, 64);
  if (dim == 0) {
    if (cost) cost[pr.p] = total;
    if (status) status[pr.p] = merge_status(true, opt_status, pr.p);  // (after the reads above: the other three lanes of the quad are in this wavefront)
  }
}

hipError_t launch_solve_general(const BatchView& b, int d, const uint8_t* mask, const double* vals, const double* seg_times,
                                double* ws, double* coeffs, int32_t* status, double* cost, hipStream_t stream,
                                const int32_t* only, const int32_t* opt_status) {
  if (b.n_paths == 0) return hipSuccess;
  if (!only && !status) return hipErrorInvalidValue;
  const unsigned grid = (unsigned)(((size_t)b.n_paths * 4 + 63) / 64);
  MRS_TG_LAUNCH(solve_general_kernel, dim3(grid), dim3(64), 0, stream, b, d, mask, vals, seg_times, ws, coeffs, status,
                     cost, only, opt_status);
  return hipGetLastError();
}

}  // namespace mrs_tg
