// mrs_tg_nl_common.hpp -- small helpers shared by the time-allocation kernels (mrs_tg_nonlinear.hip: the Mellinger outer loop;
// mrs_tg_dfo.hip: the gradient-free modes; mrs_tg_maxima.hpp: per-segment maxima).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>

namespace mrs_tg {

// cross-lane moves (DPP within a row of 16, v_readlane across rows)
// (bound_ctrl set: with all rows and banks enabled and a permutation that stays inside the row every lane is written, so
// the "old" operand is dead -- without the flag the compiler materialises it, two more v_mov_b32 per moved double)
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double row_value(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ bool relstop(double vold, double vnew, double reltol, double abstol) {
  // NLopt's scalar stopping rule
  if (isinf(vold)) return false;
  const double dv = fabs(vnew - vold);
  return dv < abstol || dv < reltol * (fabs(vnew) + fabs(vold)) * 0.5 || (reltol > 0 && vnew == vold);
}

// the same rule without branches (every term is evaluated; for the outer loop's bookkeeping, where the short-circuit form
// turns into a ladder of exec-mask branches)
__device__ __forceinline__ bool relstop_flat(double vold, double vnew, double reltol, double abstol) {
  const double dv = fabs(vnew - vold);
  const bool hit = (dv < abstol) | (dv < reltol * (fabs(vnew) + fabs(vold)) * 0.5) | ((reltol > 0) & (vnew == vold));
  return (!isinf(vold)) & hit;
}

// reciprocal: hardware estimate + two Newton steps (~1 ulp)
__device__ __forceinline__ double rcp_refined(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  y = fma(fma(-x, y, 1.0), y, y);
  return y;
}

static inline unsigned cdiv_u(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

}  // namespace mrs_tg
