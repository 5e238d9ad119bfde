// mrs_tg_dfo.hip -- the gradient-free time-allocation modes 0 / 1 / 3 / 4: kernels and launcher.
// Reference: optimizeTime / objectiveFunctionTime
// (/root/reference/include/eth_trajectory_generation/impl/polynomial_optimization_nonlinear_impl.h:121-157, 568-614),
// optimizeTimeAndFreeConstraints / objectiveFunctionTimeAndConstraints / setFreeEndpointDerivativeHardConstraints
// (:430-536, 652-722, 765-804), soft constraints (:725-762).  The search is the project's own (CPU
// statement in oracle/mto_dfo.c): NLopt's BOBYQA is not reproduced.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdint>

#include "mrs_tg_device.hpp"
#include "mrs_tg_pool.h"
#include "mrs_tg_nonlinear.h"
#include "mrs_tg_nl_common.hpp"
#include "mrs_tg_maxima.hpp"

namespace mrs_tg {

// ---------------------------------------------------------------------------------------------
// gradient-free modes 0 / 1 (objectiveFunctionTime, nonlinear_impl.h:568-614) and 3 / 4
// (objectiveFunctionTimeAndConstraints, :651-722).  Per evaluation: the trajectory of the trial point
// (modes 0 / 1: a fused solve at the trial times; modes 3 / 4: coefficients straight from the trial times and
// trial free end-point derivatives, no solve), the 4-D magnitude maxima of v, a, j
// (computeMaximumOfMagnitude, linear_impl.h:478-508: all four dimensions in one norm, quirk B6) and one step of
// the search's state machine.  The search is this project's own ("MRS-DFO"; CPU statement in
// oracle/mto_dfo.c): NLopt's BOBYQA is not reproduced.
//
// Search variables of path p (first segment s0): [T_0..T_{S-1}, free constraints of dimension 0, 1, 2, 3], at
// most S + 20 (S + 1) of them; its vectors start at 21 s0 + 20 p inside each of the six arrays
// x | x0 | best | h | lb | ub (each NV = 21 n_segments + 20 n_paths doubles).

// maxima4[seg * 3 + (k-1)] = max over the segment of |p^(k)| over x, y, z, heading
__global__ __launch_bounds__(64) void segment_maxima4_kernel(int n_segments, const double* __restrict__ coeffs,
                                                             const double* __restrict__ seg_times,
                                                             double* __restrict__ maxima4) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= n_segments) return;
  const int k = blockIdx.y + 1;
  const double* c = coeffs + (size_t)s * kD * kN;
  const double T = seg_times[s];
  double cb[kD][kN];
  double tp = 1.0;
#pragma unroll
  for (int j = 0; j < kN; ++j) {
#pragma unroll
    for (int q = 0; q < kD; ++q) cb[q][j] = c[q * kN + j] * tp;
    tp *= T;
  }
  const double ti = 1.0 / T;
  double m2, scale;
  if (k == 1) {
    m2 = max_mag2<1, kD>(cb);
    scale = ti;
  } else if (k == 2) {
    m2 = max_mag2<2, kD>(cb);
    scale = ti * ti;
  } else {
    m2 = max_mag2<3, kD>(cb);
    scale = ti * ti * ti;
  }
  maxima4[(size_t)s * 3 + (k - 1)] = sqrt(m2) * scale;
}

enum { kDfoFirst = -1, kDfoSearch = 0, kDfoRevisit = 1, kDfoInitPlus = 2, kDfoInitMinus = 3, kDfoCompass = 4 };
constexpr double kDfoMinDecrease = 1.0e-6;  // a trial must lower the best value by more than this share of it (oracle/mto_dfo.c)
// per-path int state: [0] phase [1] i [2] sg [3] neval [4] improved [5] ret [6] done [7] number of variables
//                     [8] a trial of the current coordinate was accepted [9] code to stop with after the revisit
//                     [10] the last evaluated trial is the best point
constexpr int kDfoInts = 11;

__host__ __device__ __forceinline__ size_t dfo_var_offset(int s0, int p) { return (size_t)21 * s0 + (size_t)20 * p; }
__host__ __device__ __forceinline__ size_t dfo_var_total(int n_segments, int n_paths) {
  return (size_t)21 * n_segments + (size_t)20 * n_paths;
}

struct DfoVectors {
  double *x, *x0, *best, *h, *lb, *ub;
};
__device__ __forceinline__ DfoVectors dfo_vectors(double* vec, size_t NV, size_t off) {
  return DfoVectors{vec + off, vec + NV + off, vec + 2 * NV + off, vec + 3 * NV + off, vec + 4 * NV + off, vec + 5 * NV + off};
}

// derivative k of one polynomial at t
__device__ __forceinline__ double poly_derivative_at(const double* __restrict__ c, int k, double t) {
  double acc = 0.0;
  for (int j = kN - 1; j >= k; --j) {
    double f = 1.0;
    for (int r = 0; r < k; ++r) f *= (double)(j - r);
    acc = acc * t + f * c[j];
  }
  return acc;
}

// start point, steps and bounds.  Modes 0 / 1: optimizeTime nonlinear_impl.h:121-157; modes 3 / 4:
// optimizeTimeAndFreeConstraints :429-536 with the bounds of setFreeEndpointDerivativeHardConstraints :765-804
// (its walk over derivatives 0..derivative_to_optimize is replicated as written, oracle/mto_dfo.c).
// coeffs: the linear solution at seg_times (modes 3 / 4: source of the initial free constraints).
__global__ __launch_bounds__(64) void dfo_init_kernel(BatchView b, DfoParams prm, const uint8_t* __restrict__ mask,
                                                      const double* __restrict__ limits,
                                                      const double* __restrict__ seg_times,
                                                      const double* __restrict__ coeffs, double* __restrict__ vec,
                                                      double* __restrict__ fvals, int32_t* __restrict__ state,
                                                      int32_t* __restrict__ fidx, long long* __restrict__ deadline) {
  const int p = blockIdx.x * 64 + threadIdx.x;
  if (p == 0) *deadline = prm.time_budget_ticks > 0 ? (long long)wall_clock64() + prm.time_budget_ticks : 0ll;
  if (p >= b.n_paths) return;
  const int s0 = b.seg_offsets[p], n = b.seg_offsets[p + 1] - s0, v0 = s0 + p, V = n + 1;
  const size_t NV = dfo_var_total(b.n_segments, b.n_paths);
  const DfoVectors w = dfo_vectors(vec, NV, dfo_var_offset(s0, p));
  const bool with_free = prm.mode >= 3;
  bool bad = false;
  for (int i = 0; i < n; ++i) {
    const double t = seg_times[s0 + i];
    if (t < kTimeLowerBound) bad = true;
    w.x[i] = t;
    w.lb[i] = kTimeLowerBound;
    w.ub[i] = DBL_MAX;
  }
  int n_var = n;
  if (with_free) {
    bad = false;  // the bounds are widened to contain the start point (:496-501)
    int n_free = 0;
    for (int u = 0; u < V * kHalf; ++u) {
      const bool fixed = mask[(size_t)v0 * kHalf + u] != 0;
      fidx[(size_t)v0 * kHalf + u] = fixed ? -1 : n_free;
      if (!fixed) ++n_free;
    }
    n_var = n + kD * n_free;
    for (int i = n; i < n_var; ++i) {
      w.lb[i] = -DBL_MAX;
      w.ub[i] = DBL_MAX;
    }
    // initial free constraints = derivatives of the linear solution at the vertices (getFreeConstraints)
    for (int v = 0; v < V; ++v) {
      const int seg = (v < n) ? v : n - 1;
      const double tt = (v < n) ? 0.0 : seg_times[s0 + n - 1];
      for (int k = 0; k < kHalf; ++k) {
        const int f = fidx[(size_t)(v0 + v) * kHalf + k];
        if (f < 0) continue;
        for (int dim = 0; dim < kD; ++dim)
          w.x[n + dim * n_free + f] = poly_derivative_at(coeffs + ((size_t)(s0 + seg) * kD + dim) * kN, k, tt);
      }
    }
    const double* lim = limits + (size_t)p * 9;
    for (int dim = 0; dim < kD; ++dim) {
      const int grp = (dim <= 1) ? 0 : (dim == 2 ? 1 : 2);
      for (int k = 1; k <= 3; ++k) {
        const double value = fabs(lim[(k - 1) * 3 + grp]);
        int counter = 0;
        for (int v = 0; v < V; ++v)
          for (int deriv = 0; deriv <= prm.derivative; ++deriv)
            if (!mask[(size_t)(v0 + v) * kHalf + deriv]) {
              if (deriv == k) {
                w.lb[n + dim * n_free + counter] = -value;
                w.ub[n + dim * n_free + counter] = value;
              }
              ++counter;
            }
      }
    }
  }
  for (int i = 0; i < n_var; ++i) {
    const double xi = w.x[i], ax = fabs(xi);
    w.h[i] = (with_free && ax <= DBL_EPSILON) ? 1e-13 : prm.initial_stepsize_rel * ax;
    if (with_free) {
      if (xi < w.lb[i]) w.lb[i] = xi;
      else if (xi > w.ub[i]) w.ub[i] = xi;
    }
    w.x0[i] = xi;
    w.best[i] = xi;
  }
  int32_t* st = state + (size_t)p * kDfoInts;
  st[0] = kDfoFirst;
  st[1] = 0;
  st[2] = 0;
  st[3] = 0;
  st[4] = 0;
  st[5] = bad ? -2 : -1;
  st[6] = bad ? 1 : 0;
  st[7] = n_var;
  st[8] = 0;
  st[9] = 0;
  st[10] = 0;
  fvals[p] = 0.0;
  fvals[b.n_paths + p] = 0.0;
}

// modes 3 / 4: updateSegmentTimes + setFreeConstraints (linear_impl.h:515-522, 264-282) for the trial point:
// thread = (segment, dimension); c = A^-1(T) d with d gathered from the fixed values and the trial free
// constraints, and this (segment, dimension)'s share of J_d = 1/2 d^T H(T) d into segcost.
__global__ __launch_bounds__(64) void dfo_free_eval_kernel(BatchView b, int d, const uint8_t* __restrict__ mask,
                                                           const double* __restrict__ vals,
                                                           const int32_t* __restrict__ fidx, const int32_t* __restrict__ state,
                                                           const double* __restrict__ vec, const int32_t* __restrict__ seg_path,
                                                           double* __restrict__ seg_times, double* __restrict__ coeffs,
                                                           double* __restrict__ segcost) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= b.n_segments) return;
  const int dim = blockIdx.y;
  const int p = seg_path[s];
  const int s0 = b.seg_offsets[p], n = b.seg_offsets[p + 1] - s0, i = s - s0, v0 = s0 + p;
  const double* x = vec + dfo_var_offset(s0, p);
  const int n_free = (state[(size_t)p * kDfoInts + 7] - n) / kD;
  const double T = x[i];
  if (dim == 0) seg_times[s] = T;
  const double t2 = T * T, t3 = t2 * T, t4 = t2 * t2;
  const double w[kHalf] = {1.0, T, t2, t3, t4};
  double db[kN];  // end-point derivatives in normalised time
#pragma unroll
  for (int r = 0; r < kN; ++r) {
    const size_t u = (size_t)(v0 + i + r / kHalf) * kHalf + (r % kHalf);
    const double val = mask[u] ? vals[u * kD + dim] : x[n + dim * n_free + fidx[u]];
    db[r] = val * w[r % kHalf];
  }
  const double ti = 1.0 / T;
  double tik = 1.0;
  double* c = coeffs + ((size_t)s * kD + dim) * kN;
#pragma unroll
  for (int k = 0; k < kN; ++k) {
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < kN; ++r) acc = fma(c_abar_inv[k][r], db[r], acc);
    c[k] = acc * tik;
    tik *= ti;
  }
  double q = 0.0;
#pragma unroll
  for (int r = 0; r < kN; ++r) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < kN; ++k) acc = fma(c_hbar[d][r][k], db[k], acc);
    q = fma(acc, db[r], q);
  }
  const double td = (d == 0) ? 1.0 : (d == 1) ? T : (d == 2) ? t2 : (d == 3) ? t3 : t4;
  segcost[(size_t)s * kD + dim] = 0.5 * q * (T / (td * td));
}

// consume the objective of the trial currently in x (its trajectory is in seg_times / coeffs), write the next trial
__global__ __launch_bounds__(64) void dfo_step_kernel(BatchView b, DfoParams prm, const double* __restrict__ limits,
                                                      const double* __restrict__ cost, const double* __restrict__ segcost,
                                                      const double* __restrict__ maxima4, double* __restrict__ seg_times,
                                                      double* __restrict__ vec, double* __restrict__ fvals,
                                                      int32_t* __restrict__ state,
                                                      const long long* __restrict__ deadline) {
  const int p = blockIdx.x * 64 + threadIdx.x;
  if (p >= b.n_paths) return;
  int32_t* st = state + (size_t)p * kDfoInts;
  if (st[6]) return;
  const int s0 = b.seg_offsets[p], nseg = b.seg_offsets[p + 1] - s0, n = st[7];
  const size_t NV = dfo_var_total(b.n_segments, b.n_paths);
  const DfoVectors w = dfo_vectors(vec, NV, dfo_var_offset(s0, p));
  double* x = w.x;
  // objective = J_d + time penalty + soft constraints
  double total = 0.0, jd = 0.0, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
  for (int i = 0; i < nseg; ++i) {
    total += x[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) mx[k] = fmax(mx[k], maxima4[(size_t)(s0 + i) * 3 + k]);
  }
  if (prm.mode >= 3) {
    for (int i = 0; i < nseg * kD; ++i) jd += segcost[(size_t)s0 * kD + i];
  } else {
    jd = cost[p];
  }
  const bool richter = prm.mode == 1 || prm.mode == 4;
  double f = jd + (richter ? total * prm.time_penalty : total * total * prm.time_penalty);
  if (prm.use_soft) {
    const double* lim = limits + (size_t)p * 9;
    double soft = 0.0;
#pragma unroll
    for (int dim = 0; dim < 4; ++dim) {
      const int grp = (dim <= 1) ? 0 : (dim == 2 ? 1 : 2);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double value = lim[k * 3 + grp];
        soft += fmin(1.0e12, exp((mx[k] - value) / value * prm.soft_weight));
      }
    }
    f += soft;
  }
  // ---- state machine (same transitions as oracle/mto_dfo.c::dfo_step: greedy coordinate search with step doubling /
  // halving whose last evaluation is its best point)
  const bool greedy = prm.mode < 3;  // modes 0 / 1: greedy coordinate search; 3 / 4: interpolation sweep + compass
  int phase = st[0], ci = st[1], sg = st[2], neval = st[3], improved = st[4], acc_any = st[8], pending = st[9],
      last_is_best = st[10];
  double fbest = fvals[p], f_sweep = fvals[b.n_paths + p];
  bool accepted = false;
  ++neval;
  int ret = -1;
  bool done = false;
  if (phase == kDfoRevisit) {  // back on the best point: that was the search's last evaluation
    ret = pending;
    done = true;
  } else {
    if (phase == kDfoFirst) {
      fbest = f;
      last_is_best = 1;
    } else if (greedy ? f < fbest - kDfoMinDecrease * fabs(fbest) : f < fbest) {
      fbest = f;
      for (int k = 0; k < n; ++k) w.best[k] = x[k];
      improved = 1;
      accepted = true;
      last_is_best = 1;
      if (greedy) w.h[ci] *= 2.0;
    } else {
      last_is_best = 0;
    }
    if (prm.max_iterations > 0 && neval >= prm.max_iterations) {
      ret = 5;
      done = true;
    } else if (deadline && *deadline != 0ll && (long long)wall_clock64() > *deadline) {
      ret = 6;  // nlopt maxtime
      done = true;
    }
  }
  while (!done) {
    int stop = 0;
    if (!greedy) {
      // modes 3 / 4: Powell's decision-free interpolation sweep x0 +- h_i e_i, then the compass search (oracle/mto_dfo.c,
      // dfo_step_sweep, and why these modes keep it)
      if (phase == kDfoFirst) {
        phase = kDfoInitPlus;
        ci = 0;
      } else if (phase == kDfoInitPlus) {
        if (++ci >= n) {
          phase = kDfoInitMinus;
          ci = 0;
        }
      } else if (phase == kDfoInitMinus) {
        if (++ci >= n) {
          phase = kDfoCompass;
          for (int k = 0; k < n; ++k) w.h[k] *= 0.5;
          ci = 0;
          sg = 0;
          f_sweep = fbest;
          improved = 0;
          accepted = false;
        }
      } else {
        if (sg == 0 && !accepted) {
          sg = 1;
        } else {
          sg = 0;
          ++ci;
        }
        accepted = false;
        if (ci >= n) {
          if (improved) {
            if (relstop(f_sweep, fbest, prm.f_rel, prm.f_abs)) stop = 3;
          } else {
            bool all_small = true;
            for (int k = 0; k < n; ++k) {
              w.h[k] *= 0.5;
              if (!(w.h[k] < prm.x_abs || w.h[k] < prm.x_rel * fabs(w.best[k]))) all_small = false;
            }
            if (all_small) stop = 4;
          }
          f_sweep = fbest;
          improved = 0;
          ci = 0;
          sg = 0;
        }
      }
    } else if (phase == kDfoFirst) {
      phase = kDfoSearch;
      ci = 0;
      sg = 0;
      acc_any = 0;
      improved = 0;
      f_sweep = fbest;
    } else if (accepted) {
      acc_any = 1;  // same coordinate, same direction, doubled step
    } else if (!acc_any && sg == 0) {
      sg = 1;  // the first + trial failed: the other direction
    } else {
      sg = 0;
      acc_any = 0;
      if (++ci >= n) {  // end of a sweep
        if (improved) {
          if (relstop(f_sweep, fbest, prm.f_rel, prm.f_abs)) stop = 3;
        } else {
          bool all_small = true;
          for (int k = 0; k < n; ++k) {
            w.h[k] *= 0.5;
            if (!(w.h[k] < prm.x_abs || w.h[k] < prm.x_rel * fabs(w.best[k]))) all_small = false;
          }
          if (all_small) stop = 4;
        }
        f_sweep = fbest;
        improved = 0;
        ci = 0;
      }
    }
    accepted = false;
    if (stop) {
      if (last_is_best) {
        ret = stop;
        done = true;
        break;
      }
      phase = kDfoRevisit;
      pending = stop;
      for (int k = 0; k < n; ++k) x[k] = w.best[k];
      break;
    }
    if (prm.max_iterations > 0 && neval >= prm.max_iterations - 1) {  // the budget's last evaluation belongs to the best point
      phase = kDfoRevisit;
      pending = 5;
      for (int k = 0; k < n; ++k) x[k] = w.best[k];
      break;
    }
    const double lo = w.lb[ci], hi = w.ub[ci], hc = w.h[ci];
    if (phase == kDfoInitPlus) {
      for (int k = 0; k < n; ++k) x[k] = w.x0[k];
      const double xc = w.x0[ci];
      x[ci] = fmin(fmax((xc + hc <= hi) ? xc + hc : xc - hc, lo), hi);
      break;
    }
    if (phase == kDfoInitMinus) {
      for (int k = 0; k < n; ++k) x[k] = w.x0[k];
      const double xc = w.x0[ci];
      x[ci] = fmin(fmax((xc - hc >= lo) ? xc - hc : xc + 2.0 * hc, lo), hi);
      break;
    }
    const double t = fmin(fmax(w.best[ci] + (sg == 0 ? hc : -hc), lo), hi);
    if (t == w.best[ci]) continue;  // nothing to try in this direction: as a failed trial
    for (int k = 0; k < n; ++k) x[k] = w.best[k];
    x[ci] = t;
    break;
  }
  if (!done && prm.mode < 3)
    for (int i = 0; i < nseg; ++i) seg_times[s0 + i] = x[i];  // modes 3 / 4: dfo_free_eval_kernel publishes them
  st[0] = phase;
  st[1] = ci;
  st[2] = sg;
  st[3] = neval;
  st[4] = improved;
  st[8] = acc_any;
  st[9] = pending;
  st[10] = last_is_best;
  if (done) {
    st[5] = ret;
    st[6] = 1;
  }
  fvals[p] = fbest;
  fvals[b.n_paths + p] = f_sweep;
}

// final status: -2 stays (position-free vertex), a rejected start surfaces as FAILURE -1, else the stopping reason
__global__ __launch_bounds__(64) void dfo_finalize_kernel(int n_paths, const int32_t* __restrict__ state,
                                                          int32_t* __restrict__ status) {
  const int p = blockIdx.x * 64 + threadIdx.x;
  if (p >= n_paths) return;
  if (status[p] == -2) return;
  const int ret = state[(size_t)p * kDfoInts + 5];
  status[p] = (ret == -2) ? -1 : ret;
}

__global__ __launch_bounds__(64) void dfo_sum_cost_kernel(BatchView b, const double* __restrict__ segcost,
                                                          double* __restrict__ cost) {
  const int p = blockIdx.x * 64 + threadIdx.x;
  if (p >= b.n_paths) return;
  double jd = 0.0;
  for (int i = b.seg_offsets[p] * kD; i < b.seg_offsets[p + 1] * kD; ++i) jd += segcost[i];
  cost[p] = jd;
}

// seg_path[s] = path of segment s (modes 3 / 4)
__global__ __launch_bounds__(64) void dfo_segment_path_kernel(BatchView b, int32_t* __restrict__ seg_path) {
  const int p = blockIdx.x * 64 + threadIdx.x;
  if (p >= b.n_paths) return;
  for (int s = b.seg_offsets[p]; s < b.seg_offsets[p + 1]; ++s) seg_path[s] = p;
}

// ---------------------------------------------------------------------------------------------
// host side

hipError_t launch_dfo(NonlinearPlan& nl, const BatchView& b, const DfoParams& prm, const uint8_t* mask, const double* vals,
                      const double* limits, double* seg_times, double* coeffs, int32_t* status, double* cost,
                      hipStream_t stream, bool general) {
  const KernelTimer kt = take_kernel_timer();  // family 2: the whole search, dfo_init_kernel to dfo_finalize_kernel
  if (b.n_paths == 0) return hipSuccess;
  hipError_t e = nonlinear_ensure_buffers(nl, b);
  if (e != hipSuccess) return e;
  const size_t nS = (size_t)(b.n_segments > 0 ? b.n_segments : 1), P = (size_t)b.n_paths;
  const size_t NV = dfo_var_total(b.n_segments, b.n_paths);
  const bool with_free = prm.mode >= 3;
  if (!nl.d_dfo_vec && (e = mrs_tg::pool_alloc(&nl.d_dfo_vec, sizeof(double) * 6 * NV)) != hipSuccess) return e;
  if (!nl.d_dfo_f && (e = mrs_tg::pool_alloc(&nl.d_dfo_f, sizeof(double) * 3 * P)) != hipSuccess) return e;
  if (!cost) cost = nl.d_dfo_f + 2 * P;  // J_d per evaluation needs a buffer even when the caller does not want it
  if (!nl.d_dfo_state && (e = mrs_tg::pool_alloc(&nl.d_dfo_state, sizeof(int32_t) * kDfoInts * P)) != hipSuccess) return e;
  if (!nl.d_dfo_deadline && (e = mrs_tg::pool_alloc(&nl.d_dfo_deadline, sizeof(long long))) != hipSuccess) return e;
  if (with_free) {
    if (!nl.d_dfo_fidx && (e = mrs_tg::pool_alloc(&nl.d_dfo_fidx, sizeof(int32_t) * (nS + P) * kHalf)) != hipSuccess) return e;
    if (!nl.d_dfo_segcost && (e = mrs_tg::pool_alloc(&nl.d_dfo_segcost, sizeof(double) * nS * kD)) != hipSuccess) return e;
    if (!nl.d_dfo_seg_path) {
      if ((e = mrs_tg::pool_alloc(&nl.d_dfo_seg_path, sizeof(int32_t) * nS)) != hipSuccess) return e;
      MRS_TG_LAUNCH(dfo_segment_path_kernel, dim3(cdiv_u(b.n_paths, 64)), dim3(64), 0, stream, b, nl.d_dfo_seg_path);
      if ((e = hipGetLastError()) != hipSuccess) return e;
    }
  }
  const unsigned pblocks = cdiv_u(b.n_paths, 64), sblocks = cdiv_u(b.n_segments, 64);
  // modes 3 / 4 start from the linear solution at the given times (optimizeTimeAndFreeConstraints :436-438);
  // for every mode this solve also marks position-free vertices (status -2, which stays unless the caller has switched
  // the general solve on: then every linear solve of the search is followed by the 5 x 5-block solve of those paths)
  if (general && (e = nonlinear_prepare_general(nl, b, mask, seg_times, false, nullptr, stream)) != hipSuccess) return e;
  const int32_t* general_flag = general ? nl.d_general + 4 : nullptr;
  if ((e = launch_solve_linear(b, prm.derivative, true, mask, vals, seg_times, nullptr, nullptr, nl.d_ws, coeffs, status,
                               cost, nullptr, stream)) != hipSuccess)
    return e;
  if (general && (e = launch_solve_general(b, prm.derivative, mask, vals, seg_times, nl.d_general_solve_ws, coeffs, status, cost,
                                           stream, general_flag, nullptr)) != hipSuccess)
    return e;
  MRS_TG_LAUNCH_EXT(dfo_init_kernel, dim3(pblocks), dim3(64), 0, stream, kt.start, nullptr, 0, b, prm, mask, limits, seg_times,
                        coeffs, nl.d_dfo_vec, nl.d_dfo_f, nl.d_dfo_state, nl.d_dfo_fidx, nl.d_dfo_deadline);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  // NLopt's maxeval <= 0 means "no limit"; the host loop needs one
  const int rounds = prm.max_iterations > 0 ? prm.max_iterations : 1000;
  for (int r = 0; r < rounds; ++r) {
    if (with_free) {
      MRS_TG_LAUNCH(dfo_free_eval_kernel, dim3(sblocks, kD), dim3(64), 0, stream, b, prm.derivative, mask, vals,
                         nl.d_dfo_fidx, nl.d_dfo_state, nl.d_dfo_vec, nl.d_dfo_seg_path, seg_times, coeffs,
                         nl.d_dfo_segcost);
      if ((e = hipGetLastError()) != hipSuccess) return e;
    } else if (r > 0) {  // round 0 evaluates the start point, solved above
      if ((e = launch_solve_linear(b, prm.derivative, true, mask, vals, seg_times, nullptr, nullptr, nl.d_ws, coeffs,
                                   nullptr, cost, nullptr, stream)) != hipSuccess)
        return e;
      if (general && (e = launch_solve_general(b, prm.derivative, mask, vals, seg_times, nl.d_general_solve_ws, coeffs, nullptr,
                                               cost, stream, general_flag, nullptr)) != hipSuccess)
        return e;
    }
    MRS_TG_LAUNCH(segment_maxima4_kernel, dim3(sblocks, 3), dim3(64), 0, stream, b.n_segments, coeffs, seg_times,
                       nl.d_maxima);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    MRS_TG_LAUNCH(dfo_step_kernel, dim3(pblocks), dim3(64), 0, stream, b, prm, limits, cost, nl.d_dfo_segcost,
                       nl.d_maxima, seg_times, nl.d_dfo_vec, nl.d_dfo_f, nl.d_dfo_state, nl.d_dfo_deadline);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  // paths that stopped early were re-evaluated at their final point every round; paths that used the whole budget
  // hold the trajectory of their last trial: both are "the last evaluated point"
  if (with_free) {  // J_d of the last evaluated point for the caller
    MRS_TG_LAUNCH(dfo_sum_cost_kernel, dim3(pblocks), dim3(64), 0, stream, b, nl.d_dfo_segcost, cost);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  MRS_TG_LAUNCH_EXT(dfo_finalize_kernel, dim3(pblocks), dim3(64), 0, stream, nullptr, kt.stop, 0, b.n_paths, nl.d_dfo_state,
                        status);
  return hipGetLastError();
}

}  // namespace mrs_tg
