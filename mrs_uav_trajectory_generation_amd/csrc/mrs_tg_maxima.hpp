// mrs_tg_maxima.hpp -- per-segment maxima of the derivative magnitudes (device functions; the kernels that call them are
// segment_maxima9_kernel in mrs_tg_nonlinear.hip and segment_maxima4_kernel in mrs_tg_dfo.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "mrs_tg_device.hpp"
#include "mrs_tg_nl_common.hpp"

namespace mrs_tg {

// ---------------------------------------------------------------------------------------------
// per-segment maxima of |p^(k)| over [0, T] for k = 1..3 and the groups {x,y}, {z}, {heading}.
//
// The reference finds every complex root of d/dt |p^(k)|^2 with Jenkins-Traub and keeps the real
// ones inside the segment plus both end points (segment.cpp:113-156, polynomial.cpp:36-63,
// rpoly_ak1.cpp).  Only the maximum VALUE is consumed (trajectory.cpp:625-642), so this path brackets
// the local maxima of m(tau)^2 = sum_dim q^(k)(tau)^2 on a uniform grid in normalised time tau = t/T
// (sign change + -> - of its derivative), polishes each with safeguarded Newton steps and takes the
// largest of {end points, grid points, polished maxima}.  Every candidate is a true value of the
// function, so the result never exceeds the exact maximum.

constexpr int kGridCells = 32;
constexpr int kPolishIters = 17;

// falling factorial j!/(j-k)! as a compile-time constant
__host__ __device__ constexpr double falling(int j, int k) {
  double v = 1.0;
  for (int n = 0; n < k; ++n) v *= (double)(j - n);
  return v;
}

// Coefficients of q^(K) (derivative K in normalised time) of NDIM dimensions, kept in registers; every loop below has
// compile-time bounds so nothing is indexed dynamically.  q^(K+1) and q^(K+2) come out of the same Horner pass (the nested
// recurrence p'' <- p'' t + p', p' <- p' t + p, p <- p t + c): one coefficient array per dimension instead of three -- 18
// instead of 48 doubles for the horizontal velocity -- which is what lets five wavefronts share a SIMD where three did
// (142 -> VGPRs); the kernel waits on dependent FMA chains and on its lanes' uneven Newton loops, and more resident
// wavefronts are what hides both.
template <int NDIM, int N0>
struct MagPolyBase {
  double d0[NDIM][N0];

  // m2 = sum q^(K)^2 ;  g = (1/2) d m2 / dtau
  __device__ __forceinline__ void eval(double tau, double& m2, double& g) const {
    m2 = 0.0;
    g = 0.0;
#pragma unroll
    for (int q = 0; q < NDIM; ++q) {
      double v0 = d0[q][N0 - 1], v1 = 0.0;
#pragma unroll
      for (int j = N0 - 2; j >= 0; --j) {
        v1 = fma(v1, tau, v0);
        v0 = fma(v0, tau, d0[q][j]);
      }
      m2 = fma(v0, v0, m2);
      g = fma(v0, v1, g);
    }
  }
  // the same plus dg = derivative of g (Newton)
  __device__ __forceinline__ void eval2(double tau, double& m2, double& g, double& dg) const {
    m2 = 0.0;
    g = 0.0;
    dg = 0.0;
#pragma unroll
    for (int q = 0; q < NDIM; ++q) {
      double v0 = d0[q][N0 - 1], v1 = 0.0, h2 = 0.0;  // h2 = q^(K+2) / 2
#pragma unroll
      for (int j = N0 - 2; j >= 0; --j) {
        h2 = fma(h2, tau, v1);
        v1 = fma(v1, tau, v0);
        v0 = fma(v0, tau, d0[q][j]);
      }
      m2 = fma(v0, v0, m2);
      g = fma(v0, v1, g);
      dg += fma(v1, v1, 2.0 * (v0 * h2));
    }
  }
};

template <int K, int NDIM>
struct MagPoly : MagPolyBase<NDIM, kN - K> {
  static constexpr int N0 = kN - K;

  __device__ __forceinline__ void init(const double (&cb)[NDIM][kN]) {
#pragma unroll
    for (int q = 0; q < NDIM; ++q)
#pragma unroll
      for (int j = 0; j < N0; ++j) this->d0[q][j] = cb[q][j + K] * falling(j + K, K);
  }
};

// The derivative order as DATA (lanes of one wavefront working on different k: the in-launch maxima of the pipeline kernel,
// mrs_tg_rows.hip): every lane carries the nine coefficients of k = 1, the ones a higher k does not have are 0.0.  A leading
// zero coefficient leaves Horner's recurrences at exactly 0.0 until the first real one arrives, so every value -- and with it
// every decision of the search below -- is bit for bit that of MagPoly<K, NDIM>.
template <int NDIM>
struct MagPolyAny : MagPolyBase<NDIM, kN - 1> {
  static constexpr int N0 = kN - 1;

  __device__ __forceinline__ void init(const double (&cb)[NDIM][kN], int K) {
#pragma unroll
    for (int q = 0; q < NDIM; ++q)
#pragma unroll
      for (int j = 0; j < N0; ++j) {
        const double a1 = cb[q][j + 1] * falling(j + 1, 1);
        const double a2 = (j + 2 < kN) ? cb[q][j + 2 < kN ? j + 2 : 0] * falling(j + 2, 2) : 0.0;
        const double a3 = (j + 3 < kN) ? cb[q][j + 3 < kN ? j + 3 : 0] * falling(j + 3, 3) : 0.0;
        this->d0[q][j] = (K == 1) ? a1 : (K == 2) ? a2 : a3;
      }
  }
};

// max over tau in [0,1] of sum_dim q^(K)(tau)^2.
// Pass 1 walks the grid and records, as a bit mask, the cells where g changes sign + -> - (a local
// maximum inside).  Pass 2 polishes the recorded cells.  Keeping the two apart matters on a 64-wide
// wavefront: lanes hold different polynomials, and polishing inside the grid loop would make every lane
// wait for a Newton loop in almost every cell.
// PARTS lanes share one polynomial: lane `part` takes the grid cells [part * 32 / PARTS, (part + 1) * 32 / PARTS) -- the same
// grid points and the same polished cells as one lane walking all 32, so the maximum over the PARTS lanes (taken by the
// caller) is the same number; the dependent work of a lane, and with it the time a wavefront waits for its slowest lane's
// Newton loops, is PARTS times shorter.
template <int PARTS, class Poly>
__device__ __forceinline__ double max_mag2_search(const Poly& mp, int part) {
  static_assert(kGridCells <= 32 && kGridCells % PARTS == 0, "cell mask is 32 bits");
  constexpr int kCells = kGridCells / PARTS;
  const double h = 1.0 / kGridCells;
  const int i0 = part * kCells;
  double m2, g;
  mp.eval(i0 * h, m2, g);  // (i0 = 0: tau = 0 exactly)
  double best = m2;
  double g_prev = g;
  unsigned cells = 0u;
  for (int i = i0 + 1; i <= i0 + kCells; ++i) {
    const double tau = (i == kGridCells) ? 1.0 : i * h;
    mp.eval(tau, m2, g);
    best = fmax(best, m2);
    if (g_prev > 0.0 && g <= 0.0) cells |= 1u << (i - 1);
    g_prev = g;
  }
  while (cells) {
    const int cell = __ffs(cells) - 1;
    cells &= cells - 1;
    // safeguarded Newton on g inside (cell*h, (cell+1)*h]
    double lo = cell * h, hi = (cell + 1 == kGridCells) ? 1.0 : (cell + 1) * h;
    double t = 0.5 * (lo + hi);
    // m2 is flat at its maximum: an abscissa error e costs ~ m2'' e^2 / 2, so |e| ~ 1e-8 already gives
    // the value to ~1e-16; the iteration cap bounds the slowest lane of the wavefront (pure bisection
    // from a 1/32 cell reaches 2e-7 after 17 halvings, i.e. a value error below 1e-12).
    for (int it = 0; it < kPolishIters; ++it) {
      double mm, gg, dd;
      mp.eval2(t, mm, gg, dd);
      best = fmax(best, mm);
      if (gg > 0.0) lo = t;
      else hi = t;
      // Newton step with a refined reciprocal instead of the IEEE division sequence (a step that is off by an ulp is
      // still a Newton step; the bracket test below safeguards it)
      double tn = (dd < 0.0) ? fma(-gg, rcp_refined(dd), t) : 0.5 * (lo + hi);
      if (!(tn > lo && tn < hi)) tn = 0.5 * (lo + hi);
      // the last evaluated abscissa is within |tn - t| of the stationary point, and the value error is quadratic in it:
      // 3e-7 leaves (m2''/m2) * 1e-13 / 2 < 1e-10 even for a peak as narrow as a grid cell (m2''/m2 ~ 1e3)
      if (fabs(tn - t) < 3e-7) break;
      t = tn;
    }
  }
  return best;
}

template <int K, int NDIM, int PARTS = 1>
__device__ __forceinline__ double max_mag2(const double (&cb)[NDIM][kN], int part = 0) {
  MagPoly<K, NDIM> mp;
  mp.init(cb);
  return max_mag2_search<PARTS>(mp, part);
}

// which = 3*(k-1) + group: maximum of |p^(k)| over [0, T] for one (k, group) of one segment
template <int PARTS = 1>
__device__ __forceinline__ double segment_maximum(const double* __restrict__ c, double T, int which, int part = 0) {
  const int k = which / 3 + 1, grp = which % 3;
  double tp = 1.0;
  const double ti = 1.0 / T;
  double scale = ti;
  if (k == 2) scale = ti * ti;
  else if (k == 3) scale = ti * ti * ti;
  double m2;
  if (grp == 0) {
    double cb[2][kN];
#pragma unroll
    for (int j = 0; j < kN; ++j) {
      cb[0][j] = c[0 * kN + j] * tp;
      cb[1][j] = c[1 * kN + j] * tp;
      tp *= T;
    }
    m2 = (k == 1) ? max_mag2<1, 2, PARTS>(cb, part) : (k == 2) ? max_mag2<2, 2, PARTS>(cb, part) : max_mag2<3, 2, PARTS>(cb, part);
  } else {
    double cb[1][kN];
    const int dim = (grp == 1) ? 2 : 3;
#pragma unroll
    for (int j = 0; j < kN; ++j) {
      cb[0][j] = c[dim * kN + j] * tp;
      tp *= T;
    }
    m2 = (k == 1) ? max_mag2<1, 1, PARTS>(cb, part) : (k == 2) ? max_mag2<2, 1, PARTS>(cb, part) : max_mag2<3, 1, PARTS>(cb, part);
  }
  if (PARTS == 4) {  // the four lanes of a quad share the polynomial
    m2 = fmax(m2, dpp_move<0xB1>(m2));
    m2 = fmax(m2, dpp_move<0x4E>(m2));
  }
  return sqrt(m2) * scale;
}

// The same number for a derivative order k = 1..3 that is data: NDIM = 2 is the horizontal group (dimensions 0 and 1), NDIM = 1
// the dimension `dim` (2: vertical, 3: heading).  PARTS = 2: lanes 2 j and 2 j + 1 share the polynomial, half the grid each.
template <int NDIM, int PARTS>
__device__ __forceinline__ double segment_maximum_any(const double* c, double T, int k, int dim, int part) {
  double tp = 1.0;
  const double ti = 1.0 / T;
  double scale = ti;
  if (k == 2) scale = ti * ti;
  else if (k == 3) scale = ti * ti * ti;
  double cb[NDIM][kN];
#pragma unroll
  for (int j = 0; j < kN; ++j) {
    if (NDIM == 2) {
      cb[0][j] = c[0 * kN + j] * tp;
      cb[NDIM - 1][j] = c[1 * kN + j] * tp;
    } else {
      cb[0][j] = c[dim * kN + j] * tp;
    }
    tp *= T;
  }
  MagPolyAny<NDIM> mp;
  mp.init(cb, k);
  double m2 = max_mag2_search<PARTS>(mp, part);
  if (PARTS == 2) m2 = fmax(m2, dpp_move<0xB1>(m2));
  return sqrt(m2) * scale;
}

// ---------------------------------------------------------------------------------------------
// An upper bound on max |q^(K)| over [0, 1] without a search: the largest Bernstein coefficient of q^(K) (a polynomial is a
// convex combination of its Bernstein coefficients on [0, 1]).  Used where the maxima only feed the feasibility scaling
// max(1, max / limit, ...) (trajectory.cpp:625-642): a (segment, k, group) whose bound is below its limit cannot move the
// scaling, whatever its exact maximum is, and its search is skipped (segment_maxima_scaling_kernel).
__host__ __device__ constexpr double binomial(int n, int k) {
  double v = 1.0;
  for (int i = 1; i <= k; ++i) v = v * (double)(n - k + i) / (double)i;
  return v;
}

// cb[j] = c_j T^j of one dimension -> bound on |q^(K)(tau)|, tau in [0, 1], plus an allowance for the rounding of the bound
// itself (1e-12 of the sum of the magnitudes that went into it)
template <int K>
__device__ __forceinline__ double bernstein_bound(const double (&cb)[kN]) {
  constexpr int n = kN - 1 - K;
  double a[n + 1], mag = 0.0;
#pragma unroll
  for (int j = 0; j <= n; ++j) {
    a[j] = cb[j + K] * falling(j + K, K);
    mag += fabs(a[j]);
  }
  double best = 0.0;
#pragma unroll
  for (int i = 0; i <= n; ++i) {
    double bi = 0.0;
#pragma unroll
    for (int j = 0; j <= i; ++j) bi = fma(binomial(i, j) / binomial(n, j), a[j], bi);
    best = fmax(best, fabs(bi));
  }
  return fma(1.0e-12, mag, best);
}

}  // namespace mrs_tg
