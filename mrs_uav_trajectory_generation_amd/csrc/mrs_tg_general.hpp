// mrs_tg_general.hpp -- one lane's share of the linear QP for ANY fixed / free pattern: the block-tridiagonal elimination with
// 5 x 5 vertex blocks (derivative slots 0..4, masks over all five), one lane = one (time vector, dimension), factors parked in
// global memory.  Used by solve_general_kernel (mrs_tg_general.hip: coefficients + cost at the given times) and by the outer
// loop's general evaluation (mrs_tg_nonlinear.hip: the cost at a perturbed time vector).
// setupConstraintReorderingMatrix /root/reference/include/eth_trajectory_generation/impl/polynomial_optimization_linear_impl.h:184-257,
// solveLinear :341-373, updateSegmentsFromCompactConstraints :264-282, computeCost :128-141.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "mrs_tg_device.hpp"

namespace mrs_tg {

constexpr int kGB = kHalf;                     // unknowns per vertex: derivative orders 0..4
constexpr int kGTri = kGB * (kGB + 1) / 2;     // packed lower triangle of a vertex block
constexpr int kGenWs = kGTri + kGB * kGB + kGB;  // L, W, z of one vertex

struct GenVertex {
  double f[kGB];   // constrained values (0 where free)
  unsigned free_bits;
};

__device__ __forceinline__ void gen_load_vertex(const uint8_t* __restrict__ mask, const double* __restrict__ vals, int v, int dim,
                                                GenVertex& out) {
  const uint8_t* __restrict__ mrow = mask + (size_t)v * kHalf;
  const double* __restrict__ vrow = vals + (size_t)v * kHalf * kD + dim;
  out.free_bits = 0u;
#pragma unroll
  for (int k = 0; k < kGB; ++k) {
    const bool fixed = mrow[k] != 0;
    out.f[k] = fixed ? vrow[k * kD] : 0.0;
    if (!fixed) out.free_bits |= 1u << k;
  }
}

// masked Cholesky of the vertex block, z = L^-1 y (constrained slots: unit row, zero right-hand side)
__device__ __forceinline__ void gen_factor(double (&Sm)[kGTri], double (&y)[kGB], unsigned free_bits, double (&L)[kGTri],
                                           double (&Linv)[kGB], double (&z)[kGB]) {
#pragma unroll
  for (int r = 0; r < kGB; ++r) {
    const bool fr = (free_bits >> r) & 1u;
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      const bool fc = (free_bits >> c) & 1u;
      const double v = Sm[tri(r, c)];
      Sm[tri(r, c)] = (r == c) ? (fr ? v : 1.0) : ((fr && fc) ? v : 0.0);
    }
    y[r] = fr ? y[r] : 0.0;
  }
#pragma unroll
  for (int c = 0; c < kGB; ++c) {
    double dsum = Sm[tri(c, c)];
#pragma unroll
    for (int m = 0; m < c; ++m) dsum = fma(-L[tri(c, m)], L[tri(c, m)], dsum);
    const double inv = rsqrt_refined(dsum);  // (a vanishing pivot leaves its variable at zero, see rsqrt_refined)
    L[tri(c, c)] = fmax(dsum * inv, 1.0e-300);
    Linv[c] = inv;
#pragma unroll
    for (int r = c + 1; r < kGB; ++r) {
      double s = Sm[tri(r, c)];
#pragma unroll
      for (int m = 0; m < c; ++m) s = fma(-L[tri(r, m)], L[tri(c, m)], s);
      L[tri(r, c)] = s * inv;
    }
  }
#pragma unroll
  for (int r = 0; r < kGB; ++r) {
    double s = y[r];
#pragma unroll
    for (int m = 0; m < r; ++m) s = fma(-L[tri(r, m)], z[m], s);
    z[r] = s * Linv[r];
  }
}

__device__ __forceinline__ double* gen_ws_at(double* ws, size_t stride, unsigned lane, int vertex, int e) {
  return ws + ((size_t)vertex * kGenWs + e) * stride + lane;
}


// The whole solve of one lane.  `time_of(i)` = time of segment i; `v0` = the path's first vertex row; `lane`, `stride` place
// the lane's factor store in `ws` (element e of vertex v at ws[(v * kGenWs + e) * stride + lane]).  STORE: coefficients of
// segment i, this dimension, to coeffs_path + (i * kD + dim) * kN.  Returns this dimension's share of 0.5 c^T Q c.
template <bool STORE, class TimeFn>
__device__ __forceinline__ double general_solve_lane(const uint8_t* __restrict__ mask, const double* __restrict__ vals, int v0,
                                                     int S, int d, int dim, TimeFn time_of, double* __restrict__ ws,
                                                     size_t stride, unsigned lane, double* __restrict__ coeffs_path) {
  double Sm[kGTri], y[kGB];
#pragma unroll
  for (int e = 0; e < kGTri; ++e) Sm[e] = 0.0;
#pragma unroll
  for (int r = 0; r < kGB; ++r) y[r] = 0.0;
  GenVertex vs, ve;
  gen_load_vertex(mask, vals, v0, dim, vs);
  for (int i = 0; i < S; ++i) {
    gen_load_vertex(mask, vals, v0 + i + 1, dim, ve);
    double Hs[kSym10];
    hessian_from_time(time_of(i), d, Hs);
    double u[kN];
#pragma unroll
    for (int a = 0; a < kN; ++a) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < kGB; ++c) s = fma(Hs[sym10(a, c)], vs.f[c], s);
#pragma unroll
      for (int c = 0; c < kGB; ++c) s = fma(Hs[sym10(a, kHalf + c)], ve.f[c], s);
      u[a] = s;
    }
#pragma unroll
    for (int r = 0; r < kGB; ++r) {
#pragma unroll
      for (int c = 0; c <= r; ++c) Sm[tri(r, c)] += Hs[sym10(r, c)];
      y[r] -= u[r];
    }
    double L[kGTri], Linv[kGB], z[kGB], W[kGB][kGB];
    gen_factor(Sm, y, vs.free_bits, L, Linv, z);
    // W = L^-1 E, E = coupling block restricted to (free here) x (free at the next vertex)
#pragma unroll
    for (int c = 0; c < kGB; ++c)
#pragma unroll
      for (int r = 0; r < kGB; ++r) {
        const bool on = ((vs.free_bits >> r) & 1u) && ((ve.free_bits >> c) & 1u);
        double s = on ? Hs[sym10(r, kHalf + c)] : 0.0;
#pragma unroll
        for (int m = 0; m < r; ++m) s = fma(-L[tri(r, m)], W[m][c], s);
        W[r][c] = s * Linv[r];
      }
#pragma unroll
    for (int e = 0; e < kGTri; ++e) *gen_ws_at(ws, stride, lane, i, e) = L[e];
#pragma unroll
    for (int r = 0; r < kGB; ++r)
#pragma unroll
      for (int c = 0; c < kGB; ++c) *gen_ws_at(ws, stride, lane, i, kGTri + r * kGB + c) = W[r][c];
#pragma unroll
    for (int r = 0; r < kGB; ++r) *gen_ws_at(ws, stride, lane, i, kGTri + kGB * kGB + r) = z[r];
    // state on the next vertex
#pragma unroll
    for (int r = 0; r < kGB; ++r) {
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        double s = Hs[sym10(kHalf + r, kHalf + c)];
#pragma unroll
        for (int m = 0; m < kGB; ++m) s = fma(-W[m][r], W[m][c], s);
        Sm[tri(r, c)] = s;
      }
      double s = -u[kHalf + r];
#pragma unroll
      for (int m = 0; m < kGB; ++m) s = fma(-W[m][r], z[m], s);
      y[r] = s;
    }
    vs = ve;
  }
  // last vertex
  double xn[kGB], dn[kGB];
  {
    double L[kGTri], Linv[kGB], z[kGB];
    gen_factor(Sm, y, vs.free_bits, L, Linv, z);
#pragma unroll
    for (int r = kGB - 1; r >= 0; --r) {
      double s = z[r];
#pragma unroll
      for (int m = r + 1; m < kGB; ++m) s = fma(-L[tri(m, r)], xn[m], s);
      xn[r] = s / L[tri(r, r)];
    }
#pragma unroll
    for (int k = 0; k < kGB; ++k) dn[k] = vs.f[k] + (((vs.free_bits >> k) & 1u) ? xn[k] : 0.0);
  }
  double total = 0.0;
  for (int i = S - 1; i >= 0; --i) {
    double L[kGTri], W[kGB][kGB], z[kGB];
#pragma unroll
    for (int e = 0; e < kGTri; ++e) L[e] = *gen_ws_at(ws, stride, lane, i, e);
#pragma unroll
    for (int r = 0; r < kGB; ++r)
#pragma unroll
      for (int c = 0; c < kGB; ++c) W[r][c] = *gen_ws_at(ws, stride, lane, i, kGTri + r * kGB + c);
#pragma unroll
    for (int r = 0; r < kGB; ++r) z[r] = *gen_ws_at(ws, stride, lane, i, kGTri + kGB * kGB + r);
    GenVertex vc;
    gen_load_vertex(mask, vals, v0 + i, dim, vc);
    double x[kGB], tt[kGB];
#pragma unroll
    for (int r = 0; r < kGB; ++r) {
      double s = z[r];
#pragma unroll
      for (int c = 0; c < kGB; ++c) s = fma(-W[r][c], xn[c], s);
      tt[r] = s;
    }
#pragma unroll
    for (int r = kGB - 1; r >= 0; --r) {
      double s = tt[r];
#pragma unroll
      for (int m = r + 1; m < kGB; ++m) s = fma(-L[tri(m, r)], x[m], s);
      x[r] = s / L[tri(r, r)];
    }
    double dv[kN], c[kN];
#pragma unroll
    for (int k = 0; k < kGB; ++k) {
      dv[k] = vc.f[k] + (((vc.free_bits >> k) & 1u) ? x[k] : 0.0);
      dv[kHalf + k] = dn[k];
    }
    const double T = time_of(i);
    coefficients_from_time(T, dv, c);
    double cb[kN];
    double tk = 1.0;
#pragma unroll
    for (int k = 0; k < kN; ++k) {
      if (STORE) coeffs_path[((size_t)i * kD + dim) * kN + k] = c[k];
      cb[k] = c[k] * tk;  // unit-time coefficients for the cost form
      tk *= T;
    }
    double p2[9];
    hessian_powers(T, d, p2);  // p2[0] = T^(1 - 2d)
    total = fma(cost_quadratic_form_d(d, cb), p2[0], total);
#pragma unroll
    for (int k = 0; k < kGB; ++k) {
      dn[k] = dv[k];
      xn[k] = ((vc.free_bits >> k) & 1u) ? x[k] : 0.0;
    }
  }
  return total;
}

}  // namespace mrs_tg
