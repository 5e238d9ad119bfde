// mrs_tg_rowelim.hpp -- the serial phase of the tile solve kernel: block-tridiagonal elimination of one path's reduced
// system R_pp d_p = -R_pf d_f (PolynomialOptimization::solveLinear,
// /root/reference/include/eth_trajectory_generation/impl/polynomial_optimization_linear_impl.h:341-373) by a
// column-per-lane LDL^T whose rank-one updates are single v_fmac_f64_dpp instructions.
//
// gfx950 executes the VOP1 / VOP2 double-precision instructions with the DPP control row_newbcast:N -- source operand 0
// is read from lane N of the lane's own row of 16 (scripts/dpp_probe.hip: same issue interval as a plain v_fma_f64,
// 9 instead of 6.5 cycles when dependent).  That makes one row of 16 lanes a systolic column store:
//
//   * lane c of a row holds COLUMN c of the symmetric band matrix (rows c-7 .. c+7, full storage) in registers A[row % 16]
//     and its four right-hand sides (x, y, z, heading) in B[0..3];
//   * pivot j:  inv = 1 / a_jj (broadcast from lane j);  every lane of the window (j, j+7] forms m_c = a_jc * inv from its
//     OWN register and applies  a_rc -= bcast_j(a_rj) * m_c  for the 7 rows of the band and the 4 right-hand sides:
//     one v_fmac_f64_dpp each, no data movement instruction at all;
//   * back substitution, column oriented: once x_r is final, every lane c in [r-7, r) does x_c -= l_rc * bcast_r(x_r).
//
// A vertex has four candidate unknowns (velocity .. snap), so a vertex is a quad of lanes and 16 lanes are a sliding
// window of four vertices over the chain: the quad a vertex uses, and the registers its rows use, are fixed by its
// DISTANCE w to the middle vertex (w mod 4), which keeps every lane number and register index a compile-time constant
// while paths of different length run in the same wavefront (a shorter path simply joins the loop later).  A wavefront
// works on two paths: rows 0 / 1 eliminate from the first vertex towards the middle one, rows 2 / 3 from the last
// vertex (two-sided elimination); the middle vertex receives both Schur updates (lane i <-> lane i + 32) and is solved
// redundantly by both sides, so no result has to travel back.  Up to four vertices per side stay in registers from
// their elimination to their back substitution; on longer paths the window slides and finished columns are parked in LDS.
//
// Against the block formulation of mrs_tg_tile.hip (one lane per (direction, dimension), every lane repeating the 4x4
// block algebra: ~220 in-order instructions per vertex) the four pivots of a vertex are ~100 instructions and its back
// substitution ~30.  The kernel built on this is mrs_tg_rows.hip.
//
// Hazards: a DPP read of a VGPR written by the preceding VALU instruction needs two wait states and the compiler's
// hazard recogniser does not look inside inline assembly, so every block below starts with s_nop 1.  A row_newbcast from
// a lane that EXEC has switched off does not deliver (the write is dropped, dpp_probe), so all 64 lanes stay active
// around every DPP instruction and inactive work is masked arithmetically (m = 0).
#pragma once
#include "mrs_tg_device.hpp"

namespace mrs_tg {

template <int LANE>
__device__ __forceinline__ double row_bcast(double x) {
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(LANE));
  return r;
}

#define MRS_TG_UPD(n) "v_fmac_f64_dpp %" #n ", -%" #n ", %[m] row_newbcast:%[ln] row_mask:0xf bank_mask:0xf\n\t"

// r_i -= bcast_LANE(r_i) * m for 4 .. 11 registers
template <int LANE>
__device__ __forceinline__ void row_update4(double m, double& r0, double& r1, double& r2, double& r3) {
  asm volatile("s_nop 1\n\t" MRS_TG_UPD(0) MRS_TG_UPD(1) MRS_TG_UPD(2) MRS_TG_UPD(3)
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3)
               : [m] "v"(m), [ln] "n"(LANE));
}
template <int LANE>
__device__ __forceinline__ void row_update5(double m, double& r0, double& r1, double& r2, double& r3, double& r4) {
  asm volatile("s_nop 1\n\t" MRS_TG_UPD(0) MRS_TG_UPD(1) MRS_TG_UPD(2) MRS_TG_UPD(3) MRS_TG_UPD(4)
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4)
               : [m] "v"(m), [ln] "n"(LANE));
}
template <int LANE>
__device__ __forceinline__ void row_update6(double m, double& r0, double& r1, double& r2, double& r3, double& r4,
                                            double& r5) {
  asm volatile("s_nop 1\n\t" MRS_TG_UPD(0) MRS_TG_UPD(1) MRS_TG_UPD(2) MRS_TG_UPD(3) MRS_TG_UPD(4) MRS_TG_UPD(5)
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5)
               : [m] "v"(m), [ln] "n"(LANE));
}
template <int LANE>
__device__ __forceinline__ void row_update7(double m, double& r0, double& r1, double& r2, double& r3, double& r4,
                                            double& r5, double& r6) {
  asm volatile("s_nop 1\n\t" MRS_TG_UPD(0) MRS_TG_UPD(1) MRS_TG_UPD(2) MRS_TG_UPD(3) MRS_TG_UPD(4) MRS_TG_UPD(5) MRS_TG_UPD(6)
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6)
               : [m] "v"(m), [ln] "n"(LANE));
}
template <int LANE>
__device__ __forceinline__ void row_update8(double m, double& r0, double& r1, double& r2, double& r3, double& r4,
                                            double& r5, double& r6, double& r7) {
  asm volatile("s_nop 1\n\t" MRS_TG_UPD(0) MRS_TG_UPD(1) MRS_TG_UPD(2) MRS_TG_UPD(3) MRS_TG_UPD(4) MRS_TG_UPD(5) MRS_TG_UPD(6)
                   MRS_TG_UPD(7)
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
               : [m] "v"(m), [ln] "n"(LANE));
}
template <int LANE>
__device__ __forceinline__ void row_update9(double m, double& r0, double& r1, double& r2, double& r3, double& r4,
                                            double& r5, double& r6, double& r7, double& r8) {
  asm volatile("s_nop 1\n\t" MRS_TG_UPD(0) MRS_TG_UPD(1) MRS_TG_UPD(2) MRS_TG_UPD(3) MRS_TG_UPD(4) MRS_TG_UPD(5) MRS_TG_UPD(6)
                   MRS_TG_UPD(7) MRS_TG_UPD(8)
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(r8)
               : [m] "v"(m), [ln] "n"(LANE));
}
template <int LANE>
__device__ __forceinline__ void row_update10(double m, double& r0, double& r1, double& r2, double& r3, double& r4,
                                             double& r5, double& r6, double& r7, double& r8, double& r9) {
  asm volatile("s_nop 1\n\t" MRS_TG_UPD(0) MRS_TG_UPD(1) MRS_TG_UPD(2) MRS_TG_UPD(3) MRS_TG_UPD(4) MRS_TG_UPD(5) MRS_TG_UPD(6)
                   MRS_TG_UPD(7) MRS_TG_UPD(8) MRS_TG_UPD(9)
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(r8), "+v"(r9)
               : [m] "v"(m), [ln] "n"(LANE));
}
template <int LANE>
__device__ __forceinline__ void row_update11(double m, double& r0, double& r1, double& r2, double& r3, double& r4,
                                             double& r5, double& r6, double& r7, double& r8, double& r9, double& r10) {
  asm volatile("s_nop 1\n\t" MRS_TG_UPD(0) MRS_TG_UPD(1) MRS_TG_UPD(2) MRS_TG_UPD(3) MRS_TG_UPD(4) MRS_TG_UPD(5) MRS_TG_UPD(6)
                   MRS_TG_UPD(7) MRS_TG_UPD(8) MRS_TG_UPD(9) MRS_TG_UPD(10)
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(r8), "+v"(r9),
                 "+v"(r10)
               : [m] "v"(m), [ln] "n"(LANE));
}
#undef MRS_TG_UPD

// 1 / x: hardware estimate (5e-8) + two Newton steps; a non-positive pivot is rejected (0: the variable stays at zero,
// see rsqrt_refined in mrs_tg_device.hpp)
__device__ __forceinline__ double pivot_reciprocal(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  return (x > 0.0) ? r : 0.0;
}

// One row of 16 lanes = one (path, direction): the register-resident part of the elimination.  Every index below is a
// compile-time constant after inlining, so the arrays live in VGPRs.
struct RowCore {
  double A[16];   // column entries, register = 4 * (distance of the row's vertex to the middle, mod 4) + slot
  double B[4];    // right-hand sides of this column per dimension; the solution after the back substitution
  double own_inv; // 1 / pivot of this lane's unknown; valid between the four pivots of its vertex and the scaling
  int quad, k;    // lane coordinates inside the row: quad = vertex of the window, k = slot
  int dir;        // 0: from vertex 0 towards the middle, 1: from vertex S

  // eliminate unknown (Q, K); NEXT: the window reaches into the next vertex (false for the middle vertex)
  template <int Q, int K, bool NEXT>
  __device__ __forceinline__ void pivot(bool act) {
    constexpr int J = 4 * Q + K, N = 4 * ((Q + 3) % 4);
    const double inv = pivot_reciprocal(row_bcast<J>(A[J]));
    own_inv = (k == K) ? inv : own_inv;  // (a select between values: selecting among array elements by k would turn
                                         // into a dynamically indexed load and push the whole struct to scratch memory)
    const bool inwin = (quad == Q && k > K) || (NEXT && quad == (Q + 3) % 4);
    const double m = (inwin && act) ? A[J] * inv : 0.0;
    if (NEXT) {
      if (K == 0) row_update11<J>(m, A[4 * Q + 1], A[N], A[N + 1], A[N + 2], A[N + 3], A[4 * Q + 2], A[4 * Q + 3], B[0], B[1], B[2], B[3]);
      else if (K == 1) row_update10<J>(m, A[4 * Q + 2], A[N], A[N + 1], A[N + 2], A[N + 3], A[4 * Q + 3], B[0], B[1], B[2], B[3]);
      else if (K == 2) row_update9<J>(m, A[4 * Q + 3], A[N], A[N + 1], A[N + 2], A[N + 3], B[0], B[1], B[2], B[3]);
      else row_update8<J>(m, A[N], A[N + 1], A[N + 2], A[N + 3], B[0], B[1], B[2], B[3]);
    } else {
      if (K == 0) row_update7<J>(m, A[4 * Q + 1], A[4 * Q + 2], A[4 * Q + 3], B[0], B[1], B[2], B[3]);
      else if (K == 1) row_update6<J>(m, A[4 * Q + 2], A[4 * Q + 3], B[0], B[1], B[2], B[3]);
      else if (K == 2) row_update5<J>(m, A[4 * Q + 3], B[0], B[1], B[2], B[3]);
      else row_update4<J>(m, B[0], B[1], B[2], B[3]);  // lanes outside the window: m = 0
    }
  }

  template <int Q>
  __device__ __forceinline__ void eliminate_vertex(bool act) {
    pivot<Q, 0, true>(act);
    pivot<Q, 1, true>(act);
    pivot<Q, 2, true>(act);
    pivot<Q, 3, true>(act);
  }

  // the vertex in quad Q is eliminated: its columns and right-hand sides are scaled by 1 / pivot (the back substitution
  // then needs no multiplication on its chain); `mine`: this lane belongs to that vertex
  template <int Q>
  __device__ __forceinline__ void scale_columns(bool mine) {
    if (mine) {
      constexpr int N = 4 * ((Q + 3) % 4);
      const double myinv = own_inv;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        A[4 * Q + j] *= myinv;
        A[N + j] *= myinv;
      }
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) B[dd] *= myinv;
    }
  }

  // x_r of unknown r = (Q, K) is final: x_c -= l_rc x_r for every column c whose band holds row r
  template <int Q, int K>
  __device__ __forceinline__ void back_row() {
    constexpr int J = 4 * Q + K;
    const bool inwin = (quad == Q && k < K) || quad == (Q + 1) % 4;
    const double t = inwin ? A[J] : 0.0;
    row_update4<J>(t, B[0], B[1], B[2], B[3]);
  }

  template <int Q>
  __device__ __forceinline__ void back_vertex() {
    back_row<Q, 3>();
    back_row<Q, 2>();
    back_row<Q, 1>();
    back_row<Q, 0>();
  }

  // the middle vertex (quad 0): both sides' Schur updates meet (lane i <-> lane i + 32), both sides solve it
  __device__ __forceinline__ void middle() {
    double ta[4], tb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ta[j] = __shfl_xor(A[j], 32, 64);
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) tb[dd] = __shfl_xor(B[dd], 32, 64);
    if (quad == 0) {  // the other quads hold the finished columns of the vertices around the middle
#pragma unroll
      for (int j = 0; j < 4; ++j) A[j] += ta[j];
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) B[dd] += tb[dd];
    }
    pivot<0, 0, false>(true);
    pivot<0, 1, false>(true);
    pivot<0, 2, false>(true);
    pivot<0, 3, false>(true);
    if (quad == 0) {
      const double myinv = own_inv;
#pragma unroll
      for (int j = 0; j < 4; ++j) A[j] *= myinv;
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) B[dd] *= myinv;
    }
  }
};

}  // namespace mrs_tg
