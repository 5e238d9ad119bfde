// mrs_tg_solve.hpp -- the full linear QP solve of one path (forward block-Cholesky sweep, backward
// substitution, coefficient recovery), shared by the linear kernel and the nonlinear kernel's re-solves.
//
// Reference: PolynomialOptimization::solveLinear + updateSegmentsFromCompactConstraints + computeCost
// (/root/reference/include/eth_trajectory_generation/impl/polynomial_optimization_linear_impl.h:341-373,
// 264-282, 128-141).
//
// A lane handles ND of the 4 dimensions of one path (ND = 4: one lane per path, best throughput for big
// batches; ND = 1: four lanes per path, each repeating the small matrix factorisation but carrying one
// right-hand side, ~4x shorter dependency chain for small batches).  All global loads that do not depend
// on the elimination chain (blocks, vertex constraints, times, saved factors) are issued one segment
// ahead so that their latency hides behind the arithmetic of the current segment.
#pragma once
#include "mrs_tg_device.hpp"

namespace mrs_tg {

template <int ND>
__host__ __device__ constexpr int ws_per_vertex() {
  return 10 + kNB * kNB + kNB * ND;  // L (packed), W, z
}

struct BlockSource {
  const double* H;  // slot-major SoA blocks (only read when FUSED == false)
  const double* A;
  size_t P;         // number of paths (SoA stride)
  int q;            // position of this path
};

template <int ND>
struct VertexData {
  double f[kHalf][ND];
  unsigned free_bits;
  bool pos_fixed;
};

template <int ND>
__device__ __forceinline__ void fetch_vertex(const uint8_t* __restrict__ mask, const double* __restrict__ vals, int v,
                                             int dim0, VertexData<ND>& out) {
  out.free_bits = load_vertex<ND>(mask, vals, v, dim0, out.f, out.pos_fixed);
}

// Addressing discipline for the SoA buffers: every address is (wave-uniform row pointer) + (32-bit lane
// index).  Written that way the compiler keeps the row pointers in SGPRs and emits
// global_load/store ... v_off, s[base] with one shared lane offset instead of 64-bit VALU address math
// per access.
__device__ __forceinline__ void load_H_blocks(const BlockSource& src, int seg, double (&Hs)[kSym10]) {
  const double* __restrict__ blk = src.H + (size_t)seg * 100 * src.P;
  const unsigned q = (unsigned)src.q;
#pragma unroll
  for (int a = 0; a < kN; ++a)
#pragma unroll
    for (int c = a; c < kN; ++c) Hs[sym10(a, c)] = (blk + (size_t)(a * kN + c) * src.P)[q];
}

// lower half (rows 5..9) of A^-1 and its diagonal upper half
struct AinvRows {
  double diag[kHalf];
  double low[kHalf][kN];
};

__device__ __forceinline__ void load_A_blocks(const BlockSource& src, int seg, AinvRows& a) {
  const double* __restrict__ blk = src.A + (size_t)seg * 100 * src.P;
  const unsigned q = (unsigned)src.q;
#pragma unroll
  for (int k = 0; k < kHalf; ++k) a.diag[k] = (blk + (size_t)(k * kN + k) * src.P)[q];
#pragma unroll
  for (int k = 0; k < kHalf; ++k)
#pragma unroll
    for (int c = 0; c < kN; ++c) a.low[k][c] = (blk + (size_t)((kHalf + k) * kN + c) * src.P)[q];
}

// The same loads for the tile kernel, whose lanes sit on different segments and paths: the row pointer depends only on
// the block entry (wave-uniform, SGPRs), the lane contributes one 32-bit BYTE offset ((seg * 100 * P + q) * 8 -- the
// tile kernel only runs when a block buffer is smaller than 4 GiB, tile_kernel_applies()), so each access is
// global_load ... v_off, s[base] instead of a 64-bit VALU address computation per access and lane.
__device__ __forceinline__ const double* at_byte_offset(const double* row, unsigned byte_off) {
  return reinterpret_cast<const double*>(reinterpret_cast<const char*>(row) + byte_off);
}

__device__ __forceinline__ void load_H_blocks32(const BlockSource& src, int seg, double (&Hs)[kSym10]) {
  const unsigned off = ((unsigned)seg * 100u * (unsigned)src.P + (unsigned)src.q) * 8u;
#pragma unroll
  for (int a = 0; a < kN; ++a)
#pragma unroll
    for (int c = a; c < kN; ++c) Hs[sym10(a, c)] = *at_byte_offset(src.H + (size_t)(a * kN + c) * src.P, off);
}

__device__ __forceinline__ void load_A_blocks32(const BlockSource& src, int seg, AinvRows& a) {
  const unsigned off = ((unsigned)seg * 100u * (unsigned)src.P + (unsigned)src.q) * 8u;
#pragma unroll
  for (int k = 0; k < kHalf; ++k) a.diag[k] = *at_byte_offset(src.A + (size_t)(k * kN + k) * src.P, off);
#pragma unroll
  for (int k = 0; k < kHalf; ++k)
#pragma unroll
    for (int c = 0; c < kN; ++c) a.low[k][c] = *at_byte_offset(src.A + (size_t)((kHalf + k) * kN + c) * src.P, off);
}

template <int ND>
struct SavedFactors {
  double L[10];
  double W[kNB][kNB];
  double z[kNB][ND];
};

// w: uniform pointer to element 0 of this vertex's record; element e of lane `lane` at w[e * stride + lane]
template <int ND>
__device__ __forceinline__ void store_factors(double* w, size_t stride, unsigned lane, const SavedFactors<ND>& s) {
#pragma unroll
  for (int e = 0; e < 10; ++e) (w + (size_t)e * stride)[lane] = s.L[e];
#pragma unroll
  for (int r = 0; r < kNB; ++r)
#pragma unroll
    for (int c = 0; c < kNB; ++c) (w + (size_t)(10 + r * kNB + c) * stride)[lane] = s.W[r][c];
#pragma unroll
  for (int r = 0; r < kNB; ++r)
#pragma unroll
    for (int c = 0; c < ND; ++c) (w + (size_t)(10 + kNB * kNB + r * ND + c) * stride)[lane] = s.z[r][c];
}

template <int ND>
__device__ __forceinline__ void load_factors(const double* w, size_t stride, unsigned lane, SavedFactors<ND>& s) {
#pragma unroll
  for (int e = 0; e < 10; ++e) s.L[e] = (w + (size_t)e * stride)[lane];
#pragma unroll
  for (int r = 0; r < kNB; ++r)
#pragma unroll
    for (int c = 0; c < kNB; ++c) s.W[r][c] = (w + (size_t)(10 + r * kNB + c) * stride)[lane];
#pragma unroll
  for (int r = 0; r < kNB; ++r)
#pragma unroll
    for (int c = 0; c < ND; ++c) s.z[r][c] = (w + (size_t)(10 + kNB * kNB + r * ND + c) * stride)[lane];
}

// Solve one path for dimensions [dim0, dim0 + ND).
//   times   segment times of this path (global or LDS)
//   ws      factor store (uniform base); element e of vertex v of lane `wlane` at ws[(v * ws_per_vertex<ND>() + e) * wstride + wlane]
//   coeffs  [S][4][10] of this path
// Returns this lane's share of the cost, 0.5 * (qf - red) over its dimensions.
template <int ND, bool FUSED>
__device__ __forceinline__ double solve_path(const uint8_t* __restrict__ mask, const double* __restrict__ vals, int v0,
                                             int S, int d, const double* times, int dim0, const BlockSource& src,
                                             double* ws, size_t wstride, unsigned wlane, double* __restrict__ coeffs,
                                             bool& pos_ok) {
  constexpr int WSV = ws_per_vertex<ND>();
  // one-segment-ahead prefetch of the materialised blocks doubles their register footprint: worth it in the
  // latency-bound 4-lanes-per-path mode, a spill hazard in the throughput mode (ND = 4) where other
  // wavefronts hide the latency anyway
  constexpr bool kPrefetchBlocks = (ND == 1);
  Elim<ND> st;
  st.init();
  VertexData<ND> vs, ve, vn;
  SavedFactors<ND> sf;
  fetch_vertex<ND>(mask, vals, v0, dim0, vs);
  fetch_vertex<ND>(mask, vals, v0 + 1, dim0, ve);
  pos_ok = vs.pos_fixed && ve.pos_fixed;
  double Hn[kSym10];
  double Tn = 0.0;
  if (FUSED) Tn = times[0];
  else if (kPrefetchBlocks) load_H_blocks(src, 0, Hn);

  for (int i = 0; i < S; ++i) {
    double Hs[kSym10];
    // issue the loads of the next segment before touching this segment's data
    if (i + 2 <= S) fetch_vertex<ND>(mask, vals, v0 + (i + 2 <= S ? i + 2 : S), dim0, vn);
    if (FUSED) {
      const double T = Tn;
      if (i + 1 < S) Tn = times[i + 1];
      hessian_from_time(T, d, Hs);
    } else if (kPrefetchBlocks) {
#pragma unroll
      for (int e = 0; e < kSym10; ++e) Hs[e] = Hn[e];
      if (i + 1 < S) load_H_blocks(src, i + 1, Hn);
    } else {
      load_H_blocks(src, i, Hs);
    }
    st.absorb_segment(Hs, vs.f, ve.f, vs.free_bits, ve.free_bits, sf.L, sf.z, sf.W);
    store_factors<ND>(ws + (size_t)i * WSV * wstride, wstride, wlane, sf);
    vs = ve;
    if (i + 2 <= S) {
      ve = vn;
      pos_ok = pos_ok && vn.pos_fixed;
    }
  }
  st.factor_vertex(vs.free_bits, sf.L, sf.z);
  const double cost = 0.5 * (st.qf - st.red);

  // backward sweep; vs holds vertex S
  double xn[kNB][ND], x[kNB][ND], dn[kHalf][ND], dc[kHalf][ND];
  back_substitute<ND>(sf.L, sf.z, sf.W, xn, true, x);
#pragma unroll
  for (int s = 0; s < kHalf; ++s)
#pragma unroll
    for (int q = 0; q < ND; ++q) dn[s][q] = vs.f[s][q] + (s >= kSlot0 ? x[s - kSlot0][q] : 0.0);
#pragma unroll
  for (int r = 0; r < kNB; ++r)
#pragma unroll
    for (int q = 0; q < ND; ++q) xn[r][q] = x[r][q];

  SavedFactors<ND> sn;
  VertexData<ND> vc, vp;
  AinvRows an;
  double Tb = 0.0;
  load_factors<ND>(ws + (size_t)(S - 1) * WSV * wstride, wstride, wlane, sn);
  fetch_vertex<ND>(mask, vals, v0 + S - 1, dim0, vp);
  if (FUSED) Tb = times[S - 1];
  else if (kPrefetchBlocks) load_A_blocks(src, S - 1, an);

  for (int i = S - 1; i >= 0; --i) {
    sf = sn;
    vc = vp;
    const double T = Tb;
    AinvRows ac;
    if (!FUSED) {
      if (kPrefetchBlocks) ac = an;
      else load_A_blocks(src, i, ac);
    }
    if (i > 0) {
      load_factors<ND>(ws + (size_t)(i - 1) * WSV * wstride, wstride, wlane, sn);
      fetch_vertex<ND>(mask, vals, v0 + i - 1, dim0, vp);
      if (FUSED) Tb = times[i - 1];
      else if (kPrefetchBlocks) load_A_blocks(src, i - 1, an);
    }
    back_substitute<ND>(sf.L, sf.z, sf.W, xn, false, x);
#pragma unroll
    for (int s = 0; s < kHalf; ++s)
#pragma unroll
      for (int q = 0; q < ND; ++q) dc[s][q] = vc.f[s][q] + (s >= kSlot0 ? x[s - kSlot0][q] : 0.0);
    double* cout = coeffs + (size_t)i * kD * kN;
#pragma unroll
    for (int q = 0; q < ND; ++q) {
      double c[kN];
      if (FUSED) {
        double dv[kN];
#pragma unroll
        for (int s = 0; s < kHalf; ++s) {
          dv[s] = dc[s][q];
          dv[kHalf + s] = dn[s][q];
        }
        coefficients_from_time(T, dv, c);
      } else {
        // c = A^-1 d with the materialised block: upper half diagonal, lower half dense
#pragma unroll
        for (int k = 0; k < kHalf; ++k) c[k] = ac.diag[k] * dc[k][q];
#pragma unroll
        for (int k = 0; k < kHalf; ++k) {
          double acc = 0.0;
#pragma unroll
          for (int s = 0; s < kHalf; ++s) acc += ac.low[k][s] * dc[s][q];
#pragma unroll
          for (int s = 0; s < kHalf; ++s) acc += ac.low[k][kHalf + s] * dn[s][q];
          c[kHalf + k] = acc;
        }
      }
#pragma unroll
      for (int k = 0; k < kN; ++k) cout[(dim0 + q) * kN + k] = c[k];
    }
#pragma unroll
    for (int s = 0; s < kHalf; ++s)
#pragma unroll
      for (int q = 0; q < ND; ++q) dn[s][q] = dc[s][q];
#pragma unroll
    for (int r = 0; r < kNB; ++r)
#pragma unroll
      for (int q = 0; q < ND; ++q) xn[r][q] = x[r][q];
  }
  return cost;
}

}  // namespace mrs_tg
