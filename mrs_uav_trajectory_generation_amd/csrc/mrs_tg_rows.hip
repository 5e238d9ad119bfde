// mrs_tg_rows.hip -- linear QP solve with one lane per unknown (the latency-optimised solve kernel for fixed times).
//
// What it replaces: PolynomialOptimization::updateSegmentTimes + constructR + solveLinear +
// updateSegmentsFromCompactConstraints + computeCost for every path of a batch
// (/root/reference/include/eth_trajectory_generation/impl/polynomial_optimization_linear_impl.h:289-304, 311-373,
// 264-282, 128-141).
//
// A wavefront owns two paths (one for small batches) from their constraints to their coefficients; no workgroup
// barrier, no other wavefront involved:
//   stage    one lane per vertex / segment: constraints -> LDS (constrained values, free factors); per segment the
//            powers of T the matrix entries need and the position difference of its end vertices;
//   build    one lane per unknown (vertex, slot): the lane forms ITS column of the reduced matrix R_pp and its four
//            right-hand sides directly from the time-normalised constants (mrs_tg_constants.h): column entries are
//            HBAR * T^p, the right-hand sides -H (p_start - p_end) (a constant polynomial costs nothing, so the position
//            columns of H differ by their sign only) -- nothing is assembled in memory;
//   solve    two-sided elimination, 16 lanes per (path, direction), rank-one updates as v_fmac_f64_dpp row broadcasts
//            (mrs_tg_rowelim.hpp); the four outermost vertices of a side are built by all quads at once, further
//            vertices (and the middle one) when their quad becomes free;
//   recover  one lane per (segment, dimension): c = A^-1 d, and the cost as the reference computes it, 0.5 c^T Q c
//            (no cancellation, unlike the elimination by-product 0.5 (f^T H f - sum y^2 / pivot)).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "mrs_tg_device.hpp"
#include "mrs_tg_launch.h"
#include "mrs_tg_maxima.hpp"
#include "mrs_tg_rowelim.hpp"
#include "mrs_tg_sampling.hpp"

#ifndef MRS_TG_ROWS_WAVES
#define MRS_TG_ROWS_WAVES 1
#endif

namespace mrs_tg {

// LDS records (doubles)
//   vertex : f[5][4] constrained end-point derivatives (0 where free) | free factors of the slots 1..4 (0.0 / 1.0) |
//            x[4][4] the solution of the slots 1..4 (exactly 0 on a constrained slot: identity row, zero right-hand side)
//            | parking space for the four eliminated columns of a long path (12 each)
//   segment: q[k] = T^(2 + k - 2d), k = 0..3 | T | pad | p_i - p_{i+1} per dimension
//   cost   : one partial per (segment, dimension)
constexpr int kRVtxDer = 0, kRVtxFree = 20, kRVtxX = 24, kRVtxPark = 40, kRVtxRec = 88;
constexpr int kRSegPow = 0, kRSegT = 4, kRSegDp = 6, kRSegRec = 10;

__host__ __device__ constexpr int rows_path_doubles(int S) {
  const int base = (S + 1) * kRVtxRec + S * kRSegRec + S * kD;
  return base + (base & 1) + 2;  // 16-byte aligned records, paths of a wavefront off each other's banks
}

// (LDS only: a fence over every address space also drains the global-memory counter -- s_waitcnt vmcnt(0) -- and the
// wavefront then sits out the write latency of the coefficients it has just stored before it may add up its cost or sample)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

#ifdef MRS_TG_ROWS_DEBUG
__device__ double g_rows_debug[64 * 24 * 2];
#endif

struct RowSolve : RowCore {
  int S, mid, nact;
  bool store_ok;   // false for the spare rows of a wavefront that holds one path
  bool general;    // some constrained derivative value of the path is non-zero (initial state): full right-hand sides
  int d;
  double* vtx0;
  double* seg0;
  // per-lane constants of the column (depend on slot and direction only)
  double cOwnN[4], cNext[4], cOwnP[4], cPrev[4], cRhsN, cRhsP;
  // the column being built
  double own[4], nxt[4], prv[4], rhs[4];

  __device__ __forceinline__ double* vtx_rec(int v) const { return vtx0 + v * kRVtxRec; }
  __device__ __forceinline__ double* seg_rec(int i) const { return seg0 + i * kRSegRec; }
  __device__ __forceinline__ int vertex_at(int w) const { return dir ? mid + w : mid - w; }

  // H(a, b) = HBAR_d[a][b] T^((a%5) + (b%5) + 1 - 2d).  The unknown (v, k) sits in row 1 + k of the segment that starts at
  // v and in row 6 + k of the segment that ends there; "N" is the segment towards the middle, "P" the one towards the
  // end this row started from.
  __device__ __forceinline__ void load_constants() {
    const double* hb = &c_hbar[d][0][0];
    const int row_s = 1 + k, row_e = 6 + k;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ss = (1 + j) * kN + row_s, ee = (6 + j) * kN + row_e;  // diagonal blocks: start half, end half
      const int fwd = row_s * kN + 6 + j;                                // (v, k) with slot j of the following vertex
      const int bwd = (1 + j) * kN + row_e;                              // (v, k) with slot j of the preceding vertex
      cOwnN[j] = hb[dir ? ee : ss];
      cOwnP[j] = hb[dir ? ss : ee];
      cNext[j] = hb[dir ? bwd : fwd];
      cPrev[j] = hb[dir ? fwd : bwd];
    }
    cRhsN = hb[(dir ? row_e : row_s) * kN];
    cRhsP = hb[(dir ? row_s : row_e) * kN];
  }

  // column of the unknown (vertex at distance w, slot k) -> own / nxt / prv / rhs
  __device__ __forceinline__ void build(int w) {
    const int v = min(max(vertex_at(w), 0), S);  // (a shorter path that shares the wavefront builds out of turn; unused)
    const int iN = dir ? v - 1 : v, iP = dir ? v : v - 1;
    const bool hasN = iN >= 0 && iN < S, hasP = iP >= 0 && iP < S;
    const double* sn = seg_rec(min(max(iN, 0), S - 1));
    const double* sp = seg_rec(min(max(iP, 0), S - 1));
    const double* vo = vtx_rec(v);
    const double* vn = vtx_rec(min(max(dir ? v - 1 : v + 1, 0), S));
    const double* vp = vtx_rec(min(max(dir ? v + 1 : v - 1, 0), S));
    double qN = sn[kRSegPow + k], qP = sp[kRSegPow + k];
    const double TN = sn[kRSegT], TP = sp[kRSegT];
    const double2 dN01 = *reinterpret_cast<const double2*>(sn + kRSegDp), dN23 = *reinterpret_cast<const double2*>(sn + kRSegDp + 2);
    const double2 dP01 = *reinterpret_cast<const double2*>(sp + kRSegDp), dP23 = *reinterpret_cast<const double2*>(sp + kRSegDp + 2);
    const double2 mo01 = *reinterpret_cast<const double2*>(vo + kRVtxFree), mo23 = *reinterpret_cast<const double2*>(vo + kRVtxFree + 2);
    const double2 mn01 = *reinterpret_cast<const double2*>(vn + kRVtxFree), mn23 = *reinterpret_cast<const double2*>(vn + kRVtxFree + 2);
    const double2 mp01 = *reinterpret_cast<const double2*>(vp + kRVtxFree), mp23 = *reinterpret_cast<const double2*>(vp + kRVtxFree + 2);
    const double mk = vo[kRVtxFree + k];
    qN = hasN ? qN : 0.0;  // a segment that does not exist contributes nothing
    qP = hasP ? qP : 0.0;
    const double mo[4] = {mo01.x, mo01.y, mo23.x, mo23.y}, mn[4] = {mn01.x, mn01.y, mn23.x, mn23.y},
                 mp[4] = {mp01.x, mp01.y, mp23.x, mp23.y};
    const double dN[4] = {dN01.x, dN01.y, dN23.x, dN23.y}, dP[4] = {dP01.x, dP01.y, dP23.x, dP23.y};
    double pN[4], pP[4];
    pN[0] = qN * TN;
    pP[0] = qP * TP;
#pragma unroll
    for (int j = 1; j < 4; ++j) {
      pN[j] = pN[j - 1] * TN;
      pP[j] = pP[j - 1] * TP;
    }
    // the backward direction brings only its Schur update to the middle vertex: its copy starts from zero
    const bool zero = (w == 0 && dir == 1);
    const double mkz = zero ? 0.0 : mk;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double both = fma(cOwnN[j], pN[j], cOwnP[j] * pP[j]);
      double o = (mkz * mo[j]) * both;
      if (j == k) o += zero ? 0.0 : 1.0 - mk;  // constrained slot: identity row (everything else in it is masked to zero)
      own[j] = o;
      nxt[j] = (mk * mn[j]) * (cNext[j] * pN[j]);
      prv[j] = (mk * mp[j]) * (cPrev[j] * pP[j]);
    }
    const double gN = -mkz * (cRhsN * qN), gP = -mkz * (cRhsP * qP);
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) rhs[dd] = fma(gN, dN[dd], gP * dP[dd]);
    if (general && !zero) {
      // constrained derivative values that are not zero (the initial state of a path that starts in motion,
      // src/mrs_trajectory_generation.cpp:946-957): the remaining columns of H f
      const int rowN = dir ? 6 + k : 1 + k, rowP = dir ? 1 + k : 6 + k;
      const int ownN = dir ? 5 : 0, othN = dir ? 0 : 5, ownP = dir ? 0 : 5, othP = dir ? 5 : 0;
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      for (int bb = 1; bb <= 4; ++bb) {
        const double hNo = c_hbar[d][rowN][ownN + bb] * pN[bb - 1], hNn = c_hbar[d][rowN][othN + bb] * pN[bb - 1];
        const double hPo = c_hbar[d][rowP][ownP + bb] * pP[bb - 1], hPp = c_hbar[d][rowP][othP + bb] * pP[bb - 1];
        for (int dd = 0; dd < 4; ++dd) {
          const double fo = vo[kRVtxDer + bb * kD + dd], fn = vn[kRVtxDer + bb * kD + dd], fp = vp[kRVtxDer + bb * kD + dd];
          acc[dd] += (hNo + hPo) * fo + hNn * fn + hPp * fp;
        }
      }
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) rhs[dd] -= mk * acc[dd];
    }
  }

  template <int Q>
  __device__ __forceinline__ void place(bool mine) {
    if (mine) {
      constexpr int N = 4 * ((Q + 3) % 4), P = 4 * ((Q + 1) % 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        A[4 * Q + j] = own[j];
        A[N + j] = nxt[j];
        A[P + j] = prv[j];
      }
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) B[dd] = rhs[dd];
    }
  }

  template <int Q>
  __device__ __forceinline__ void park(int w, bool mine) {
    if (mine && store_ok) {
      constexpr int N = 4 * ((Q + 3) % 4);
      double2* fr = reinterpret_cast<double2*>(vtx_rec(vertex_at(w)) + kRVtxPark + 12 * k);
      fr[0] = make_double2(A[4 * Q], A[4 * Q + 1]);
      fr[1] = make_double2(A[4 * Q + 2], A[4 * Q + 3]);
      fr[2] = make_double2(A[N], A[N + 1]);
      fr[3] = make_double2(A[N + 2], A[N + 3]);
      fr[4] = make_double2(B[0], B[1]);
      fr[5] = make_double2(B[2], B[3]);
    }
  }

  template <int Q>
  __device__ __forceinline__ void restore(int w, bool mine) {
    if (mine) {
      constexpr int N = 4 * ((Q + 3) % 4);
      const double2* fr = reinterpret_cast<const double2*>(vtx_rec(vertex_at(w)) + kRVtxPark + 12 * k);
      const double2 a = fr[0], b = fr[1], c = fr[2], e = fr[3], f = fr[4], g = fr[5];
      A[4 * Q] = a.x, A[4 * Q + 1] = a.y, A[4 * Q + 2] = b.x, A[4 * Q + 3] = b.y;
      A[N] = c.x, A[N + 1] = c.y, A[N + 2] = e.x, A[N + 3] = e.y;
      B[0] = f.x, B[1] = f.y, B[2] = g.x, B[3] = g.y;
    }
  }

  // forward step at distance w >= 1 (quad Q = w % 4).  REFILL (w >= 4): the quad is needed again for the vertex four
  // steps further in (the middle vertex at w = 4): its columns are parked and the new vertex is built in their place.
  template <int Q, bool REFILL>
  __device__ __forceinline__ void forward_step(int w) {
    const bool act = w <= nact;
    const bool mine = quad == Q && act;
    eliminate_vertex<Q>(act);
    scale_columns<Q>(mine);
    if (REFILL) {
      park<Q>(w, mine);
      build(w - 4);
      place<Q>(mine);
    }
  }

  // backward step at distance w (quad Q): hand this vertex's solution to the vertex one further out and to the own
  // lower slots; RESTORE: that vertex was parked.  Then x goes to LDS (the recover phase forms d = f + x).
  template <int Q, bool RESTORE>
  __device__ __forceinline__ void backward_step(int w) {
    if (RESTORE) restore<(Q + 1) % 4>(w + 1, quad == (Q + 1) % 4 && w + 1 <= nact);
    back_vertex<Q>();
    if (quad == Q && w <= nact && store_ok && (w > 0 || dir == 0)) {
      double2* xr = reinterpret_cast<double2*>(vtx_rec(vertex_at(w)) + kRVtxX + k * kD);
      xr[0] = make_double2(B[0], B[1]);
      xr[1] = make_double2(B[2], B[3]);
    }
  }

  template <bool REFILL>
  __device__ __forceinline__ void forward_dispatch(int w) {
    switch (w & 3) {
      case 0: forward_step<0, REFILL>(w); break;
      case 1: forward_step<1, REFILL>(w); break;
      case 2: forward_step<2, REFILL>(w); break;
      default: forward_step<3, REFILL>(w); break;
    }
  }

  // wmax = largest nact of the wavefront (wave-uniform)
  __device__ __forceinline__ void run(int wmax) {
#pragma unroll
    for (int j = 0; j < 16; ++j) A[j] = 0.0;
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) B[dd] = 0.0;
    own_inv = 0.0;
    // every quad builds the outermost vertex it is responsible for, all at once
    {
      int w0 = -1;
      if (quad <= nact) w0 = nact - ((nact - quad) & 3);  // largest w <= nact with w % 4 == quad
      if (quad == 0 && nact < 4) w0 = 0;                  // quad 0 starts with the middle vertex on a short side
      build(w0 < 0 ? 0 : w0);
      const bool have = w0 >= 0;
      place<0>(have && quad == 0);
      place<1>(have && quad == 1);
      place<2>(have && quad == 2);
      place<3>(have && quad == 3);
    }
    MRS_TG_PHASE_MARK(10);
#ifdef MRS_TG_ROWS_DEBUG
    if (blockIdx.x == 0) {
      for (int j = 0; j < 16; ++j) g_rows_debug[threadIdx.x * 24 + j] = A[j];
      for (int j = 0; j < 4; ++j) g_rows_debug[threadIdx.x * 24 + 16 + j] = B[j];
      g_rows_debug[threadIdx.x * 24 + 20] = nact;
      g_rows_debug[threadIdx.x * 24 + 21] = cOwnN[0];
      g_rows_debug[threadIdx.x * 24 + 22] = cNext[0];
      g_rows_debug[threadIdx.x * 24 + 23] = cRhsN;
    }
#endif
    for (int w = wmax; w > 4; --w) forward_dispatch<true>(w);
    if (wmax >= 4) forward_step<0, true>(4);
    if (wmax >= 3) forward_step<3, false>(3);
    if (wmax >= 2) forward_step<2, false>(2);
    if (wmax >= 1) forward_step<1, false>(1);
    MRS_TG_PHASE_MARK(11);
    middle();
    MRS_TG_PHASE_MARK(12);
#ifdef MRS_TG_ROWS_DEBUG
    if (blockIdx.x == 0) {
      for (int j = 0; j < 16; ++j) g_rows_debug[64 * 24 + threadIdx.x * 24 + j] = A[j];
      for (int j = 0; j < 4; ++j) g_rows_debug[64 * 24 + threadIdx.x * 24 + 16 + j] = B[j];
    }
#endif
    backward_step<0, false>(0);
    if (wmax >= 1) backward_step<1, false>(1);
    if (wmax >= 2) backward_step<2, false>(2);
    if (wmax >= 3) {
      if (wmax >= 4) backward_step<3, true>(3);
      else backward_step<3, false>(3);
    }
    for (int w = 4; w <= wmax; ++w) {
      if (w < wmax) {
        switch (w & 3) {
          case 0: backward_step<0, true>(w); break;
          case 1: backward_step<1, true>(w); break;
          case 2: backward_step<2, true>(w); break;
          default: backward_step<3, true>(w); break;
        }
      } else {
        switch (w & 3) {
          case 0: backward_step<0, false>(w); break;
          case 1: backward_step<1, false>(w); break;
          case 2: backward_step<2, false>(w); break;
          default: backward_step<3, false>(w); break;
        }
      }
    }
  }
};

// T^(2 + k - 2 d), k = 0..3, and T of a segment record
__device__ __forceinline__ void rows_segment_time(double* r, int d, double T) {
  // T^(2 - 2d) as the assembly kernel forms it: T / (T^d)^2 ... times T
  const double t2 = T * T;
  double td = 1.0;
  if (d == 1) td = T;
  else if (d == 2) td = t2;
  else if (d == 3) td = t2 * T;
  else if (d == 4) td = t2 * t2;
  double qk = t2 / (td * td);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    r[kRSegPow + kk] = qk;
    qk *= T;
  }
  r[kRSegT] = T;
}

// TAIL = 1: the launch scales the segment times first and / or samples afterwards (RowsTail); the plain solve (TAIL = 0) is
// its own instantiation, which the tail's code would otherwise cost 0.4 us at 1024 x 10.
// TAIL = 2 (RowsTail::maxima_in_launch): the closing stages of a time-allocation pipeline in ONE launch, for the small
// batches whose wavefronts hold one path -- solve at the incoming times, the per-segment maxima of that trajectory, the
// feasibility scaling, the solve at the scaled times (only if some time moved: the same solve on the same numbers
// otherwise), cost, status, sampling.  A workgroup is two wavefronts: the second one only helps with the maxima (3 S
// horizontal searches, two lanes each, in wavefront 0; 6 S vertical and heading searches in wavefront 1 -- the numbers of
// segment_maxima9_kernel, mrs_tg_maxima.hpp) and leaves.  Every other stage is the code of the separate launches on the
// same numbers; what goes away is two launches, the gaps between three, and two staging passes.
template <int TAIL>
__device__ __forceinline__ void solve_rows_body(const BatchView& b, int d, int ppw, int Smax, const uint8_t* __restrict__ mask,
                                                const double* __restrict__ vals,
                                                const double* seg_times /* may be tail.seg_times_out */,
                                                double* __restrict__ coeffs, int32_t* __restrict__ status,
                                                double* __restrict__ cost, const int32_t* __restrict__ status_in,
                                                const RowsTail& tail, int block) {
  extern __shared__ double lds[];
  // every kernel argument the staging pass reads is wanted in SGPRs HERE: left alone the compiler fetches them from the
  // kernarg segment where they are first used -- four s_load / s_waitcnt round trips strung through the staging code, each
  // an exposed scalar-cache miss on the only wavefront of its SIMD -- instead of in one batch at the top
  asm volatile("" ::"s"(mask), "s"(vals), "s"(seg_times), "s"(coeffs), "s"(status), "s"(cost), "s"(status_in), "s"(d), "s"(ppw),
               "s"(Smax), "s"(b.n_paths), "s"(b.uniform_S), "s"(b.order), "s"(b.seg_offsets));
  const int lane = threadIdx.x & 63;
  const int helper = (TAIL == 2) ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : 0;
  const int q0 = block * ppw;
  const int n_here = min(ppw, b.n_paths - q0);
  const int PS = rows_path_doubles(Smax);
  MRS_TG_PHASE_MARK(0);

  // row = (path of the wavefront, direction)
  const int pl = (lane >> 4) & 1;
  const int t = pl < n_here ? pl : n_here - 1;
  const PathRef pr = path_at(b, q0 + t);
  double* base = lds + (size_t)t * PS;
  double* vtx0 = base;
  double* seg0 = base + (size_t)(Smax + 1) * kRVtxRec;
  double* pc0 = seg0 + (size_t)Smax * kRSegRec;

  // the per-lane constants of the build are requested first: their latency hides behind the staging pass
  RowSolve rs;
  rs.dir = lane >> 5;
  rs.quad = (lane >> 2) & 3;
  rs.k = lane & 3;
  rs.d = d;
  rs.load_constants();

  // TAIL = 2: the path's limits and the search's stopping reason, for the scaling between the two solves
  // (parked in LDS behind the maxima: nine doubles held in registers across the solve made it spill)
  double* mx_lds = lds + (size_t)ppw * PS + (size_t)ppw * ((size_t)Smax * (kD * kN + 1)) + kSampleBuffer + kSampleBuffer / 4 + 2;
  int st_keep = -2;
  if (TAIL == 2 && !helper) {
    st_keep = __builtin_amdgcn_readfirstlane(tail.opt_status[__builtin_amdgcn_readfirstlane(pr.p)]);
    if (lane < 9) mx_lds[(size_t)Smax * 9 + lane] = tail.limits[(size_t)pr.p * 9 + lane];
  }

  // ---- stage: constraints and segment data of the wavefront's paths -> LDS
  bool pos_ok_lane = true, nonzero_lane = false;
  unsigned long long pos_bad[2] = {0ull, 0ull}, gen_any[2] = {0ull, 0ull};
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    if (tt >= n_here || helper) break;
    const int S_t = __builtin_amdgcn_readlane(pr.S, tt * 16), s0_t = __builtin_amdgcn_readlane(pr.s0, tt * 16),
              v0_t = __builtin_amdgcn_readlane(pr.v0, tt * 16);
    double* vb = lds + (size_t)tt * PS;
    double* sb = vb + (size_t)(Smax + 1) * kRVtxRec;
    pos_ok_lane = true;
    nonzero_lane = false;
    // the first round of the segment loop's loads (time, the two positions) is requested before the vertex loop: the
    // compiler otherwise issues them after the vertex loop's stores, and the time after the positions -- three trips to
    // memory in a row
    double T_first = 1.0, ps_first[2 * kD] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const int p_t = __builtin_amdgcn_readlane(pr.p, tt * 16);
    // TAIL with feasibility scaling: the segment's nine maxima, the path's nine limits and the search's stopping reason are
    // requested here as well (behind the segment's time they were two more trips to memory in a row)
    double mx_first[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, lim_first[9] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
    int st_first = -2;
    if (TAIL == 1 && tail.maxima && lane < S_t) {
      st_first = tail.opt_status[p_t];
#pragma unroll
      for (int e = 0; e < 9; ++e) {
        mx_first[e] = tail.maxima[(size_t)(s0_t + lane) * 9 + e];
        lim_first[e] = tail.limits[(size_t)p_t * 9 + e];
      }
    }
    if (lane < S_t) {
      T_first = seg_times[s0_t + lane];
      const double* ps = vals + (size_t)(v0_t + lane) * kHalf * kD;
#pragma unroll
      for (int dd = 0; dd < kD; ++dd) {
        ps_first[dd] = ps[dd];
        ps_first[kD + dd] = ps[kHalf * kD + dd];
      }
    }
    MRS_TG_PHASE_MARK(20);  // constants and the first segment round requested
    for (int v = lane; v <= S_t; v += 64) {
      double f[kHalf][kD];
      bool pos_fixed;
      const unsigned fb = load_vertex<kD>(mask, vals, v0_t + v, 0, f, pos_fixed);
      double* r = vb + (size_t)v * kRVtxRec;
#pragma unroll
      for (int kk = 0; kk < kHalf; ++kk)
#pragma unroll
        for (int dd = 0; dd < kD; ++dd) {
          r[kRVtxDer + kk * kD + dd] = f[kk][dd];
          if (kk >= 1) nonzero_lane = nonzero_lane || (f[kk][dd] != 0.0);
        }
#pragma unroll
      for (int j = 0; j < kNB; ++j) r[kRVtxFree + j] = (double)((fb >> j) & 1u);
#pragma unroll
      for (int e = 0; e < kNB * kD; ++e) r[kRVtxX + e] = 0.0;  // stays for a fully constrained end vertex, which nobody solves
      pos_ok_lane = pos_ok_lane && pos_fixed;
    }
    MRS_TG_PHASE_MARK(21);  // vertices staged
    // segment `i` of the path: its record from its time and the constrained positions of its two vertices
    auto stage_segment = [&](int i, double T, const double (&pp)[2 * kD], int opt_st, const double* mx, const double* lim) {
      if (TAIL == 1 && tail.maxima) {  // feasibility scaling of this segment (trajectory.cpp:625-657), then the solve at the scaled times
        if (opt_st != -2) T *= violation_scaling(mx, lim);
        tail.seg_times_out[s0_t + i] = T;
      }
      double* r = sb + (size_t)i * kRSegRec;
      rows_segment_time(r, d, T);
#pragma unroll
      for (int dd = 0; dd < kD; ++dd) r[kRSegDp + dd] = pp[dd] - pp[kD + dd];
    };
    // the first round works on what was requested in front of the vertex loop; written as a select inside one loop the
    // compiler loaded time and positions again (a second trip to memory behind the first, ~800 cycles on the only wavefront
    // of its SIMD)
    if (lane < S_t) stage_segment(lane, T_first, ps_first, st_first, mx_first, lim_first);
    for (int i = lane + 64; i < S_t; i += 64) {  // paths of more than 64 segments
      const double* ps = vals + (size_t)(v0_t + i) * kHalf * kD;  // constrained position of vertex i, then of i + 1
      double pp[2 * kD];
#pragma unroll
      for (int dd = 0; dd < kD; ++dd) {
        pp[dd] = ps[dd];
        pp[kD + dd] = ps[kHalf * kD + dd];
      }
      stage_segment(i, seg_times[s0_t + i], pp, (TAIL == 1 && tail.maxima) ? tail.opt_status[p_t] : -2,
                    tail.maxima + (size_t)(s0_t + i) * 9, tail.limits + (size_t)p_t * 9);
    }
    MRS_TG_PHASE_MARK(22);  // segments staged
    pos_bad[tt] = __ballot(!pos_ok_lane);
    gen_any[tt] = __ballot(nonzero_lane);
  }
  wave_lds_sync();
  MRS_TG_PHASE_MARK(1);

  // TAIL = 2: mx_lds = [Smax][9] maxima of the first solve's trajectory, behind the sampler's areas
#pragma clang loop unroll(disable)
  for (int pass = 0; pass < (TAIL == 2 ? 2 : 1); ++pass) {
    if (!helper) {
      // ---- build + solve
      {
        rs.store_ok = pl < n_here;
        rs.general = (t ? gen_any[1] : gen_any[0]) != 0ull;
        rs.S = pr.S;
        rs.mid = pr.S / 2;
        rs.vtx0 = vtx0;
        rs.seg0 = seg0;
        // vertices a side eliminates: [0, m) or (m, S]; a fully constrained end vertex (every rest-to-rest path has two)
        // has nothing to eliminate -- its rows are identity rows, its couplings are masked -- and is left out
        {
          const int len = rs.dir ? pr.S - rs.mid : rs.mid;
          const double* ef = vtx0 + (size_t)(rs.dir ? pr.S : 0) * kRVtxRec + kRVtxFree;
          const bool end_fixed = (ef[0] + ef[1] + ef[2] + ef[3]) == 0.0;
          rs.nact = (len > 0 && end_fixed) ? len - 1 : len;
        }
        int wmax = max(max(__builtin_amdgcn_readlane(rs.nact, 0), __builtin_amdgcn_readlane(rs.nact, 16)),
                       max(__builtin_amdgcn_readlane(rs.nact, 32), __builtin_amdgcn_readlane(rs.nact, 48)));
        rs.run(wmax);
      }
      wave_lds_sync();
      MRS_TG_PHASE_MARK(4);

      // ---- recover: coefficients c = A^-1 [d_i; d_{i+1}] and the cost share 0.5 c^T Q c per (segment, dimension)
      const int per_path = Smax * kD;
      for (int item = lane; item < n_here * per_path; item += 64) {
        const int tt = item >= per_path ? 1 : 0;
        const int r = item - tt * per_path;
        const int i = r >> 2, dim = r & 3;
        const int S_t = __builtin_amdgcn_readlane(pr.S, 0);
        const int S_u = __builtin_amdgcn_readlane(pr.S, 16);
        if (i >= (tt ? S_u : S_t)) continue;
        const int s0_t = tt ? __builtin_amdgcn_readlane(pr.s0, 16) : __builtin_amdgcn_readlane(pr.s0, 0);
        const double* vb = lds + (size_t)tt * PS;
        const double* sb = vb + (size_t)(Smax + 1) * kRVtxRec;
        double* pcb = const_cast<double*>(sb) + (size_t)Smax * kRSegRec;
        const double* vs = vb + (size_t)i * kRVtxRec;
        const double* ve = vs + kRVtxRec;
        const double T = sb[(size_t)i * kRSegRec + kRSegT];
        double dv[kN];
    #pragma unroll
        for (int kk = 0; kk < kHalf; ++kk) {
          dv[kk] = vs[kRVtxDer + kk * kD + dim];
          dv[kHalf + kk] = ve[kRVtxDer + kk * kD + dim];
          if (kk >= kSlot0) {
            dv[kk] += vs[kRVtxX + (kk - kSlot0) * kD + dim];
            dv[kHalf + kk] += ve[kRVtxX + (kk - kSlot0) * kD + dim];
          }
        }
        // c_k = T^-k sum_j ABAR_INV[k][j] T^(j%5) d_j ; cb_k = c_k T^k is what the cost needs
        double w[kHalf];
        w[0] = 1.0;
    #pragma unroll
        for (int kk = 1; kk < kHalf; ++kk) w[kk] = w[kk - 1] * T;
        double db[kN], cb[kN], c[kN];
    #pragma unroll
        for (int j = 0; j < kN; ++j) db[j] = dv[j] * w[j % kHalf];
        const double ti = 1.0 / T;
        double tik = 1.0;
    #pragma unroll
        for (int kk = 0; kk < kN; ++kk) {
          double s = 0.0;
          if (kk < kHalf) {
            s = c_abar_inv[kk][kk] * db[kk];  // the upper half of ABAR_INV is diag(1/k!)
          } else {
    #pragma unroll
            for (int j = 0; j < kN; ++j) s += c_abar_inv[kk][j] * db[j];
          }
          cb[kk] = s;
          c[kk] = s * tik;
          tik *= ti;
        }
        double* out = coeffs + ((size_t)(s0_t + i) * kD + dim) * kN;
    #pragma unroll
        for (int kk = 0; kk < kN; ++kk) out[kk] = c[kk];
        if ((TAIL && tail.sampling_dt > 0.0 && tail.samples) || TAIL == 2) {  // the sampler (the maxima searches) of this launch read them from LDS
          double* sc = lds + (size_t)ppw * PS + (size_t)tt * ((size_t)Smax * (kD * kN + 1)) + Smax + (size_t)r * kN;
    #pragma unroll
          for (int kk = 0; kk < kN; ++kk) sc[kk] = c[kk];
        }
        const double quad_form = cost_quadratic_form_d(d, cb);
        // T^(1 - 2d) = q[0] / T
        pcb[r] = quad_form * (sb[(size_t)i * kRSegRec + kRSegPow] * ti);
      }
      wave_lds_sync();
    }
    if (TAIL == 2 && pass == 0) {
      // ---- maxima of the trajectory just solved: which = 3 (k - 1) + group
      const int S_p = __builtin_amdgcn_readfirstlane(pr.S);
      const double* sb = lds + (size_t)(Smax + 1) * kRVtxRec;
      const double* sc = lds + (size_t)ppw * PS + Smax;
      __syncthreads();
      MRS_TG_PHASE_MARK(6);
      if (!helper) {  // horizontal: (k, segment) x two halves of the grid
        for (int t = lane; t < 6 * S_p; t += 64) {
          const int task = t >> 1, k = task / S_p, i = task - k * S_p;
          mx_lds[i * 9 + 3 * k] = segment_maximum_any<2, 2>(sc + (size_t)i * kD * kN, sb[(size_t)i * kRSegRec + kRSegT], k + 1, 0, t & 1);
        }
      } else {        // vertical and heading: (k, group, segment)
        for (int t = lane; t < 6 * S_p; t += 64) {
          const int kg = t / S_p, i = t - kg * S_p, k = kg >> 1, grp = 1 + (kg & 1);
          mx_lds[i * 9 + 3 * k + grp] = segment_maximum_any<1, 1>(sc + (size_t)i * kD * kN, sb[(size_t)i * kRSegRec + kRSegT], k + 1, 1 + grp, 0);
        }
      }
      __syncthreads();
      MRS_TG_PHASE_MARK(7);
      if (helper) return;
      // ---- feasibility scaling (trajectory.cpp:625-657): T_i <- s_i T_i; nothing moved = the solve above is the last one
      bool moved = false;
      for (int i = lane; i < S_p; i += 64) {
        double* r = const_cast<double*>(sb) + (size_t)i * kRSegRec;
        const double T = r[kRSegT];
        double Tn = T;
        if (st_keep != -2) Tn = T * violation_scaling(mx_lds + (size_t)i * 9, mx_lds + (size_t)Smax * 9);
        if (Tn != T) {
          moved = true;
          rows_segment_time(r, d, Tn);
          tail.seg_times_out[pr.s0 + i] = Tn;
        }
      }
      MRS_TG_PHASE_MARK(14);
      if (__ballot(moved) == 0ull) break;
      rs.load_constants();  // (again: held across the searches above they cost the registers the searches need)
      wave_lds_sync();
    }
  }

  // ---- cost and status: the 16 lanes of a path's first row add its partial costs in a fixed order
  if ((lane >> 5) == 0 && pl < n_here) {
    const int l16 = lane & 15;
    double s = 0.0;
    for (int e = l16; e < pr.S * kD; e += 16) s += pc0[e];
    // the path's total (scaled) time, for the runaway test of a pipeline's last solve
    double ts = 0.0;
    const bool runaway_test = TAIL && tail.sum_t0 != nullptr;
    if (runaway_test)
      for (int i = l16; i < pr.S; i += 16) ts += seg0[(size_t)i * kRSegRec + kRSegT];
    s += __shfl_xor(s, 8, 64);
    if (runaway_test) ts += __shfl_xor(ts, 8, 64);
    s += __shfl_xor(s, 4, 64);
    if (runaway_test) ts += __shfl_xor(ts, 4, 64);
    s += __shfl_xor(s, 2, 64);
    if (runaway_test) ts += __shfl_xor(ts, 2, 64);
    s += __shfl_xor(s, 1, 64);
    if (runaway_test) ts += __shfl_xor(ts, 1, 64);
    if (l16 == 0) {
      if (cost) cost[pr.p] = s;
      if (status) {
        int st = merge_status((pl ? pos_bad[1] : pos_bad[0]) == 0ull, status_in, pr.p);
        if (runaway_test && st > 0 && ts > MRS_TG_RUNAWAY_TIME_FACTOR * tail.sum_t0[pr.p]) st = MRS_TG_STATUS_ROUNDOFF_LIMITED;
        status[pr.p] = st;
      }
    }
  }
  MRS_TG_PHASE_MARK(5);

  // ---- sample (sampleWholeTrajectory): the wavefront walks its paths one after the other
  if (TAIL && tail.sampling_dt > 0.0) {
    double* samp = lds + (size_t)ppw * PS;                       // per path: [Smax] times | [Smax][4][10] coefficients
    double* s_t = samp + (size_t)ppw * ((size_t)Smax * (kD * kN + 1));   // sample buffer of the wavefront
    unsigned short* s_seg = reinterpret_cast<unsigned short*>(s_t + kSampleBuffer);
    for (int tt = 0; tt < n_here; ++tt) {
      const int S_t = __builtin_amdgcn_readlane(pr.S, tt * 16), p_t = __builtin_amdgcn_readlane(pr.p, tt * 16);
      double* sT = samp + (size_t)tt * ((size_t)Smax * (kD * kN + 1));
      const double* sb = lds + (size_t)tt * PS + (size_t)(Smax + 1) * kRVtxRec;
      for (int i = lane; i < S_t; i += 64) sT[i] = sb[(size_t)i * kRSegRec + kRSegT];
      wave_lds_sync();
      double* out = tail.samples ? tail.samples + (size_t)p_t * tail.sample_capacity * kD : nullptr;
      const int n = sample_path_walk(sT, sT + Smax, s_t, s_seg, S_t, tail.sampling_dt, tail.sample_capacity, out, tail.sample_acc,
                                     tail.sample_acc_n);
      if (lane == 0 && tail.n_samples) tail.n_samples[p_t] = n;
      wave_lds_sync();
    }
  }
  MRS_TG_PHASE_MARK(8);
}

template <int TAIL>
__global__ __launch_bounds__(64, MRS_TG_ROWS_WAVES) void solve_rows_kernel(BatchView b, int d, int ppw, int Smax,
                                                        const uint8_t* __restrict__ mask, const double* __restrict__ vals,
                                                        const double* seg_times, double* __restrict__ coeffs,
                                                        int32_t* __restrict__ status, double* __restrict__ cost,
                                                        const int32_t* __restrict__ status_in, RowsTail tail) {
  solve_rows_body<TAIL>(b, d, ppw, Smax, mask, vals, seg_times, coeffs, status, cost, status_in, tail, (int)blockIdx.x);
}

// the closing stages of a pipeline in one launch (TAIL = 2 above): two wavefronts per path
__global__ __launch_bounds__(128, 2) void solve_rows_pipeline_kernel(BatchView b, int d, int Smax, const uint8_t* __restrict__ mask,
                                                                     const double* __restrict__ vals, const double* seg_times,
                                                                     double* __restrict__ coeffs, int32_t* __restrict__ status,
                                                                     double* __restrict__ cost,
                                                                     const int32_t* __restrict__ status_in, RowsTail tail) {
  solve_rows_body<2>(b, d, 1, Smax, mask, vals, seg_times, coeffs, status, cost, status_in, tail, (int)blockIdx.x);
}

// Several batches of ONE plan (same structure, their own input / output arrays) in one launch: workgroups
// [j * blocks_per_batch, (j + 1) * blocks_per_batch) solve batch j.  Every path runs the instructions of the single-batch
// kernel; what changes is the number of dispatches a host that keeps several batches in flight has to issue.
__global__ __launch_bounds__(64, MRS_TG_ROWS_WAVES) void solve_rows_group_kernel(BatchView b, int d, int ppw, int Smax,
                                                                                  RowsGroup g, int blocks_per_batch) {
  const int j = __builtin_amdgcn_readfirstlane((int)blockIdx.x / blocks_per_batch);
  solve_rows_body<0>(b, d, ppw, Smax, g.mask[j], g.vals[j], g.seg_times[j], g.coeffs[j], g.status[j], g.cost[j], nullptr,
                         RowsTail(), (int)blockIdx.x - j * blocks_per_batch);
}

// ---------------------------------------------------------------------------------------------
// launcher

static constexpr size_t kRowsLdsBudget = 144 * 1024;

// LDS of one wavefront: the records of its paths, and for a launch that samples the times and coefficients of those paths
// plus one sample buffer
static size_t rows_lds_bytes(int Smax, int ppw, bool sampling, bool maxima = false) {
  size_t doubles = (size_t)ppw * rows_path_doubles(Smax);
  if (sampling || maxima) doubles += (size_t)ppw * (size_t)Smax * (kD * kN + 1) + kSampleBuffer + kSampleBuffer / 4 + 2;
  if (maxima) doubles += (size_t)Smax * 9 + 10;  // + the path's nine limits
  return doubles * sizeof(double);
}

bool rows_kernel_applies(const BatchView& b, bool with_sampling) {
  if (b.n_paths == 0) return false;
  // tuning / test knob, read once per process (a getenv per launch is a measurable share of a 3 us launch):
  // MRS_TG_ROWS_KERNEL=0 falls back to the tile and lane kernels
  static const bool disabled = [] {
    const char* e = std::getenv("MRS_TG_ROWS_KERNEL");
    return e != nullptr && std::atoi(e) == 0;
  }();
  if (disabled) return false;
  return rows_lds_bytes(b.max_segments, 1, with_sampling) <= kRowsLdsBudget;
}

bool rows_tail_sampling_pays(const BatchView& b) { return b.n_paths <= 2048 && rows_kernel_applies(b, true); }

bool rows_pipeline_applies(const BatchView& b) {
  static const bool on = [] {  // MRS_TG_ROWS_PIPELINE=0: the separate launches (tuning / test knob, read once per process)
    const char* e = std::getenv("MRS_TG_ROWS_PIPELINE");
    return e == nullptr || std::atoi(e) != 0;
  }();
  // up to one solving wavefront per SIMD (its helper shares the SIMD of another path's solver): 1024 x 10 pipeline 106.6 ->
  // 101.0 us; at 2048 paths two solvers share every SIMD and the separate launches win (153 vs 169 us)
  // (long paths: the separate launches again -- one 80-segment request 0.399 -> 0.379 ms, 64 x 80 0.706 -> 0.682, 48 / 49 segments
  // on either side of a first threshold 0.182 / 0.176, equal at 30
  // segments; the results are the same bits either way, tests/test_gpu_pipeline_shortcuts.py)
  return on && b.n_paths > 0 && b.n_paths <= 1024 && b.max_segments <= 32 &&
         rows_lds_bytes(b.max_segments, 1, true, true) <= kRowsLdsBudget;
}

hipError_t launch_solve_rows(const BatchView& b, int d, const uint8_t* mask, const double* vals, const double* seg_times,
                             double* coeffs, int32_t* status, double* cost, const int32_t* status_in, hipStream_t stream,
                             const RowsTail& tail) {
  const bool sampling = tail.sampling_dt > 0.0;
  RowsTail tail_k = tail;  // (the copy the kernel gets: with the sampling walk's table filled in)
  AccPin pin;  // (released when this function returns: behind the enqueue of the kernel that reads the table)
  if (sampling && !dry_run()) {
    hipError_t et = sample_acc_table(tail.sampling_dt, tail.sample_capacity, stream, &tail_k.sample_acc, &tail_k.sample_acc_n, &pin);
    if (et != hipSuccess) return et;
  }
  if (tail.maxima_in_launch) {
    if (!tail.limits || !tail.opt_status || !tail.seg_times_out) return hipErrorInvalidValue;
    const size_t lds_bytes = rows_lds_bytes(b.max_segments, 1, true, true);
    if (lds_bytes > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)solve_rows_pipeline_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)kRowsLdsBudget);
      if (e != hipSuccess) return e;
    }
    MRS_TG_LAUNCH_TIMED(solve_rows_pipeline_kernel, dim3((unsigned)b.n_paths), dim3(128), lds_bytes, stream, b, d, b.max_segments,
                        mask, vals, seg_times, coeffs, status, cost, status_in, tail_k);
    return hipGetLastError();
  }
  // one path per wavefront while that still leaves SIMDs idle (256 CUs x 4); two paths per wavefront otherwise -- and
  // when the caller says other batches share the device: at 255 VGPRs a SIMD holds two wavefronts, so 1024 one-path
  // wavefronts per launch let two launches run side by side, 512 two-path wavefronts four (1024 x 10, four streams:
  // 4.9 -> 4.1 us per step; alone on the device the two-path launch is 0.9 us slower)
  int ppw = (b.n_paths <= 2048 && !(shared_device_hint() && !sampling)) ? 1 : 2;
  static const int forced_ppw = [] {  // MRS_TG_ROWS_PPW=1|2, read once per process
    const char* e = std::getenv("MRS_TG_ROWS_PPW");
    return e ? (std::atoi(e) == 1 ? 1 : 2) : 0;
  }();
  if (forced_ppw) ppw = forced_ppw;
  if (rows_lds_bytes(b.max_segments, 2, sampling) > kRowsLdsBudget) ppw = 1;
  const size_t lds_bytes = rows_lds_bytes(b.max_segments, ppw, sampling);
  const bool with_tail = sampling || tail.maxima != nullptr;
  if (lds_bytes > 64 * 1024) {  // beyond the default limit of a launch: raise it (a driver call, so only when needed)
    hipError_t e = hipFuncSetAttribute(with_tail ? (const void*)solve_rows_kernel<1> : (const void*)solve_rows_kernel<0>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRowsLdsBudget);
    if (e != hipSuccess) return e;
  }
  const unsigned grid = (unsigned)((b.n_paths + ppw - 1) / ppw);
  if (with_tail)
    MRS_TG_LAUNCH_TIMED(solve_rows_kernel<1>, dim3(grid), dim3(64), lds_bytes, stream, b, d, ppw, b.max_segments, mask, vals,
                        seg_times, coeffs, status, cost, status_in, tail_k);
  else
    MRS_TG_LAUNCH_TIMED(solve_rows_kernel<0>, dim3(grid), dim3(64), lds_bytes, stream, b, d, ppw, b.max_segments, mask, vals,
                        seg_times, coeffs, status, cost, status_in, tail);
  return hipGetLastError();
}

hipError_t launch_solve_rows_group(const BatchView& b, int d, const RowsGroup& g, hipStream_t stream) {
  if (g.n < 1 || g.n > kRowsGroupMax) return hipErrorInvalidValue;
  // two paths per wavefront as soon as the launch carries more than one small batch: a host that groups launches keeps
  // several of them in flight (on alternating streams), and at 255 VGPRs a SIMD holds two wavefronts (launch_solve_rows)
  int ppw = ((long long)b.n_paths * g.n <= 1024) ? 1 : 2;
  static const int forced_ppw = [] {  // MRS_TG_ROWS_PPW=1|2, read once per process
    const char* e = std::getenv("MRS_TG_ROWS_PPW");
    return e ? (std::atoi(e) == 1 ? 1 : 2) : 0;
  }();
  if (forced_ppw) ppw = forced_ppw;
  if (rows_lds_bytes(b.max_segments, 2, false) > kRowsLdsBudget) ppw = 1;
  const size_t lds_bytes = rows_lds_bytes(b.max_segments, ppw, false);
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)solve_rows_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRowsLdsBudget);
    if (e != hipSuccess) return e;
  }
  const int per_batch = (b.n_paths + ppw - 1) / ppw;
  MRS_TG_LAUNCH_TIMED(solve_rows_group_kernel, dim3((unsigned)(per_batch * g.n)), dim3(64), lds_bytes, stream, b, d, ppw,
                      b.max_segments, g, per_batch);
  return hipGetLastError();
}

}  // namespace mrs_tg
