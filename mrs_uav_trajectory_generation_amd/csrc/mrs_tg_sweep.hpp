// mrs_tg_sweep.hpp -- the forward cost sweep's building blocks, shared by the outer-loop kernels (mrs_tg_nonlinear.hip: the
// sweeping / two-wavefront / lean kernels; mrs_tg_wave.hip: one wavefront per path): perturbed time vectors of the Mellinger
// gradient, LDS staging of a path's vertices and segments, the specialised segment steps (FastStep), the direction tables.
//
// Reference behaviour (relative to /root/reference/include/eth_trajectory_generation/impl/):
//   polynomial_optimization_nonlinear_impl.h:257-333  getCostAndGradientMellinger (the S + 1 time vectors of an evaluation)
//   polynomial_optimization_linear_impl.h:311-373     constructR + solveLinear (here: block Cholesky over the vertex chain)
#pragma once
#include <hip/hip_runtime.h>

#include "mrs_tg_device.hpp"
#include "mrs_tg_nl_common.hpp"

namespace mrs_tg {

constexpr int kLbfgsM = 5;
constexpr double kGradStep = 0.1;  // increment_time, nonlinear_impl.h:281

// ---------------------------------------------------------------------------------------------
// cost of the QP at the k-th perturbed time vector (k = 0: xs itself), forward sweep only

__device__ __forceinline__ double perturbed_time(const double* xs, int i, int k, double corr) {
  double T = xs[i];
  if (k > 0) {
    T += (i == k - 1) ? kGradStep : -corr;
    T = fmax(T, kTimeLowerBound);
  }
  return T;
}

// The vertex constraints of a path do not change during the outer loop, and every objective evaluation walks all
// of them on every lane: they are staged once per kernel in LDS (kVtxLds doubles per vertex: the 5 x 4 constrained
// values, 0 where free, then the free mask).  Read from global memory inside the sweep, each segment step exposed
// one L2 round trip on the only wavefront of its SIMD.
constexpr int kVtxLds = 22;

__device__ __forceinline__ void stage_vertices(const uint8_t* __restrict__ mask, const double* __restrict__ vals, int v0,
                                               int S, double* vtx, int g, int G) {
  for (int v = g; v <= S; v += G) {
    double f[kHalf][kD];
    bool pf;
    const unsigned fb = load_vertex<kD>(mask, vals, v0 + v, 0, f, pf);
    double* r = vtx + (size_t)v * kVtxLds;
#pragma unroll
    for (int k = 0; k < kHalf; ++k)
#pragma unroll
      for (int dd = 0; dd < kD; ++dd) r[k * kD + dd] = f[k][dd];
    r[20] = (double)fb;
    r[21] = pf ? 1.0 : 0.0;
  }
}

template <int ND>
__device__ __forceinline__ unsigned staged_vertex(const double* vtx, int v, int dim0, double (&f)[kHalf][ND]) {
  const double* r = vtx + (size_t)v * kVtxLds;
#pragma unroll
  for (int k = 0; k < kHalf; ++k)
#pragma unroll
    for (int dd = 0; dd < ND; ++dd) f[k][dd] = r[k * kD + dim0 + dd];
  return (unsigned)r[20];
}

// The sweep delivers the optimal cost as 0.5 (qf - red), a difference.  On an ordinary path it cancels 1-4 digits.  A
// trial point that puts a segment on the 0.01 s lower bound next to 10 s neighbours makes qf ~ T^-7 |dp|^2 ~ 1e18 while the
// true cost is ~1e4: all 16 digits cancel and the difference is noise of either sign -- and a negative "cost" passes the
// Armijo test, after which the optimiser runs away (seen on 1 of 65536 random paths: times of 1e17 s).  The reference
// evaluates 0.5 c^T Q c from the coefficients there (computeCost, linear_impl.h:128-141), a large positive number, and
// its line search backtracks.  Measured, the difference carries an absolute error of about 1e-16 qf (2e-7 relative at
// J = 0.5e-9 qf; up to ~100 times that through a long elimination chain), so a cost below 1e-12 qf -- fewer than two to
// four digits left -- is reported as "very large": the same decision (reject, backtrack) without claiming a value.
// The perturbed evaluations of the forward-difference gradient get one more decade: a sentinel there turns one
// gradient component into 1e301 and wrecks the next direction, which is worse than a component with 1e-3 noise
// (a first version used 0.5e-9 for both and stopped early on a path with a 0.11 s segment).
constexpr double kUnreliableCost = 1.0e300;
constexpr double kGuardBase = 1.0e-12, kGuardPerturbed = 1.0e-13;

// J and qf are the sums over all four dimensions (a dimension whose waypoints do not move has qf = J = rounding noise
// of either sign, so the test cannot be made per dimension); the absolute floor keeps a path that does not move at all
// out of it.
__device__ __forceinline__ double guarded_cost(double J, double qf, bool base) {
  return (J >= (base ? kGuardBase : kGuardPerturbed) * qf || qf < 1e-9) ? J : kUnreliableCost;
}

// ---- specialised segment steps of the forward sweep ---------------------------------------------------------
// Almost every segment of almost every path is one of three shapes: both end vertices constrain their position
// only (interior segment), the start vertex is fully constrained with zero derivatives (first segment, or the one
// after a full stop), or the end vertex is (last segment, or the one before a full stop).  For these the general
// masked step (Elim::absorb_segment: ~650 instructions per dimension-lane) collapses:
//   * the right-hand-side terms u = H [f_s; f_e] involve only the two position columns, and
//     H[a][0] f_s + H[a][5] f_e = T^(a%5 + 1 - 2d) * (HBAR[a][0] f_s + HBAR[a][5] f_e): the bracket does not depend
//     on the segment time, so it is computed once per kernel and staged in LDS (kSegLds doubles per segment: per
//     dimension the eight brackets of the derivative rows and the f^T HBAR f term of qf);
//   * no masks, no materialised 10 x 10 block: the 4 x 4 blocks are HBAR constants times T^(r + c + 2 - 2d), fused
//     into the accumulations that consume them;
//   * a fully constrained vertex is not factorised at all.
// ~230 instructions per interior step for one dimension per lane, ~360 for four.  A first segment that starts from a
// moving state has its own variant of the start step (kSegStartState below); any other mask / value pattern
// (partially constrained stop vertices, a moving start straight into a stop) takes the general step.
constexpr int kSegLds = 38;
enum { kSegGeneral = 0, kSegStart = 1, kSegInterior = 2, kSegEnd = 3, kSegStartState = 4, kSegMasked = 5, kSegMaskedStartState = 6 };
// kSegMasked: every constrained derivative value is zero (so the position brackets are the whole right-hand side) but
// the free masks of the two vertices are not one of the three patterns above: the end vertices of a rest-to-rest path
// under the minimum-acceleration or minimum-jerk objective (d = 2, the reference's shipping default, leaves jerk and
// snap free there), stop_at vertices (snap free).  One masked variant of the interior step covers them, in either sweep
// direction; record slot 37 holds the two masks (start | end << 4) for every non-general kind.
// kSegStartState: the first segment of a path that starts from a moving state (what the service layer sends: the
// current velocity / acceleration / jerk as a fully constrained vertex with non-zero values).  As with kSegStart nothing
// is eliminated at its start vertex, but the right-hand side of the far vertex is a polynomial in T -- row r gets
// sum_c HBAR[6+r][c] f_c T^(r+1+c+1-2d) -- and so is f^T H f.  Their time-independent coefficients (per dimension 4 x 4
// for the rows, 8 for the powers 1..8 of f^T H f; power 0 and the position bracket are the ordinary record) sit in
// kStartExtra doubles in front of the segment records.
// kSegMaskedStartState: the same with free slots beside the constrained values (a moving start under an objective order
// below snap -- the nodelet's default config: velocity, acceleration and jerk come from the vehicle's state, snap is an
// unknown): the masked step, plus the terms above for vertex 1 and f^T H f, plus the start vertex's own right-hand side
// -- row r gets sum_c HBAR[1+r][c] f_c T^(r+1+c+1-2d) -- whose coefficients are 16 more doubles per dimension.  Only
// optimize_wave_kernel has a step for it; every other evaluation takes the general step.
constexpr int kStartExtraDim = 16 + 8 + 16;
constexpr int kStartExtra = kD * kStartExtraDim;

__device__ __forceinline__ void stage_segments(const double* vtx, int S, int d, double* seg, int g, int G, bool extras) {
  const double (*hb)[kN] = c_hbar[d];
  for (int i = g; i < S; i += G) {
    const double* vs = vtx + (size_t)i * kVtxLds;
    const double* ve = vs + kVtxLds;
    const unsigned fs = (unsigned)vs[20], fe = (unsigned)ve[20];
    // "plain": both positions constrained, every constrained derivative value zero (bit-wise tests on values that
    // are loaded unconditionally: a short-circuit chain would put each LDS load behind its own branch)
    double nz = 0.0;
#pragma unroll
    for (int k = 1; k < kHalf; ++k)
#pragma unroll
      for (int q = 0; q < kD; ++q) nz += fabs(vs[k * kD + q]) + fabs(ve[k * kD + q]);
    const bool plain = (nz == 0.0) & (vs[21] != 0.0) & (ve[21] != 0.0);
    int kind = kSegGeneral;
    if (plain) {
      if (fs == 0xFu && fe == 0xFu) kind = kSegInterior;
      else if (fs == 0u && fe == 0xFu) kind = kSegStart;
      else if (fs == 0xFu && fe == 0u) kind = kSegEnd;
      else kind = kSegMasked;
    } else if (extras && i == 0 && vs[21] != 0.0 && ve[21] != 0.0) {
      // (the staged values of unconstrained slots are zero: load_vertex)
      double nze = 0.0;
#pragma unroll
      for (int k = 1; k < kHalf; ++k)
#pragma unroll
        for (int q = 0; q < kD; ++q) nze += fabs(ve[k * kD + q]);
      if (nze == 0.0) {
        kind = (fs == 0u && fe == 0xFu) ? kSegStartState : kSegMaskedStartState;
        double* ex = seg - kStartExtra;
#pragma unroll
        for (int q = 0; q < kD; ++q) {
          double* e = ex + q * kStartExtraDim;
          const double dp = vs[q] - ve[q];
#pragma unroll
          for (int r = 0; r < kNB; ++r)
#pragma unroll
            for (int c = 1; c < kHalf; ++c) {
              e[r * 4 + (c - 1)] = hb[kHalf + kSlot0 + r][c] * vs[c * kD + q];
              e[24 + r * 4 + (c - 1)] = hb[kSlot0 + r][c] * vs[c * kD + q];
            }
          double Q[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int c = 1; c < kHalf; ++c) {
            Q[c] += 2.0 * vs[c * kD + q] * (hb[c][0] * dp);  // derivative x position (HBAR[c][5] = -HBAR[c][0])
#pragma unroll
            for (int c2 = 1; c2 < kHalf; ++c2) Q[c + c2] += vs[c * kD + q] * vs[c2 * kD + q] * hb[c][c2];
          }
#pragma unroll
          for (int m = 1; m < 9; ++m) e[16 + (m - 1)] = Q[m];
        }
      }
    }
    double* r = seg + (size_t)i * kSegLds;
#pragma unroll
    // A constant polynomial costs nothing, so HBAR[a][5] = -HBAR[a][0] (exactly, the constants are correctly rounded):
    // the position terms depend on the difference of the two positions only.  Formed from the difference they carry
    // a rounding error of eps |H dp| instead of eps |H p| -- at 10 m from the origin with 0.3 m between waypoints that
    // is 30 times less noise in the right-hand sides and 1000 times less in f^T H f, the quantities whose difference is
    // the cost.
    for (int q = 0; q < kD; ++q) {
      const double dp = vs[q] - ve[q];
#pragma unroll
      for (int k = 0; k < kNB; ++k) {
        r[q * 9 + k] = hb[kSlot0 + k][0] * dp;
        r[q * 9 + kNB + k] = hb[kHalf + kSlot0 + k][0] * dp;
      }
      r[q * 9 + 8] = hb[0][0] * dp * dp;
    }
    r[36] = (double)kind;
    r[37] = (double)(fs | (fe << 4));
  }
}


// p2[m] = T^(m + 1 - 2d) without the IEEE division sequence
__device__ __forceinline__ void segment_powers(double T, int d, double (&p2)[9]) {
  const double t2 = T * T;
  const double td = (d == 0) ? 1.0 : (d == 1) ? T : (d == 2) ? t2 : (d == 3) ? t2 * T : t2 * t2;
  const double t4 = t2 * t2;
  p2[0] = T * rcp_refined(td * td);
  p2[1] = p2[0] * T;   // depth-3 product tree instead of an 8-long dependent chain
  p2[2] = p2[0] * t2;
  p2[3] = p2[1] * t2;
  p2[4] = p2[0] * t4;
  p2[5] = p2[1] * t4;
  p2[6] = p2[2] * t4;
  p2[7] = p2[3] * t4;
  p2[8] = p2[4] * t4;
}

// The 36 HBAR_d constants of the three 4 x 4 blocks (start-start packed | start-end | end-end packed) are staged in
// LDS as well: addressed through the constant bank with a run-time d they were scalar loads inside the sweep, whose
// latency stalled every step and whose SGPR footprint forced spills.
constexpr int kBlockConsts = 36;

__device__ __forceinline__ void stage_block_constants(int d, double* hc, int tid, int nthreads) {
  const double (*hb)[kN] = c_hbar[d];
  for (int e = tid; e < kBlockConsts; e += nthreads) {
    double v;
    if (e < 10) {
      int r = 0;
      while (tri(r + 1, 0) <= e) ++r;
      v = hb[kSlot0 + r][kSlot0 + (e - tri(r, 0))];
    } else if (e < 26) {
      v = hb[kSlot0 + (e - 10) / kNB][kHalf + kSlot0 + (e - 10) % kNB];
    } else {
      int r = 0;
      while (tri(r + 1, 0) <= e - 26) ++r;
      v = hb[kHalf + kSlot0 + r][kHalf + kSlot0 + (e - 26 - tri(r, 0))];
    }
    hc[e] = v;
  }
}

template <int ND>
struct FastStep {
  // brackets of this (segment, dimension): w[q][0..3] start rows, [4..7] end rows, [8] qf term
  double w[ND][9];

  __device__ __forceinline__ void load(const double* seg, int dim0) {
#pragma unroll
    for (int q = 0; q < ND; ++q)
#pragma unroll
      for (int j = 0; j < 9; ++j) w[q][j] = seg[(dim0 + q) * 9 + j];
  }

  // REV = false: the sweep runs left to right (a segment is entered at its start vertex); REV = true: right to left
  // (entered at its end vertex): "near" / "far" blocks and brackets swap and the coupling block is transposed.
  template <bool REV> static __device__ __forceinline__ constexpr int near_blk(int r, int c) { return (REV ? 26 : 0) + tri(r, c); }
  template <bool REV> static __device__ __forceinline__ constexpr int far_blk(int r, int c) { return (REV ? 0 : 26) + tri(r, c); }
  template <bool REV> static __device__ __forceinline__ constexpr int cpl_blk(int r, int c) { return 10 + (REV ? c * kNB + r : r * kNB + c); }
  template <bool REV> static __device__ __forceinline__ constexpr int near_w(int r) { return REV ? kNB + r : r; }
  template <bool REV> static __device__ __forceinline__ constexpr int far_w(int r) { return REV ? r : kNB + r; }

  // entry vertex fully constrained: nothing to eliminate; the state moves to the far vertex
  template <bool REV, class HC>
  __device__ __forceinline__ void start_t(Elim<ND>& st, const HC& hc, const double (&p2)[9]) const {
#pragma unroll
    for (int r = 0; r < kNB; ++r) {
#pragma unroll
      for (int c = 0; c <= r; ++c) st.Sm[tri(r, c)] = hc[far_blk<REV>(r, c)] * p2[r + c + 2];
#pragma unroll
      for (int q = 0; q < ND; ++q) st.y[r][q] = -(w[q][far_w<REV>(r)] * p2[r + 1]);
    }
#pragma unroll
    for (int q = 0; q < ND; ++q) st.qf = fma(p2[0], w[q][8], st.qf);
  }
  template <class HC>
  __device__ __forceinline__ void start(Elim<ND>& st, const HC& hc, const double (&p2)[9]) const {
    start_t<false>(st, hc, p2);
  }
  // kSegStartState (see there): the ordinary start step plus the terms of the start vertex's derivative values
  template <class HC>
  __device__ __forceinline__ void start_state(Elim<ND>& st, const HC& hc, const double (&p2)[9], const double* ex,
                                              int dim0) const {
    start_t<false>(st, hc, p2);
#pragma unroll
    for (int q = 0; q < ND; ++q) {
      const double* e = ex + (dim0 + q) * kStartExtraDim;
#pragma unroll
      for (int r = 0; r < kNB; ++r) {
        double u = 0.0;
#pragma unroll
        for (int c = 1; c < kHalf; ++c) u = fma(e[r * 4 + (c - 1)], p2[r + 1 + c], u);
        st.y[r][q] -= u;
      }
#pragma unroll
      for (int m = 1; m < 9; ++m) st.qf = fma(e[16 + (m - 1)], p2[m], st.qf);
    }
  }

  // Cholesky of st.Sm (all four slots free), z = L^-1 y, red += |z|^2
  __device__ __forceinline__ void factor(Elim<ND>& st, double (&L)[10], double (&Linv)[kNB], double (&z)[kNB][ND]) const {
#pragma unroll
    for (int c = 0; c < kNB; ++c) {
      double dsum = st.Sm[tri(c, c)];
#pragma unroll
      for (int m = 0; m < c; ++m) dsum = fma(-L[tri(c, m)], L[tri(c, m)], dsum);
      const double inv = rsqrt_refined(dsum);
      L[tri(c, c)] = dsum * inv;
      Linv[c] = inv;
#pragma unroll
      for (int r = c + 1; r < kNB; ++r) {
        double s = st.Sm[tri(r, c)];
#pragma unroll
        for (int m = 0; m < c; ++m) s = fma(-L[tri(r, m)], L[tri(c, m)], s);
        L[tri(r, c)] = s * inv;
      }
    }
#pragma unroll
    for (int q = 0; q < ND; ++q)
#pragma unroll
      for (int r = 0; r < kNB; ++r) {
        double s = st.y[r][q];
#pragma unroll
        for (int m = 0; m < r; ++m) s = fma(-L[tri(r, m)], z[m][q], s);
        z[r][q] = s * Linv[r];
        st.red = fma(z[r][q], z[r][q], st.red);
      }
  }

  // end vertex fully constrained: eliminate the start vertex, nothing is passed on
  template <class HC>
  __device__ __forceinline__ void end(Elim<ND>& st, const HC& hc, const double (&p2)[9]) const {
#pragma unroll
    for (int r = 0; r < kNB; ++r) {
#pragma unroll
      for (int c = 0; c <= r; ++c) st.Sm[tri(r, c)] = fma(hc[tri(r, c)], p2[r + c + 2], st.Sm[tri(r, c)]);
#pragma unroll
      for (int q = 0; q < ND; ++q) st.y[r][q] = fma(-w[q][r], p2[r + 1], st.y[r][q]);
    }
#pragma unroll
    for (int q = 0; q < ND; ++q) st.qf = fma(p2[0], w[q][8], st.qf);
    double L[10], Linv[kNB], z[kNB][ND];
    factor(st, L, Linv, z);
  }

  // both vertices constrain their position only
  template <bool REV, class HC>
  __device__ __forceinline__ void interior_t(Elim<ND>& st, const HC& hc, const double (&p2)[9]) const {
#pragma unroll
    for (int r = 0; r < kNB; ++r) {
#pragma unroll
      for (int c = 0; c <= r; ++c) st.Sm[tri(r, c)] = fma(hc[near_blk<REV>(r, c)], p2[r + c + 2], st.Sm[tri(r, c)]);
#pragma unroll
      for (int q = 0; q < ND; ++q) st.y[r][q] = fma(-w[q][near_w<REV>(r)], p2[r + 1], st.y[r][q]);
    }
#pragma unroll
    for (int q = 0; q < ND; ++q) st.qf = fma(p2[0], w[q][8], st.qf);
    double L[10], Linv[kNB], z[kNB][ND], W[kNB][kNB];
    factor(st, L, Linv, z);
    // W = L^-1 E, E[r][c] = HBAR[1+r][6+c] T^(r+c+2-2d) (transposed when sweeping right to left)
#pragma unroll
    for (int c = 0; c < kNB; ++c)
#pragma unroll
      for (int r = 0; r < kNB; ++r) {
        double s = hc[cpl_blk<REV>(r, c)] * p2[r + c + 2];
#pragma unroll
        for (int m = 0; m < r; ++m) s = fma(-L[tri(r, m)], W[m][c], s);
        W[r][c] = s * Linv[r];
      }
    // next vertex: Sm = H_far - W^T W ; y = -u_far - W^T z
#pragma unroll
    for (int r = 0; r < kNB; ++r) {
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        double s = hc[far_blk<REV>(r, c)] * p2[r + c + 2];
#pragma unroll
        for (int m = 0; m < kNB; ++m) s = fma(-W[m][r], W[m][c], s);
        st.Sm[tri(r, c)] = s;
      }
#pragma unroll
      for (int q = 0; q < ND; ++q) {
        double s = -(w[q][far_w<REV>(r)] * p2[r + 1]);
#pragma unroll
        for (int m = 0; m < kNB; ++m) s = fma(-W[m][r], z[m][q], s);
        st.y[r][q] = s;
      }
    }
  }
  template <class HC>
  __device__ __forceinline__ void interior(Elim<ND>& st, const HC& hc, const double (&p2)[9]) const {
    interior_t<false>(st, hc, p2);
  }

  // constrained slots of the vertex the state stands on: identity row, zero right-hand side (as Elim::factor_vertex)
  static __device__ __forceinline__ void apply_mask(Elim<ND>& st, unsigned free_mask) {
#pragma unroll
    for (int r = 0; r < kNB; ++r) {
      const bool fr = (free_mask >> r) & 1u;
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        const bool fc = (free_mask >> c) & 1u;
        const double v = st.Sm[tri(r, c)];
        st.Sm[tri(r, c)] = (r == c) ? (fr ? v : 1.0) : ((fr && fc) ? v : 0.0);
      }
#pragma unroll
      for (int q = 0; q < ND; ++q) st.y[r][q] = fr ? st.y[r][q] : 0.0;
    }
  }

  // kSegMasked: the interior step with the free masks of the near (mn) and far (mf) vertex.  mn = 0 reduces it to the
  // start step, mf = 0 to the end step, both 0xF to the plain interior step.
  template <bool REV, class HC>
  __device__ __forceinline__ void masked_t(Elim<ND>& st, const HC& hc, const double (&p2)[9], unsigned mn, unsigned mf) const {
#pragma unroll
    for (int r = 0; r < kNB; ++r) {
#pragma unroll
      for (int c = 0; c <= r; ++c) st.Sm[tri(r, c)] = fma(hc[near_blk<REV>(r, c)], p2[r + c + 2], st.Sm[tri(r, c)]);
#pragma unroll
      for (int q = 0; q < ND; ++q) st.y[r][q] = fma(-w[q][near_w<REV>(r)], p2[r + 1], st.y[r][q]);
    }
#pragma unroll
    for (int q = 0; q < ND; ++q) st.qf = fma(p2[0], w[q][8], st.qf);
    apply_mask(st, mn);
    double L[10], Linv[kNB], z[kNB][ND], W[kNB][kNB];
    factor(st, L, Linv, z);
#pragma unroll
    for (int c = 0; c < kNB; ++c)
#pragma unroll
      for (int r = 0; r < kNB; ++r) {
        const bool on = ((mn >> r) & 1u) && ((mf >> c) & 1u);
        double s = on ? hc[cpl_blk<REV>(r, c)] * p2[r + c + 2] : 0.0;
#pragma unroll
        for (int m = 0; m < r; ++m) s = fma(-L[tri(r, m)], W[m][c], s);
        W[r][c] = s * Linv[r];
      }
#pragma unroll
    for (int r = 0; r < kNB; ++r) {
      const bool fr = (mf >> r) & 1u;
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        const bool fc = (mf >> c) & 1u;
        double s = (fr && fc) ? hc[far_blk<REV>(r, c)] * p2[r + c + 2] : 0.0;
#pragma unroll
        for (int m = 0; m < kNB; ++m) s = fma(-W[m][r], W[m][c], s);
        st.Sm[tri(r, c)] = s;
      }
#pragma unroll
      for (int q = 0; q < ND; ++q) {
        double s = fr ? -(w[q][far_w<REV>(r)] * p2[r + 1]) : 0.0;
#pragma unroll
        for (int m = 0; m < kNB; ++m) s = fma(-W[m][r], z[m][q], s);
        st.y[r][q] = s;
      }
    }
  }
};

// ---- direction tables of the lean sweeps ------------------------------------------------------------------------------
// (These tables and stage_ps below were built for the prefix / suffix evaluation of the Mellinger gradient -- an experiment
// that was verified, measured slower than the sweeping kernels and removed from the source in round 3; DESIGN.md section 13
// keeps its description and numbers.  The lean sweeps use the tables and the staging pass.)
constexpr int kPsTable = 45;  // near block [10] | coupling [16] | far block [10] | near bracket consts [4] | far [4] | qf const

// [0]: the table of a left-to-right lane, [1]: of a right-to-left lane (near / far swapped, coupling transposed), so that
// FastStep's REV = false forms compute either direction
__device__ __forceinline__ void stage_ps_tables(int d, double* tab, int tid, int nthreads) {
  const double (*hb)[kN] = c_hbar[d];
  for (int e = tid; e < 2 * kPsTable; e += nthreads) {
    const int rev = e / kPsTable, i = e % kPsTable;
    double v;
    if (i < 10 || (i >= 26 && i < 36)) {
      const bool start_block = (i < 10) != (rev != 0);  // the block of the vertex the sweep enters the segment at
      const int t = i < 10 ? i : i - 26;
      int r = 0;
      while (tri(r + 1, 0) <= t) ++r;
      const int c = t - tri(r, 0);
      v = start_block ? hb[kSlot0 + r][kSlot0 + c] : hb[kHalf + kSlot0 + r][kHalf + kSlot0 + c];
    } else if (i < 26) {
      const int r = (i - 10) / kNB, c = (i - 10) % kNB;
      v = rev ? hb[kSlot0 + c][kHalf + kSlot0 + r] : hb[kSlot0 + r][kHalf + kSlot0 + c];
    } else if (i < 44) {
      const int k = (i - 36) % kNB;
      const bool start_rows = (i < 40) != (rev != 0);
      v = start_rows ? hb[kSlot0 + k][0] : hb[kHalf + kSlot0 + k][0];
    } else {
      v = hb[0][0];
    }
    tab[e] = v;
  }
}

// (Used by the A / B build -DMRS_TG_LEAN_CONST_TABLE=1 only: measured slower than the per-lane register table, see
// mrs_tg_nonlinear.hip.)  The left-to-right table of stage_ps_tables as COMPILE-TIME constants of objective order D (round 5).
// The plain-path lean kernels only ever run D = 4 (orders below snap leave slots free at the end vertices and take the masked kernel), and a
// table that is the same in every lane need not live in 90 VGPRs per lane: a constant operand of a VOP3 instruction is an
// SGPR pair the scalar unit sets up beside the vector work.  ONE table serves both sweep directions: reversing a segment's
// time maps end-point derivative r to (-1)^r times the other end's, so every entry of the right-to-left table is the
// left-to-right entry times sigma_r sigma_c, sigma_r = (-1)^r over the free slots r = 0..3 (velocity .. snap) -- exactly, the
// constants being correctly rounded images of rationals with that symmetry (tests/test_oracle_golden.py checks the bits).
// A right-to-left lane therefore runs the left-to-right arithmetic on x' = diag(sigma) x: its blocks are D A D, its
// right-hand sides D y, its Cholesky factor D L D, z' = D z -- the same magnitudes, sign-flipped where r + c is odd, so the
// cost terms (qf, |z|^2) are the SAME BITS as with the mirrored table; only where a state crosses between lanes of opposite
// direction (the hand-over of evaluate_lean_shared) the signs are applied: lean_flip_state.
constexpr double k_hbar_cx[kHalf][kN][kN] = MRS_TG_HBAR_INIT;

template <int D>
struct PsTab {
  static __host__ __device__ constexpr int tri_row(int t) { return t < 1 ? 0 : t < 3 ? 1 : t < 6 ? 2 : 3; }
  static __host__ __device__ constexpr double at(int i) {
    if (i < 10) {
      const int r = tri_row(i), c = i - r * (r + 1) / 2;
      return k_hbar_cx[D][kSlot0 + r][kSlot0 + c];
    }
    if (i < 26) return k_hbar_cx[D][kSlot0 + (i - 10) / kNB][kHalf + kSlot0 + (i - 10) % kNB];
    if (i < 36) {
      const int t = i - 26, r = tri_row(t), c = t - r * (r + 1) / 2;
      return k_hbar_cx[D][kHalf + kSlot0 + r][kHalf + kSlot0 + c];
    }
    if (i < 40) return k_hbar_cx[D][kSlot0 + (i - 36)][0];
    if (i < 44) return k_hbar_cx[D][kHalf + kSlot0 + (i - 40)][0];
    return k_hbar_cx[D][0][0];
  }
};

// x' = diag(sigma) x for a half-sweep state (Sm packed lower, y[row][dimension]): entries with r + c odd and rows r = 1, 3
// change sign when flip is -1.0 (a multiplication by +-1 is exact)
__device__ __forceinline__ void lean_flip_state(double (&Sm)[10], double (&y)[kNB][4], double flip) {
  Sm[tri(1, 0)] *= flip;
  Sm[tri(2, 1)] *= flip;
  Sm[tri(3, 0)] *= flip;
  Sm[tri(3, 2)] *= flip;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    y[1][q] *= flip;
    y[3][q] *= flip;
  }
}

// p2[m] = T^(m + 1 - 2 D) for a compile-time objective order (segment_powers with its order folded)
template <int D>
__device__ __forceinline__ void segment_powers_c(double T, double (&p2)[9]) {
  const double t2 = T * T;
  const double td = (D == 0) ? 1.0 : (D == 1) ? T : (D == 2) ? t2 : (D == 3) ? t2 * T : t2 * t2;
  const double t4 = t2 * t2;
  p2[0] = T * rcp_refined(td * td);
  p2[1] = p2[0] * T;
  p2[2] = p2[0] * t2;
  p2[3] = p2[1] * t2;
  p2[4] = p2[0] * t4;
  p2[5] = p2[1] * t4;
  p2[6] = p2[2] * t4;
  p2[7] = p2[3] * t4;
  p2[8] = p2[4] * t4;
}

__device__ __forceinline__ void ps_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// MASKED4: the four-dimensions-per-lane sweep with the masked step compiled in (launches whose objective order d < 4
// makes every rest-to-rest path end on masked vertices); see the comment at the general branch below
template <int ND, bool MASKED4 = false>
__device__ __forceinline__ double forward_cost(const double* vtx, const double* seg, const double* hc, int S, int d,
                                               const double* xs, int k, int dim0, double& qf_out) {
  Elim<ND> st;
  st.init();
  // one lane = four dimensions (big batches, throughput regime): the 36 block constants stay in registers for the whole
  // sweep.  One lane = one dimension (small batches): they are read from LDS in every step, which keeps the kernel
  // under 256 VGPRs so that the two wavefronts of a path share a SIMD.
  constexpr int kRegConsts = (ND == 4) ? kBlockConsts : 1;
  double hcr[kRegConsts];
  if (ND == 4) {
#pragma unroll
    for (int e = 0; e < kRegConsts; ++e) hcr[e] = hc[e];
  }
  const double corr = kGradStep / ((double)S - 1.0);
  int last_kind = kSegGeneral;
  for (int i = 0; i < S; ++i) {
    const double T = perturbed_time(xs, i, k, corr);
    const double* sr = seg + (size_t)i * kSegLds;
    const int kind = (int)sr[36];
    last_kind = kind;
    // The four-dimensions-per-lane sweep (large batches) keeps the general step for moving starts and masked
    // vertices: their specialised steps would sit in the same loop as the plain ones and cost those ~10 % (register
    // allocation of a 440-VGPR kernel), which the BASELINE batches -- all plain -- would pay for nothing.
    if (kind == kSegGeneral || kind == kSegMaskedStartState ||
        (ND == 4 && (kind == kSegStartState || (kind == kSegMasked && !MASKED4)))) {
      double fs[kHalf][ND], fe[kHalf][ND];
      double L[10], z[kNB][ND], W[kNB][kNB];
      const unsigned free_s = staged_vertex<ND>(vtx, i, dim0, fs);
      const unsigned free_e = staged_vertex<ND>(vtx, i + 1, dim0, fe);
      // positions relative to the segment's start (see stage_segments): same cost, less cancellation noise
#pragma unroll
      for (int q = 0; q < ND; ++q) {
        fe[0][q] -= fs[0][q];
        fs[0][q] = 0.0;
      }
      double Hs[kSym10];
      hessian_from_time(T, d, Hs);
      st.absorb_segment(Hs, fs, fe, free_s, free_e, L, z, W);
    } else {
      double p2[9];
      segment_powers(T, d, p2);
      FastStep<ND> fast;
      fast.load(sr, dim0);
      if (ND == 4) {
        if (kind == kSegInterior) fast.interior(st, hcr, p2);
        else if (kind == kSegStart) fast.start(st, hcr, p2);
        else if (MASKED4 && kind == kSegMasked) fast.template masked_t<false>(st, hcr, p2, (unsigned)sr[37] & 0xFu, (unsigned)sr[37] >> 4);
        else fast.end(st, hcr, p2);
      } else {
        if (kind == kSegInterior) fast.interior(st, hc, p2);
        else if (kind == kSegStart) fast.start(st, hc, p2);
        else if (kind == kSegEnd) fast.end(st, hc, p2);
        else if (kind == kSegStartState) fast.start_state(st, hc, p2, seg - kStartExtra, dim0);
        else fast.template masked_t<false>(st, hc, p2, (unsigned)sr[37] & 0xFu, (unsigned)sr[37] >> 4);
      }
    }
  }
  if (last_kind != kSegEnd) {  // after an end-type segment the state stands on a fully constrained vertex: nothing left
    double fl[kHalf][ND];
    double L[10], z[kNB][ND];
    const unsigned free_l = staged_vertex<ND>(vtx, S, dim0, fl);
    st.factor_vertex(free_l, L, z);
  }
  qf_out = st.qf;
  return 0.5 * (st.qf - st.red);
}

// objective evaluation at `pt`: cost returned to every lane of the group, gradient to `grad` (LDS).
// (objectiveFunctionTimeMellingerOuterLoop + getCostAndGradientMellinger)
// DS lanes share one time vector (DS = 1: one lane, four dimensions; DS = 4: four lanes, one dimension each).
template <int DS, bool MASKED4 = false>
__device__ __forceinline__ double evaluate_objective(const double* vtx, const double* seg, const double* hc, int S, int d,
                                                     const double* pt, double* grad, int g, int G, bool active,
                                                     int* tripped = nullptr) {
  constexpr int ND = kD / DS;
  const int kl_shift = __builtin_ctz((unsigned)G) - (DS == 4 ? 2 : 0);  // G is a power of two >= DS: no integer division
  const int kl = 1 << kl_shift;  // time vectors handled per round
  const int kk = g / DS, dim0 = (g % DS) * ND;
  double J0 = 0.0;
  const int rounds = (S + kl) >> kl_shift;
  for (int r = 0; r < rounds; ++r) {
    const int k = kk + r * kl;
    double Jk = 0.0, qfk = 0.0;
    if (active && k <= S && (k == 0 || S > 1)) Jk = forward_cost<ND, MASKED4>(vtx, seg, hc, S, d, pt, k, dim0, qfk);
    if (DS == 4) {  // the four dimensions of one time vector sit in one quad
      Jk += dpp_move<0xB1>(Jk);
      Jk += dpp_move<0x4E>(Jk);
      qfk += dpp_move<0xB1>(qfk);
      qfk += dpp_move<0x4E>(qfk);
    }
    Jk = guarded_cost(Jk, qfk, k == 0);
    if (tripped && active && k <= S && Jk == kUnreliableCost) *tripped = 1;  // (every writer writes the same value)
    if (r == 0) J0 = __shfl(Jk, (threadIdx.x & ~(G - 1)), 64);  // lane 0 of the group holds k = 0
    if (active && dim0 == 0 && k >= 1 && k <= S) grad[k - 1] = (S > 1) ? (Jk - J0) / kGradStep : 0.0;
  }
  return J0;
}

}  // namespace mrs_tg
