// mrs_tg_nonlinear.hip -- segment-time outer loop (mode 2, Mellinger), feasibility scaling and the
// final re-solve; plus the two building-block kernels exposed for parity tests.
//
// Reference behaviour being reproduced (paths relative to /root/reference/):
//   include/eth_trajectory_generation/impl/polynomial_optimization_nonlinear_impl.h
//     :160-234  optimizeTimeMellingerOuterLoop      :257-333  getCostAndGradientMellinger
//     :336-408  scaleSegmentTimesWithViolation      :617-649  objectiveFunctionTimeMellingerOuterLoop
//   src/eth_trajectory_generation/trajectory.cpp:598-692  scaleSegmentTimesToMeetConstraints
//   src/eth_trajectory_generation/segment.cpp:113-212     magnitude extremum candidates
//
// Pipeline of a call (host-orchestrated, one stream; launch_nonlinear at the end of this file):
//   optimize_lean_kernel (plain paths of large batches) and / or optimize_split_kernel / optimize_compact_kernel (one
//                        dimension / four dimensions per lane): the outer loop, every bin of the plan in one launch;
//                        every tick = one objective evaluation = S+1 cost sweeps.  On request behind them:
//                        optimize_careful_kernel (MRS_TG_FLAG_CAREFUL_COST), optimize_general_kernel (paths with a
//                        position-free vertex, MRS_TG_FLAG_GENERAL_PATTERNS)
//   solve                coefficients at the last evaluated times              (mrs_tg_rows / mrs_tg_kernels)
//   segment_maxima9      max |p^(k)| per segment (mrs_tg_maxima.hpp), one (k, group) per blockIdx.y
//   solve with its tail  T <- T * max(1, v, sqrt a, cbrt j) in the staging pass, final coefficients, cost, merged status,
//                        samples (mrs_tg_rows.hip); batches that kernel does not take: apply_scaling + runaway + solve
// The gradient-free modes 0 / 1 / 3 / 4 live in mrs_tg_dfo.hip.
//
// Mapping of the outer loop onto the wavefront: one GROUP of G lanes per path, one wavefront per workgroup.
// An objective evaluation needs the cost at S+1 time vectors (unperturbed + one per segment,
// nonlinear_impl.h:282-323).  Compact mapping (DS = 1, big batches): lane k runs the whole forward sweep of
// vector k for all four dimensions, G = min(64, pow2ceil(S+1)).  Split mapping (DS = 4, small batches):
// four lanes per vector, one dimension each (the 4x4 factorisation is repeated, the dependency chain is
// ~3x shorter), G = min(64, pow2ceil(4(S+1))).  Only the cost is needed, so the sweeps are forward-only
// (cost = (f^T R_ff f - |L^-1 b|^2)/2) and touch no memory besides the vertex constraints.  The optimiser's
// vectors live in LDS; its scalar control flow is replicated in every lane of the group (all lanes see
// identical reduced values).  NLopt's LD_LBFGS is not reproducible (un-vendored, Luksan PLIS); the optimiser
// is the project's own projected L-BFGS, specified in DESIGN.md and restated on the CPU in
// oracle/mto_nonlinear.c.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "mrs_tg_device.hpp"
#include "mrs_tg_estimate.hpp"
#include "mrs_tg_pool.h"
#include "mrs_tg_solve.hpp"
#include "mrs_tg_nonlinear.h"
#include "mrs_tg_general.hpp"
#include "mrs_tg_nl_common.hpp"
#include "mrs_tg_maxima.hpp"
#include "mrs_tg_sweep.hpp"
#include "mrs_tg_wave.h"

namespace mrs_tg {

#ifndef MRS_TG_MAXIMA_WAVES
#define MRS_TG_MAXIMA_WAVES 4
#endif
#ifndef MRS_TG_LEAN_WAVES
#define MRS_TG_LEAN_WAVES 2
#endif
// The careful re-run (MRS_TG_FLAG_CAREFUL_COST: a second outer-loop kernel with the primal cost for the paths whose by-product
// cost failed the guard) moved 65536 x 10 from 99.9435 % to 99.9481 % agreement with the oracle at 2-3x the outer loop's time
// (DESIGN.md section 5): a compile-time option, ON in the shipped library since ABI 4 (build.py: MRS_TG_WITH_CAREFUL=0 in the
// environment builds without it; mrs_tg_capabilities() reports which).
#ifndef MRS_TG_WITH_CAREFUL
#define MRS_TG_WITH_CAREFUL 1
#endif

// ---- two-sided evaluation for small batches ------------------------------------------------------------------
// With one path per wavefront (DS = 4, 4 (S+1) <= 64 lanes) the machine holds one wavefront per SIMD and a tick is
// a chain of S dependent segment steps.  A second wavefront per path runs the same elimination from the other end
// (twisted factorisation: the two half sweeps meet at the middle vertex, whose block receives a Schur update from each
// side); the chain per tick is S/2 steps + one join, and the two wavefronts of a path hide each other's latencies.
// Only for "plain" paths (start, interior..., end segment kinds); anything else keeps the one-sided sweep.
constexpr int kPairState = 16;  // Sm[10], y[4], qf, red handed from the backward to the forward wavefront, per lane

// special: bit i set = segment i takes the masked step, bit 31 = the first segment is a moving start (wave-uniform,
// worked out once per kernel).  The plain start | interior ... | end path is its own instantiation: even a never-taken
// branch to the masked step inside its loop cost it 8 % (register allocation at the 256-VGPR cap).
template <bool REV, int SPECIAL>  // 0: plain path, 1: moving start only, 2: masked segments (and possibly a moving start)
__device__ __forceinline__ void half_sweep(const double* seg, const double* hc, int S, int m, int d, const double* xs, int k,
                                           int dim0, unsigned special, Elim<1>& st) {
  st.init();
  // the block constants are read from LDS in every step here: the two wavefronts of a path must fit one SIMD
  // together (<= 256 VGPRs each), which the register-resident copy of the one-sided sweep would not allow
  const double corr = kGradStep / ((double)S - 1.0);
  const int n = REV ? S - m : m;
  for (int s = 0; s < n; ++s) {
    const int i = REV ? S - 1 - s : s;
    double p2[9];
    segment_powers(perturbed_time(xs, i, k, corr), d, p2);
    FastStep<1> fast;
    fast.load(seg + (size_t)i * kSegLds, dim0);
    if (SPECIAL == 2 && ((special >> i) & 1u)) {
      const unsigned masks = (unsigned)seg[(size_t)i * kSegLds + 37];
      const unsigned ms = masks & 0xFu, me = masks >> 4;
      fast.template masked_t<REV>(st, hc, p2, REV ? me : ms, REV ? ms : me);
    } else if (s == 0) {
      if (SPECIAL != 0 && !REV && (special >> 31)) fast.start_state(st, hc, p2, seg - kStartExtra, dim0);
      else fast.template start_t<REV>(st, hc, p2);
    } else {
      fast.template interior_t<REV>(st, hc, p2);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// group helpers (G lanes, G a power of two <= 64, groups aligned to G)

// The optimiser's bookkeeping is a chain of ~20 small reductions per tick.  Butterflies through ds_bpermute
// (__shfl_xor) cost ~700 shader cycles per 64-lane reduction; DPP moves inside a row of 16 lanes and v_readlane
// across rows cost ~150.  Every lane of a group receives the bit-identical sum (each step adds the same two partial
// sums on both sides), which the control flow relies on.
// G = 0: the group is the wavefront but only lanes 0..15 hold non-zero terms (a vector of at most 16 elements in a
// 64-lane group): the sum of row 0, read from lane 0 -- the same number as the four-row sum (the other rows add + 0.0), three
// cross-row reads and adds cheaper.
__device__ __forceinline__ double group_sum(double v, int G) {
  if (G == 0) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    return row_value(v, 0);
  }
  v += dpp_move<0xB1>(v);                 // quad_perm [1,0,3,2]: lane ^ 1
  if (G >= 4) v += dpp_move<0x4E>(v);     // quad_perm [2,3,0,1]: lane ^ 2
  if (G >= 8) v += dpp_move<0x141>(v);    // row_half_mirror: the other quad of each 8
  if (G >= 16) v += dpp_move<0x140>(v);   // row_mirror: the other half of each row of 16
  if (G >= 32) {
    // every lane of a row now holds its row's sum.  Rows of a group that is masked off at this point deliver stale
    // values, which only that group's (inactive) lanes would consume.
    const double a = row_value(v, 0) + row_value(v, 16), b = row_value(v, 32) + row_value(v, 48);
    v = (G == 64) ? a + b : ((threadIdx.x & 32) ? b : a);
  }
  return v;
}

// N sums at once: the levels (and their branches on G) are shared, the N dependent chains interleave.  Separate
// group_sum calls cannot overlap, each carries its own control flow.
template <int N>
__device__ __forceinline__ void group_sum_n(double (&v)[N], int G) {
  if (G == 0) {  // row 0 only, see group_sum
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] += dpp_move<0xB1>(v[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] += dpp_move<0x4E>(v[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] += dpp_move<0x141>(v[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] += dpp_move<0x140>(v[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] = row_value(v[n], 0);
    return;
  }
#pragma unroll
  for (int n = 0; n < N; ++n) v[n] += dpp_move<0xB1>(v[n]);
  if (G >= 4) {
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] += dpp_move<0x4E>(v[n]);
  }
  if (G >= 8) {
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] += dpp_move<0x141>(v[n]);
  }
  if (G >= 16) {
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] += dpp_move<0x140>(v[n]);
  }
  if (G >= 32) {
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const double a = row_value(v[n], 0) + row_value(v[n], 16), b = row_value(v[n], 32) + row_value(v[n], 48);
      v[n] = (G == 64) ? a + b : ((threadIdx.x & 32) ? b : a);
    }
  }
}

__device__ __forceinline__ int group_and(int v, int G) {
  const unsigned long long m = __ballot(v != 0);
  const unsigned long long gm = (G == 64) ? ~0ull : (((1ull << G) - 1ull) << (threadIdx.x & ~(G - 1)));
  return (m & gm) == gm;
}

__device__ __forceinline__ double group_dot(const double* a, const double* b, int S, int g, int G) {
  double s = 0.0;
  for (int i = g; i < S; i += G) s += a[i] * b[i];
  return group_sum(s, G);
}

// The optimiser's scalar state (f, alpha | neval, npairs, head, ret) is the same in every lane of a group; it is parked
// in LDS while the objective is evaluated, so that it does not occupy registers there (the kernel sits on its 256-VGPR
// cap and the compiler spilled exactly these to scratch memory, a global-memory round trip per reload).
constexpr int kTickState = 5;  // doubles: f, alpha, then six ints (neval, npairs, head, ret | guard tripped, spare)
// per-group LDS block (doubles): x, g, xn, gn, dir [5*Sb], s[M][Sb], y[M][Sb], rho[M + 1], tick state, staged vertices
// [(Sb+1)*kVtxLds], moving-start extras [kStartExtra], staged segment records [Sb*kSegLds]
// extras: the moving-start area (only the one-dimension-per-lane kernels use it; in the four-dimensions-per-lane mapping
// of large batches many groups share a workgroup and 96 doubles each cost a workgroup per CU)
__host__ __device__ constexpr int group_lds_doubles(int Sb, bool extras) {
  return (5 + 2 * kLbfgsM) * Sb + (kLbfgsM + 1) + kTickState + (Sb + 1) * kVtxLds + (extras ? kStartExtra : 0) + Sb * kSegLds;
}
// cost_gradient_kernel: x, g [2*Sb], staged vertices
__host__ __device__ constexpr int gradient_lds_doubles(int Sb, bool extras) {
  return 2 * Sb + (Sb + 1) * kVtxLds + (extras ? kStartExtra : 0) + Sb * kSegLds;
}

// Two-sided objective evaluation (forward wavefront's side): forward half sweep, barrier, join with the backward
// half that the partner wavefront left in LDS, then the same reductions as evaluate_objective<4>.
__device__ __forceinline__ double evaluate_pair(const double* seg, const double* hc, const double* pair_state, int S, int d,
                                                const double* pt, double* grad, int g, bool active, unsigned special,
                                                int* tripped) {
  const int k = g >> 2, dim0 = g & 3, m = S / 2;
  const bool work = active && k <= S;
  Elim<1> st;
  st.init();
  if (work) {
    if (special & 0x7FFFFFFFu) half_sweep<false, 2>(seg, hc, S, m, d, pt, k, dim0, special, st);
    else if (special) half_sweep<false, 1>(seg, hc, S, m, d, pt, k, dim0, special, st);
    else half_sweep<false, 0>(seg, hc, S, m, d, pt, k, dim0, 0u, st);
  }
  __syncthreads();  // the partner's half sweeps are in LDS
  double Jk = 0.0, qfk = 0.0;
  if (work) {
    // element-major (element e of lane g at [e * 64 + g]): lane-major records of 16 doubles put every lane of a
    // half-wavefront on the same two banks -- a 16-way conflict on each of the 16 stores and 16 loads of a hand-over
    const double* ps = pair_state + g;
#pragma unroll
    for (int e = 0; e < 10; ++e) st.Sm[e] += ps[e * 64];
#pragma unroll
    for (int r = 0; r < kNB; ++r) st.y[r][0] += ps[(10 + r) * 64];
    st.qf += ps[14 * 64];
    st.red += ps[15 * 64];
    FastStep<1> fs;
    double L[10], Linv[kNB], z[kNB][1];
    // free mask of the middle vertex = end mask of the segment in front of it
    if (special & 0x7FFFFFFFu) FastStep<1>::apply_mask(st, (unsigned)seg[(size_t)(m - 1) * kSegLds + 37] >> 4);
    fs.factor(st, L, Linv, z);
    Jk = 0.5 * (st.qf - st.red);
    qfk = st.qf;
  }
  Jk += dpp_move<0xB1>(Jk);
  Jk += dpp_move<0x4E>(Jk);
  qfk += dpp_move<0xB1>(qfk);
  qfk += dpp_move<0x4E>(qfk);
  Jk = guarded_cost(Jk, qfk, k == 0);
  if (work && Jk == kUnreliableCost) *tripped = 1;
  const double J0 = row_value(Jk, 0);
  if (work && dim0 == 0 && k >= 1) grad[k - 1] = (Jk - J0) / kGradStep;
  return J0;
}

// ---- careful evaluation: the cost as the reference computes it -----------------------------------------------------
// 0.5 c^T Q c from the coefficients (computeCost, linear_impl.h:128-141): a sum of non-negative terms, accurate where the
// by-product 0.5 (qf - red) has cancelled.  One lane = one (time vector, dimension): the general elimination of
// mrs_tg_solve.hpp with the factors of every vertex parked in global memory, back substitution, coefficients segment by
// segment (not stored), their quadratic form.  ~6 times the work of a forward-only sweep and none of its specialised
// steps: used by optimize_careful_kernel only, on the few paths whose fast evaluation failed the guard.
//   ws: element e of vertex v of lane `wlane` at ws[(v * ws_per_vertex<1>() + e) * wstride + wlane]
__device__ __forceinline__ double primal_cost_lane(const uint8_t* __restrict__ mask, const double* __restrict__ vals, int v0,
                                                   int S, int d, const double* xs, int k, int dim0, double* ws,
                                                   size_t wstride, unsigned wlane) {
  constexpr int WSV = ws_per_vertex<1>();
  const double corr = kGradStep / ((double)S - 1.0);
  Elim<1> st;
  st.init();
  VertexData<1> vs, ve;
  SavedFactors<1> sf;
  fetch_vertex<1>(mask, vals, v0, dim0, vs);
  for (int i = 0; i < S; ++i) {
    fetch_vertex<1>(mask, vals, v0 + i + 1, dim0, ve);
    double Hs[kSym10];
    hessian_from_time(perturbed_time(xs, i, k, corr), d, Hs);
    st.absorb_segment(Hs, vs.f, ve.f, vs.free_bits, ve.free_bits, sf.L, sf.z, sf.W);
    store_factors<1>(ws + (size_t)i * WSV * wstride, wstride, wlane, sf);
    vs = ve;
  }
  st.factor_vertex(vs.free_bits, sf.L, sf.z);
  double xn[kNB][1], x[kNB][1], dn[kHalf][1], dc[kHalf][1];
  back_substitute<1>(sf.L, sf.z, sf.W, xn, true, x);
#pragma unroll
  for (int sl = 0; sl < kHalf; ++sl) dn[sl][0] = vs.f[sl][0] + (sl >= kSlot0 ? x[sl - kSlot0][0] : 0.0);
#pragma unroll
  for (int r = 0; r < kNB; ++r) xn[r][0] = x[r][0];
  double total = 0.0;
  for (int i = S - 1; i >= 0; --i) {
    load_factors<1>(ws + (size_t)i * WSV * wstride, wstride, wlane, sf);
    VertexData<1> vc;
    fetch_vertex<1>(mask, vals, v0 + i, dim0, vc);
    back_substitute<1>(sf.L, sf.z, sf.W, xn, false, x);
#pragma unroll
    for (int sl = 0; sl < kHalf; ++sl) dc[sl][0] = vc.f[sl][0] + (sl >= kSlot0 ? x[sl - kSlot0][0] : 0.0);
    // unit-time coefficients cb_k = c_k T^k (coefficients_from_time without its final T^-k) and their quadratic form
    const double T = perturbed_time(xs, i, k, corr);
    double w[kHalf];
    w[0] = 1.0;
#pragma unroll
    for (int kk = 1; kk < kHalf; ++kk) w[kk] = w[kk - 1] * T;
    double db[kN], cb[kN];
#pragma unroll
    for (int sl = 0; sl < kHalf; ++sl) {
      db[sl] = dc[sl][0] * w[sl];
      db[kHalf + sl] = dn[sl][0] * w[sl];
    }
#pragma unroll
    for (int kk = 0; kk < kN; ++kk) {
      double acc = 0.0;
      if (kk < kHalf) {
        acc = c_abar_inv[kk][kk] * db[kk];
      } else {
#pragma unroll
        for (int j = 0; j < kN; ++j) acc += c_abar_inv[kk][j] * db[j];
      }
      cb[kk] = acc;
    }
    double p2[9];
    segment_powers(T, d, p2);  // p2[0] = T^(1 - 2d)
    total = fma(cost_quadratic_form_d(d, cb), p2[0], total);
#pragma unroll
    for (int sl = 0; sl < kHalf; ++sl) dn[sl][0] = dc[sl][0];
#pragma unroll
    for (int r = 0; r < kNB; ++r) xn[r][0] = x[r][0];
  }
  return total;
}

// evaluate_objective<4> with the primal cost (one path per wavefront, G = 64: 16 time vectors per round, four lanes each)
__device__ __forceinline__ double evaluate_careful(const uint8_t* __restrict__ mask, const double* __restrict__ vals, int v0,
                                                   int S, int d, const double* pt, double* grad, int g, bool active,
                                                   double* ws, size_t wstride, unsigned wlane) {
  const int kk = g >> 2, dim0 = g & 3;
  double J0 = 0.0;
  const int rounds = (S + 16) >> 4;
  for (int r = 0; r < rounds; ++r) {
    const int k = kk + r * 16;
    double Jk = 0.0;
    if (active && k <= S && (k == 0 || S > 1)) Jk = primal_cost_lane(mask, vals, v0, S, d, pt, k, dim0, ws, wstride, wlane);
    Jk += dpp_move<0xB1>(Jk);
    Jk += dpp_move<0x4E>(Jk);
    if (r == 0) J0 = row_value(Jk, 0);
    if (active && dim0 == 0 && k >= 1 && k <= S) grad[k - 1] = (S > 1) ? (Jk - J0) / kGradStep : 0.0;
  }
  return J0;
}

// evaluate_careful for ANY fixed / free pattern (paths with a vertex whose position is free, optimize_general_kernel): the
// 5 x 5-block solve of mrs_tg_general.hpp at every perturbed time vector, the cost from the coefficients
__device__ __forceinline__ double evaluate_general(const uint8_t* __restrict__ mask, const double* __restrict__ vals, int v0,
                                                   int S, int d, const double* pt, double* grad, int g, bool active,
                                                   double* ws, size_t wstride, unsigned wlane) {
  const int kk = g >> 2, dim0 = g & 3;
  const double corr = kGradStep / ((double)S - 1.0);
  double J0 = 0.0;
  const int rounds = (S + 16) >> 4;
  for (int r = 0; r < rounds; ++r) {
    const int k = kk + r * 16;
    double Jk = 0.0;
    if (active && k <= S && (k == 0 || S > 1))
      Jk = general_solve_lane<false>(mask, vals, v0, S, d, dim0, [&](int i) { return perturbed_time(pt, i, k, corr); }, ws,
                                     wstride, wlane, nullptr);
    Jk += dpp_move<0xB1>(Jk);
    Jk += dpp_move<0x4E>(Jk);
    if (r == 0) J0 = row_value(Jk, 0);
    if (active && dim0 == 0 && k >= 1 && k <= S) grad[k - 1] = (S > 1) ? (Jk - J0) / kGradStep : 0.0;
  }
  return J0;
}

// ---- lean sweeps (large batches of plain paths) --------------------------------------------------------------------
// The sweeping evaluation of evaluate_objective<1> for plain paths only (start | interior ... | end): the block constants
// come from the LDS table, the brackets from dp on the fly, and nothing of the general step is compiled in -- which is what
// keeps the kernel at <= 256 VGPRs, i.e. TWO wavefronts per SIMD where optimize_compact_kernel (436 VGPRs) has one.  The
// kernels are bound by the dependent chain of a step, not by issue: the second wavefront runs in
// the first one's idle issue slots.  Lane k of a group of G sweeps time vector k (k, k + G, ... when S + 1 > G).
constexpr int kLeanPub = 3 * 28;  // evaluate_lean_shared's hand-over area: three half-sweep states of 28 doubles
// behind it: the start vertex's constrained derivative values of a path that starts from a MOVING state (a replanning request
// in flight: velocity / acceleration / jerk of the first vertex constrained to non-zero values) -- [0] non-zero = there are
// such values, [2 + c * 4 + q] = derivative c + 1 of dimension q (zero where the slot is free)
constexpr int kLeanMoving = 18;
__host__ __device__ constexpr int lean_eval_doubles(int Sb) { return 4 * Sb + 4 * (Sb + 1) + kLeanPub + kLeanMoving; }  // dp, staging area, hand-over, moving start

// MRS_TG_LEAN_CONST_TABLE=1 (an A / B build, NOT the default): the 45 table constants as compile-time operands -- SGPR pairs the
// scalar unit sets up -- instead of 90 VGPRs per lane, and ONE table for both sweep directions (mrs_tg_sweep.hpp, PsTab).
// Bit-identical results; measured in round 5 (same box, scripts/lean_ab.py): optimize_lean_shared_kernel 256 VGPRs + 80 B of
// scratch -> 202 VGPRs without scratch and 3 % SLOWER (8192 x 10: 126.4 -> 130.9 us, 65536 x 10: 569 -> 575 us), the mixed
// kernel of ragged batches 9 % slower (334 -> 365 us); compiled for three wavefronts per SIMD (168 VGPRs + 80 B) slower again
// (149 us).  The kernels issue one FP64 instruction per 6.2 SIMD cycles with two wavefronts resident against 4.9-5.1 at best
// (scripts/dpp_probe.hip): bound by their instruction count, not by registers or occupancy -- and the two s_mov_b32 per
// constant use are instructions too.
#ifndef MRS_TG_LEAN_CONST_TABLE
#define MRS_TG_LEAN_CONST_TABLE 0
#endif
#define LEAN1_TAB(e) (MRS_TG_LEAN_CONST_TABLE ? PsTab<4>::at(e) : tab[e])
template <bool MASKED>
__device__ __forceinline__ double evaluate_lean(const double* tabs, const double* ev, int S, int Sb, int d, const double* pt,
                                                double* grad, int g, int G, bool active, int* tripped) {
  const double* dp = ev;
  // MASKED: free masks of the two end vertices (stage_ps); 0 = fully constrained, the plain start / end step
  const unsigned m_first = MASKED ? (unsigned)ev[4 * (size_t)Sb] : 0u, m_last = MASKED ? (unsigned)ev[4 * (size_t)Sb + 1] : 0u;
  const double* tab = tabs;  // left-to-right table
  const double corr = kGradStep / ((double)S - 1.0);
  double J0 = 0.0;
  const int g_shift = __builtin_ctz((unsigned)G);  // G is a power of two
  const int rounds = (S + G) >> g_shift;
  for (int r = 0; r < rounds; ++r) {
    const int k = g + (r << g_shift);
    double Jk = 0.0, qfk = 0.0;
    if (!MASKED && active && k <= S && (k == 0 || S > 1)) {
      // (the table: compile-time constants of order 4 -- the plain-path kernels never run another order; mrs_tg_sweep.hpp, PsTab)
      // The plain sweep (start | interior ... | end), written out for four dimensions per lane with the smallest live set
      // the step allows.  The right-hand-side brackets are not materialised per (dimension, row) (FastStep::w: 36 doubles
      // that lived across the whole step and pushed the kernel to 256 VGPRs + 116 bytes of scratch): row r of every
      // dimension shares cN[r] = HBAR[1+r][0] T^(r+2-2d), so y[r][q] -= cN[r] dp[q] is one multiply per row and one FMA
      // per (row, dimension) -- 8 multiplies + 32 FMAs per step where the bracket form had 32 + 32 -- and the f^T H f term
      // of a segment is one FMA with the staged HBAR[0][0] |dp|^2.
      const double* qs = ev + 4 * (size_t)Sb + 2;  // HBAR[0][0] |dp_i|^2 per segment (stage_ps)
      double Sm[10], y[kNB][4], qf, red = 0.0;
      {  // segment 0: the start vertex is fully constrained, the state moves to vertex 1
        double p2[9];
        segment_powers_c<4>(perturbed_time(pt, 0, k, corr), p2);
        qf = p2[0] * qs[0];
#pragma unroll
        for (int r = 0; r < kNB; ++r) {
#pragma unroll
          for (int c = 0; c <= r; ++c) Sm[tri(r, c)] = LEAN1_TAB(26 + tri(r, c)) * p2[r + c + 2];
          const double cF = LEAN1_TAB(40 + r) * p2[r + 1];
#pragma unroll
          for (int q = 0; q < 4; ++q) y[r][q] = -(cF * dp[q]);
        }
      }
      for (int i = 1; i < S; ++i) {
        double p2[9];
        segment_powers_c<4>(perturbed_time(pt, i, k, corr), p2);
        double dq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) dq[q] = dp[i * 4 + q];
        qf = fma(p2[0], qs[i], qf);
        // vertex i: its block and right-hand side are complete with this segment's near part
#pragma unroll
        for (int r = 0; r < kNB; ++r) {
#pragma unroll
          for (int c = 0; c <= r; ++c) Sm[tri(r, c)] = fma(LEAN1_TAB(tri(r, c)), p2[r + c + 2], Sm[tri(r, c)]);
          const double cN = LEAN1_TAB(36 + r) * p2[r + 1];
#pragma unroll
          for (int q = 0; q < 4; ++q) y[r][q] = fma(-cN, dq[q], y[r][q]);
        }
        // Cholesky of the block (only the off-diagonal entries and the reciprocal diagonal are ever used), z = L^-1 y
        double L[10], Linv[kNB], z[kNB][4];
#pragma unroll
        for (int c = 0; c < kNB; ++c) {
          double dsum = Sm[tri(c, c)];
#pragma unroll
          for (int m = 0; m < c; ++m) dsum = fma(-L[tri(c, m)], L[tri(c, m)], dsum);
          const double inv = rsqrt_refined(dsum);
          Linv[c] = inv;
#pragma unroll
          for (int rr = c + 1; rr < kNB; ++rr) {
            double t = Sm[tri(rr, c)];
#pragma unroll
            for (int m = 0; m < c; ++m) t = fma(-L[tri(rr, m)], L[tri(c, m)], t);
            L[tri(rr, c)] = t * inv;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int rr = 0; rr < kNB; ++rr) {
            double t = y[rr][q];
#pragma unroll
            for (int m = 0; m < rr; ++m) t = fma(-L[tri(rr, m)], z[m][q], t);
            z[rr][q] = t * Linv[rr];
            red = fma(z[rr][q], z[rr][q], red);
          }
        if (i < S - 1) {  // interior vertex: W = L^-1 E, then the Schur complement and right-hand side of vertex i + 1
          double W[kNB][kNB];
#pragma unroll
          for (int c = 0; c < kNB; ++c)
#pragma unroll
            for (int rr = 0; rr < kNB; ++rr) {
              double t = LEAN1_TAB(10 + rr * kNB + c) * p2[rr + c + 2];
#pragma unroll
              for (int m = 0; m < rr; ++m) t = fma(-L[tri(rr, m)], W[m][c], t);
              W[rr][c] = t * Linv[rr];
            }
#pragma unroll
          for (int rr = 0; rr < kNB; ++rr) {
#pragma unroll
            for (int c = 0; c <= rr; ++c) {
              double t = LEAN1_TAB(26 + tri(rr, c)) * p2[rr + c + 2];
#pragma unroll
              for (int m = 0; m < kNB; ++m) t = fma(-W[m][rr], W[m][c], t);
              Sm[tri(rr, c)] = t;
            }
            const double cF = LEAN1_TAB(40 + rr) * p2[rr + 1];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              double t = -(cF * dq[q]);
#pragma unroll
              for (int m = 0; m < kNB; ++m) t = fma(-W[m][rr], z[m][q], t);
              y[rr][q] = t;
            }
          }
        }
      }
      Jk = 0.5 * (qf - red);
      qfk = qf;
    }
    if (MASKED && active && k <= S && (k == 0 || S > 1)) {
      Elim<4> st;
      st.init();
      for (int i = 0; i < S; ++i) {
        double p2[9];
        segment_powers(perturbed_time(pt, i, k, corr), d, p2);
        FastStep<4> fast;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double dq = dp[i * 4 + q];
#pragma unroll
          for (int j = 0; j < 8; ++j) fast.w[q][j] = tab[36 + j] * dq;
          fast.w[q][8] = tab[44] * dq * dq;
        }
        if (i == 0) {
          if (m_first != 0u) fast.template masked_t<false>(st, tab, p2, m_first, 0xFu);
          else fast.template start_t<false>(st, tab, p2);
        } else if (i < S - 1) {
          fast.template interior_t<false>(st, tab, p2);
        } else if (m_last != 0u) {
          fast.template masked_t<false>(st, tab, p2, 0xFu, m_last);
          double L[10], z[kNB][4];
          st.factor_vertex(m_last, L, z);
        } else {
          fast.end(st, tab, p2);
        }
      }
      Jk = 0.5 * (st.qf - st.red);
      qfk = st.qf;
    }
    Jk = guarded_cost(Jk, qfk, k == 0);
    if (tripped && active && k <= S && Jk == kUnreliableCost) *tripped = 1;
    if (r == 0) J0 = __shfl(Jk, (int)(threadIdx.x & ~(unsigned)(G - 1)), 64);
    if (active && k >= 1 && k <= S) grad[k - 1] = (S > 1) ? (Jk - J0) / kGradStep : 0.0;
  }
  return J0;
}

// ---- lean sweeps with shared halves ------------------------------------------------------------------------------------
// The S + 1 time vectors of an evaluation share their half sweeps across the middle vertex m = S / 2 (mrs_tg_wave.hip has the
// argument): m + 2 distinct left halves (x, B', one per left segment) and S - m + 2 right ones.  Here every lane runs ONE half
// sweep for all four dimensions -- S + 4 lanes of the group instead of S + 1, each for S - m steps and a join instead of S
// steps: 0.6 of the lane-steps at S = 10.  A lane's direction is data: it reads the table of its direction (stage_ps_tables)
// into its registers; dp and the f^T H f terms do not depend on the direction.  The three half sweeps that more than one vector
// needs (right half of x, right half of B', left half of B') go through LDS (28 doubles each), the other S + 1 lanes add
// their partner's state to their own and factor the middle vertex.  For plain paths with 4 <= S and S + 4 <= G.
//   lanes of the group: 0: x left | 1 .. m: vector k = lane, left | m + 1: B' left |
//                       m + 2: x right | m + 2 + r: vector m + r, right (r = 1 .. S - m) | S + 3: B' right
#if MRS_TG_LEAN_CONST_TABLE
#define LEAN_TAB(e) PsTab<4>::at(e)
#else
#define LEAN_TAB(e) Tab_regs.at(e)
#endif
// ENDS: the two end vertices may leave derivative slots free (rest-to-rest paths under an objective order below snap: jerk
// and / or snap stay free there; stage_ps leaves the two free masks in front of qs).  The first step of a half sweep then
// eliminates the end vertex like any other vertex -- the block of the segment's near part with the rows and columns of the
// constrained slots replaced by the identity, and their reciprocal pivots by zero, so that they contribute nothing to W, z
// and the cost -- instead of moving the state to the next vertex; every other step is the plain one.
// MOVING: the path may start from a moving state (stage_ps has left the start vertex's constrained derivative values f behind
// the hand-over area).  Nothing changes in the elimination; the first step of the half sweeps that start at vertex 0 gets
// the terms of f -- right-hand side of vertex 1: - sum_c E[c][r] f_c T^(r+c+2-2d); of vertex 0's own free slots (ENDS):
// - sum_c N[r][c] f_c T^(r+c+2-2d); f^T H f: sum_c 2 f_c N0[c] dp T^(c+1-2d) + sum_cc' f_c N[c][c'] f_c' T^(c+c'+2-2d) -- as
// FastStep::start_state has them (mrs_tg_sweep.hpp), from the table entries instead of staged products.
// One pass: lane `g` of the group runs the half sweep of VIRTUAL lane gv (the numbering above); `j0_lane` = the wavefront lane that
// holds virtual lane 0 in this pass, or have_j0: the cost of x comes from an earlier pass (j0_in).
template <bool ENDS, bool MOVING>
__device__ __forceinline__ double evaluate_lean_shared_pass(const double* tabs, const double* ev, double* pub, int S, int Sb, int d,
                                                            const double* pt, double* grad, int gv, bool active, int* tripped,
                                                            int j0_lane, bool have_j0, double j0_in) {
  static_assert(!(ENDS && MRS_TG_LEAN_CONST_TABLE), "the compile-time table is the order-4 one");
  const double* mv = pub + kLeanPub;
  const bool moving = MOVING && mv[0] != 0.0;
  const double* dp = ev;
  const double* qs = ev + 4 * (size_t)Sb + 2;  // HBAR[0][0] |dp_i|^2 per segment (stage_ps)
  const int m = S >> 1, nL = m + 2;
  const bool left = gv < nL;
  const int r = left ? gv : gv - nL;
  const int nhalf = left ? m : S - m;
  const bool valid = active && gv < S + 4;
  const bool pure = r == nhalf + 1, base = r == 0;
  const int k = base ? 0 : pure ? (left ? S : 1) : (left ? r : m + r);
  const double corr = kGradStep / ((double)S - 1.0);
  // the table: compile-time constants of order 4, the same for both directions (mrs_tg_sweep.hpp, PsTab): a right-to-left
  // lane sweeps in sign-transformed variables and `flip` = -1 puts its state into the common frame where it meets a left one
#if MRS_TG_LEAN_CONST_TABLE
  using Tab = PsTab<4>;
  const double flip = left ? 1.0 : -1.0;
  (void)tabs;
  (void)d;
#else  // (A / B build: the per-lane table of its direction in 90 registers, as until round 4)
  struct {
    double t[kPsTable];
    __device__ __forceinline__ double at(int e) const { return t[e]; }
  } Tab_regs;
  {
    const double* tsrc = tabs + (left ? 0 : kPsTable);
#pragma unroll
    for (int e = 0; e < kPsTable; ++e) Tab_regs.t[e] = tsrc[e];
  }
  const double flip = 1.0;
#endif
  double Sm[10], y[kNB][4], qf = 0.0, red = 0.0;
#pragma unroll
  for (int e = 0; e < 10; ++e) Sm[e] = 0.0;
#pragma unroll
  for (int rr = 0; rr < kNB; ++rr)
#pragma unroll
    for (int q = 0; q < 4; ++q) y[rr][q] = 0.0;
  int i = left ? 0 : S - 1;
  const int di = left ? 1 : -1;
  // ENDS: free slots of the end vertex this lane starts from (bit r: derivative r + 1), and the pivots' scale: 0 for a
  // constrained slot of the end vertex in the first step, 1 everywhere else
  unsigned fm = 0u;
  double rs[kNB];
#pragma unroll
  for (int rr = 0; rr < kNB; ++rr) rs[rr] = 1.0;
  if (ENDS) fm = (unsigned)ev[4 * (size_t)Sb + (left ? 0 : 1)] & 0xFu;
  for (int s = 0; __ballot(valid && s < nhalf) != 0ull; ++s) {
    if (valid && s < nhalf) {
      double T = pt[i];
      if (k > 0) T = (i == k - 1) ? T + kGradStep : fmax(T - corr, kTimeLowerBound);
      double p2[9];
      if (ENDS) segment_powers(T, d, p2);
      else segment_powers_c<4>(T, p2);
      double dq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) dq[q] = dp[i * 4 + q];
      qf = fma(p2[0], qs[i], qf);
      // ENDS: the free mask of the vertex this step eliminates -- the end vertex the lane starts from (0: fully constrained,
      // nothing to eliminate), else the vertex the sweep stands on (0xF unless it is a stop_at vertex)
      unsigned fmv = fm;
      if (ENDS && s > 0) fmv = (unsigned)ev[4 * (size_t)Sb + 2 + Sb + (left ? i : i + 1)] & 0xFu;
      const bool free_end = ENDS && (s == 0 ? fm != 0u : fmv != 0xFu);  // (a step whose vertex has constrained slots)
      if (s == 0 && !free_end) {  // the end vertex is fully constrained: the state moves to the next vertex
#pragma unroll
        for (int rr = 0; rr < kNB; ++rr) {
#pragma unroll
          for (int c = 0; c <= rr; ++c) Sm[tri(rr, c)] = LEAN_TAB(26 + tri(rr, c)) * p2[rr + c + 2];
          const double cF = LEAN_TAB(40 + rr) * p2[rr + 1];
#pragma unroll
          for (int q = 0; q < 4; ++q) y[rr][q] = -(cF * dq[q]);
        }
      } else {
        // the vertex the sweep stands on: its block and right-hand side are complete with this segment's near part
        // (a free end vertex: Sm and y are still zero, the near part IS the block -- with the identity in the constrained slots)
#pragma unroll
        for (int rr = 0; rr < kNB; ++rr) {
#pragma unroll
          for (int c = 0; c <= rr; ++c) Sm[tri(rr, c)] = fma(LEAN_TAB(tri(rr, c)), p2[rr + c + 2], Sm[tri(rr, c)]);
          const double cN = LEAN_TAB(36 + rr) * p2[rr + 1];
#pragma unroll
          for (int q = 0; q < 4; ++q) y[rr][q] = fma(-cN, dq[q], y[rr][q]);
        }
        if (ENDS && MOVING && s == 0 && free_end && moving && left) {  // the start vertex's own free slots see its constrained values
#pragma unroll
          for (int rr = 0; rr < kNB; ++rr)
#pragma unroll
            for (int c = 0; c < kNB; ++c) {
              const double nrc = LEAN_TAB(rr >= c ? tri(rr, c) : tri(c, rr)) * p2[rr + c + 2];
#pragma unroll
              for (int q = 0; q < 4; ++q) y[rr][q] = fma(-nrc, mv[2 + c * 4 + q], y[rr][q]);
            }
        }
        if (ENDS) {
#pragma unroll
          for (int rr = 0; rr < kNB; ++rr) rs[rr] = (free_end && !((fmv >> rr) & 1u)) ? 0.0 : 1.0;
          if (free_end) {
#pragma unroll
            for (int rr = 0; rr < kNB; ++rr)
#pragma unroll
              for (int c = 0; c <= rr; ++c)
                if (!((fmv >> rr) & 1u) || !((fmv >> c) & 1u)) Sm[tri(rr, c)] = (rr == c) ? 1.0 : 0.0;
          }
        }
        double L[10], Linv[kNB], z[kNB][4], W[kNB][kNB];
#pragma unroll
        for (int c = 0; c < kNB; ++c) {
          double dsum = Sm[tri(c, c)];
#pragma unroll
          for (int mm = 0; mm < c; ++mm) dsum = fma(-L[tri(c, mm)], L[tri(c, mm)], dsum);
          const double inv = rsqrt_refined(dsum);
          Linv[c] = ENDS ? inv * rs[c] : inv;
#pragma unroll
          for (int rr = c + 1; rr < kNB; ++rr) {
            double t = Sm[tri(rr, c)];
#pragma unroll
            for (int mm = 0; mm < c; ++mm) t = fma(-L[tri(rr, mm)], L[tri(c, mm)], t);
            L[tri(rr, c)] = t * inv;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int rr = 0; rr < kNB; ++rr) {
            double t = y[rr][q];
#pragma unroll
            for (int mm = 0; mm < rr; ++mm) t = fma(-L[tri(rr, mm)], z[mm][q], t);
            z[rr][q] = t * Linv[rr];
            red = fma(z[rr][q], z[rr][q], red);
          }
#pragma unroll
        for (int c = 0; c < kNB; ++c)
#pragma unroll
          for (int rr = 0; rr < kNB; ++rr) {
            double t = LEAN_TAB(10 + rr * kNB + c) * p2[rr + c + 2];
#pragma unroll
            for (int mm = 0; mm < rr; ++mm) t = fma(-L[tri(rr, mm)], W[mm][c], t);
            W[rr][c] = t * Linv[rr];
          }
#pragma unroll
        for (int rr = 0; rr < kNB; ++rr) {
#pragma unroll
          for (int c = 0; c <= rr; ++c) {
            double t = LEAN_TAB(26 + tri(rr, c)) * p2[rr + c + 2];
#pragma unroll
            for (int mm = 0; mm < kNB; ++mm) t = fma(-W[mm][rr], W[mm][c], t);
            Sm[tri(rr, c)] = t;
          }
          const double cF = LEAN_TAB(40 + rr) * p2[rr + 1];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            double t = -(cF * dq[q]);
#pragma unroll
            for (int mm = 0; mm < kNB; ++mm) t = fma(-W[mm][rr], z[mm][q], t);
            y[rr][q] = t;
          }
        }
      }
      if (MOVING && s == 0 && moving && left) {  // vertex 1's right-hand side and f^T H f: the terms of the moving start
#pragma unroll
        for (int c = 0; c < kNB; ++c) {
          double fc[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) fc[q] = mv[2 + c * 4 + q];
#pragma unroll
          for (int rr = 0; rr < kNB; ++rr) {
            const double ecr = LEAN_TAB(10 + c * kNB + rr) * p2[rr + c + 2];  // near slot c (vertex 0) x far slot rr (vertex 1)
#pragma unroll
            for (int q = 0; q < 4; ++q) y[rr][q] = fma(-ecr, fc[q], y[rr][q]);
          }
          const double n0 = 2.0 * (LEAN_TAB(36 + c) * p2[c + 1]);
#pragma unroll
          for (int q = 0; q < 4; ++q) qf = fma(n0 * fc[q], dq[q], qf);
#pragma unroll
          for (int c2 = 0; c2 < kNB; ++c2) {
            const double ncc = LEAN_TAB(c >= c2 ? tri(c, c2) : tri(c2, c)) * p2[c + c2 + 2];
#pragma unroll
            for (int q = 0; q < 4; ++q) qf = fma(ncc * fc[q], mv[2 + c2 * 4 + q], qf);
          }
        }
      }
    }
    i += di;
  }
  // hand-over: slot 0 right half of x | 1 right half of B' | 2 left half of B'
  const bool base_right = !left && base;
  if (valid && (pure || base_right)) {
    double* ps = pub + (base_right ? 0 : (left ? 2 : 1)) * 28;
    lean_flip_state(Sm, y, flip);  // a right lane's state leaves in the common (left-to-right) frame
#pragma unroll
    for (int e = 0; e < 10; ++e) ps[e] = Sm[e];
#pragma unroll
    for (int rr = 0; rr < kNB; ++rr)
#pragma unroll
      for (int q = 0; q < 4; ++q) ps[10 + rr * 4 + q] = y[rr][q];
    ps[26] = qf;
    ps[27] = red;
  }
  ps_wave_sync();
  const bool join = valid && !pure && !base_right;
  double Jk = 0.0, qfk = 0.0;
  if (join) {
    const double* ps = pub + (left ? (base ? 0 : 1) : 2) * 28;
    lean_flip_state(Sm, y, flip);  // a right lane joins in the common frame as well (the factorisation's |z|^2 is frame-independent)
#pragma unroll
    for (int e = 0; e < 10; ++e) Sm[e] += ps[e];
#pragma unroll
    for (int rr = 0; rr < kNB; ++rr)
#pragma unroll
      for (int q = 0; q < 4; ++q) y[rr][q] += ps[10 + rr * 4 + q];
    qf += ps[26];
    red += ps[27];
    // ENDS: the middle vertex may be a stop_at vertex too (identity in its constrained slots, zero for their reciprocal pivots)
    unsigned fmm = 0xFu;
    if (ENDS) {
      fmm = (unsigned)ev[4 * (size_t)Sb + 2 + Sb + m] & 0xFu;
      if (fmm != 0xFu) {
#pragma unroll
        for (int rr = 0; rr < kNB; ++rr)
#pragma unroll
          for (int c = 0; c <= rr; ++c)
            if (!((fmm >> rr) & 1u) || !((fmm >> c) & 1u)) Sm[tri(rr, c)] = (rr == c) ? 1.0 : 0.0;
      }
    }
    double L[10], Linv[kNB], z[kNB][4];
#pragma unroll
    for (int c = 0; c < kNB; ++c) {
      double dsum = Sm[tri(c, c)];
#pragma unroll
      for (int mm = 0; mm < c; ++mm) dsum = fma(-L[tri(c, mm)], L[tri(c, mm)], dsum);
      const double inv = rsqrt_refined(dsum);
      Linv[c] = (ENDS && !((fmm >> c) & 1u)) ? 0.0 : inv;
#pragma unroll
      for (int rr = c + 1; rr < kNB; ++rr) {
        double t = Sm[tri(rr, c)];
#pragma unroll
        for (int mm = 0; mm < c; ++mm) t = fma(-L[tri(rr, mm)], L[tri(c, mm)], t);
        L[tri(rr, c)] = t * inv;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int rr = 0; rr < kNB; ++rr) {
        double t = y[rr][q];
#pragma unroll
        for (int mm = 0; mm < rr; ++mm) t = fma(-L[tri(rr, mm)], z[mm][q], t);
        z[rr][q] = t * Linv[rr];
        red = fma(z[rr][q], z[rr][q], red);
      }
    Jk = 0.5 * (qf - red);
    qfk = qf;
  }
  ps_wave_sync();  // (the area is written again by the next evaluation)
  Jk = guarded_cost(Jk, qfk, k == 0);
  if (tripped && join && Jk == kUnreliableCost) *tripped = 1;
  const double J0 = have_j0 ? j0_in : __shfl(Jk, j0_lane, 64);  // virtual lane 0: x, joined
  if (join && k >= 1) grad[k - 1] = (Jk - J0) / kGradStep;
  return J0;
}


// The evaluation of one path.  S + 4 <= G: one pass, lane g = virtual lane g.  A path of 61 .. 121 segments has more half sweeps
// than a wavefront has lanes (G = 64, one path per wavefront): TWO passes, each of which runs the three half sweeps that more
// than one vector needs (virtual lanes m + 1, m + 2 and S + 3: their lanes publish them again -- three lanes of 64) next to
// half of the other S + 1, so that every pass is complete in itself: S steps per evaluation where the one-sided sweeps of such
// a path need 2 S (one 80-segment request 0.79 -> ms, round 5).
// TWOPASS is a template parameter: the loop around the pass, even with one iteration, cost the min-snap kernel of the benchmark
// configs 512 bytes of scratch per lane (16 -> 528); launches without such paths run the instantiations without it.
template <bool ENDS, bool MOVING = false, bool TWOPASS = false>
__device__ __forceinline__ double evaluate_lean_shared(const double* tabs, const double* ev, double* pub, int S, int Sb, int d,
                                                       const double* pt, double* grad, int g, int G, bool active, int* tripped) {
  const int lane0 = (int)(threadIdx.x & ~(unsigned)(G - 1));  // lane 0 of the group
  if (!TWOPASS)
    return evaluate_lean_shared_pass<ENDS, MOVING>(tabs, ev, pub, S, Sb, d, pt, grad, g, active, tripped, lane0, false, 0.0);
  const bool two = G == 64 && __builtin_amdgcn_readfirstlane(S + 4 > G ? 1 : 0) != 0;  // (wave-uniform: one path per wavefront)
  const int m = S >> 1;
  double J0 = 0.0;
  for (int pass = 0; pass < (two ? 2 : 1); ++pass) {
    // two passes -- lanes 0 .. 2: the publishers; lane 3 + o: the o-th of the other virtual lanes ([0, m] and [m + 3, S + 2]),
    // 61 per pass
    const int o = g - 3 + pass * 61;
    const int gv = !two ? g : (g == 0 ? m + 1 : g == 1 ? m + 2 : g == 2 ? S + 3 : (o <= m ? o : o + 2));
    const bool on = !two || g < 3 || o <= S;
    const double J = evaluate_lean_shared_pass<ENDS, MOVING>(tabs, ev, pub, S, Sb, d, pt, grad, on ? gv : S + 4, active, tripped,
                                                            two ? lane0 + 3 : lane0, pass == 1, J0);
    if (pass == 0) J0 = J;
  }
  return J0;
}

#undef LEAN_TAB
// per-group LDS of the plain-path kernels and their staging
__host__ __device__ constexpr int lean_group_doubles(int Sb) {
  return (5 + 2 * kLbfgsM) * Sb + (kLbfgsM + 1) + kTickState + lean_eval_doubles(Sb);
}

// positions -> dp of every segment; returns whether the path is one the evaluation takes (group-uniform)
// end_masks: the two end vertices may leave slots free (rest-to-rest paths under an objective order below snap: jerk and /
// or snap stay free there); their free masks are left at ev[4 Sb] and ev[4 Sb + 1] for evaluate_lean<true>
// moving: where the evaluation takes a start vertex whose constrained derivatives carry non-zero values (evaluate_lean_shared):
// the area they are left in (kLeanMoving doubles), else nullptr and such a path is not taken
__device__ __forceinline__ bool stage_ps(const uint8_t* __restrict__ mask, const double* __restrict__ vals, int v0, int S,
                                         int Sb, double* ev, int g, int G, bool active, int min_segments, int d,
                                         bool end_masks = false, double* moving = nullptr, bool interior_masks = false) {
  double* dp = ev;
  double* tmp = ev + 4 * (size_t)Sb;  // the record area, not in use yet
  int ok = (S >= min_segments) ? 1 : 0;
  unsigned fb_first = 0u, fb_last = 0u, my_fb = 0xFu, my_fb2 = 0xFu;  // (my_fb, my_fb2: of vertex g and of vertex g + G)
  if (active)
    for (int v = g; v <= S; v += G) {
      double f[kHalf][kD];
      bool pf;
      const unsigned fb = load_vertex<kD>(mask, vals, v0 + v, 0, f, pf);
      double nz = 0.0;
#pragma unroll
      for (int k = 1; k < kHalf; ++k)
#pragma unroll
        for (int q = 0; q < kD; ++q) nz += fabs(f[k][q]);
      const bool end = v == 0 || v == S;
      // (interior_masks: interior vertices may hold slots constrained to zero as well -- stop_at vertices: velocity =
      // acceleration = jerk = 0, snap free; every vertex's free mask goes behind the f^T H f terms, tmp[2 + Sb + v])
      const bool pattern_ok = end ? (end_masks || fb == 0u) : (interior_masks || fb == 0xFu);
      if (v == g) my_fb = fb;  // (a path has at most 2 G vertices where the masks are used: a lane loads at most two)
      else my_fb2 = fb;
      if (v == 0 && moving != nullptr) {  // (load_vertex has zeroed the values of unconstrained slots)
        moving[0] = nz;
#pragma unroll
        for (int k = 1; k < kHalf; ++k)
#pragma unroll
          for (int q = 0; q < kD; ++q) moving[2 + (k - 1) * kD + q] = f[k][q];
        nz = 0.0;
      }
      if (!(pf && nz == 0.0 && pattern_ok)) ok = 0;
      if (v == 0) fb_first = fb;
      if (v == S) fb_last = fb;
#pragma unroll
      for (int q = 0; q < kD; ++q) tmp[v * 4 + q] = f[0][q];
    }
  const bool all_ok = group_and(ok, G) != 0;
  ps_wave_sync();
  if (active)
    for (int i = g; i < S; i += G)
#pragma unroll
      for (int q = 0; q < kD; ++q) dp[i * 4 + q] = tmp[i * 4 + q] - tmp[(i + 1) * 4 + q];
  ps_wave_sync();
  if (end_masks && active) {
    if (g == 0) tmp[0] = (double)fb_first;
    if (g == (S & (G - 1))) tmp[1] = (double)fb_last;  // the lane that loaded vertex S
    if (g <= S) tmp[2 + Sb + g] = (double)my_fb;       // every vertex's free mask (interior: 0xF unless it is a stop_at vertex)
    if (g + G <= S) tmp[2 + Sb + g + G] = (double)my_fb2;
  }
  if (active) {  // f^T HBAR f of every segment (position terms only: all a plain path has), behind the two mask slots
    const double h00 = c_hbar[d][0][0];
    for (int i = g; i < S; i += G) {
      double n2 = 0.0;
#pragma unroll
      for (int q = 0; q < kD; ++q) n2 = fma(dp[i * 4 + q], dp[i * 4 + q], n2);
      tmp[2 + i] = h00 * n2;
    }
  }
  ps_wave_sync();
  return all_ok;
}

// ---------------------------------------------------------------------------------------------
// the time a search starts segment i from: the incoming one, or -- the call asked for the estimate and no launch computed it
// in front of this kernel (NonlinearParams::estimate_wp) -- estimateSegmentTimesEuclidean of the segment's waypoints
// (a real call: inlined, the estimate's trigonometry raised the lean kernels' scratch from 72-96 to 168-180 bytes per lane)
__device__ __attribute__((noinline)) double start_time_estimate(const double* wp, const double* lim) {
  return estimate_segment_time(wp, lim);
}
__device__ __forceinline__ double start_time(const NonlinearParams& prm, const double* t_src, const PathRef& pr, int i) {
  return prm.estimate_wp ? start_time_estimate(prm.estimate_wp + (size_t)(pr.v0 + i) * 4, prm.estimate_limits + (size_t)pr.p * 9)
                         : t_src[pr.s0 + i];
}

// the outer-loop kernel: optimiser ticks (one objective evaluation each).  On exit seg_times holds the
// last evaluated point and opt_status the stopping reason (-2: start rejected, as NLopt would).

// All bins of a plan run in ONE launch: blocks [block_begin, block_begin + n_blocks) of the grid belong to a
// bin (paths sorted longest first, so the long paths are dispatched first and the short ones fill the tail).
struct BinTable {
  int n;
  int group[5], q_begin[5], q_count[5], max_S[5], block_begin[5];
};

// CAREFUL (optimize_careful_kernel): one listed path per workgroup of 64, the primal cost in every evaluation, `careful_ws`
// the factor store of its lanes.
// LEAN (optimize_lean_kernel): plain paths only, the lean sweeps of evaluate_lean; a path it does not take is flagged in
// `fallback` and left to the general instantiation launched behind it.
// GENERAL (optimize_general_kernel, with CAREFUL): the listed paths are those with a position-free vertex, every evaluation
// is evaluate_general, and the start point is read from `start_times` (a copy taken before the fast kernels ran over the
// batch: they do not know such paths and leave garbage in seg_times for them).
// LEANSHARED (optimize_lean_shared_kernel, with LEAN): every bin of the launch holds paths of 4 <= S <= G - 4 segments only (the
// host checks), so evaluate_lean_shared is the ONLY evaluation compiled in -- next to the one-sided sweeps it costs both their
// registers (124 instead of 28 bytes of scratch).
template <int DS, bool MASKED4 = false, bool CAREFUL = false, bool LEAN = false, bool GENERAL = false, bool LEANSHARED = false,
          bool TWOPASS = false>
__device__ __forceinline__ void optimize_body(const BatchView& b, const NonlinearParams& prm, const BinTable& bins,
                                              const uint8_t* __restrict__ mask, const double* __restrict__ vals,
                                              double* __restrict__ seg_times, int32_t* __restrict__ opt_status,
                                              double* careful_ws = nullptr, int32_t* __restrict__ fallback = nullptr,
                                              const double* __restrict__ start_times = nullptr) {
  extern __shared__ double lds[];
  MRS_TG_PHASE_MARK(0);
  int bin = 0;
#pragma unroll
  for (int i = 1; i < 5; ++i)
    if (!CAREFUL && i < bins.n && (int)blockIdx.x >= bins.block_begin[i]) bin = i;
  int G = CAREFUL ? 64 : bins.group[bin];  // not const: laundered at the top of every tick, see there
  int n_listed = 0;
  if (CAREFUL) {
    n_listed = prm.careful_count[0] < prm.careful_cap ? prm.careful_count[0] : prm.careful_cap;
    if ((int)blockIdx.x >= n_listed) return;
  }
  // (careful: the workgroup's path is entry blockIdx.x of the list, a "bin" of one path that starts at its position)
  const int q_begin = CAREFUL ? prm.careful_list[blockIdx.x] : bins.q_begin[bin];
  const int q_count = CAREFUL ? 1 : bins.q_count[bin], Sb = CAREFUL ? b.max_segments : bins.max_S[bin];
  const int block_in_bin = CAREFUL ? 0 : (int)blockIdx.x - bins.block_begin[bin];
  // 64 threads: the wavefront that runs everything.  128 threads (DS = 4, one path per block): wavefront 1 is the
  // partner that runs the backward half sweeps of the two-sided evaluation and mirrors every barrier of wavefront 0.
  const bool two_wave = (DS == 4) && blockDim.x == 128;  // compile-time false for DS = 1: none of the partner code is emitted there
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  int g = lane & (G - 1);
  const int g_shift = __builtin_ctz((unsigned)G);  // G is a power of two
  const int grp = lane >> g_shift;
  const int per_block = 64 >> g_shift;
  const int qi = block_in_bin * per_block + grp;
  bool active = qi < q_count;
  int q = q_begin + (active ? qi : 0);  // (LEAN with a queue: a lane group takes further paths once its own is done)
  if (!CAREFUL && !LEAN && prm.only_flagged) {  // the plain-path kernel in front of this launch took the other paths
    active = active && prm.only_flagged[q] != 0;
    if (!__syncthreads_or(active ? 1 : 0)) return;  // (an all-plain batch: every workgroup ends here)
  }
  PathRef pr = path_at(b, q);
  int S = pr.S;
  const int d = prm.derivative;

  double* hc = lds;  // [kBlockConsts] (LEAN: the two direction tables of stage_ps_tables), shared by the groups of the block
  if (LEAN) stage_ps_tables(d, hc, lane, 64);
  else if (wave == 0) stage_block_constants(d, hc, lane, 64);
  constexpr bool kExtras = DS == 4;
  constexpr int kConsts = LEAN ? 2 * kPsTable : kBlockConsts;
  const int group_doubles = LEAN ? lean_group_doubles(Sb) : group_lds_doubles(Sb, kExtras);
  double* base = lds + kConsts + (size_t)grp * group_doubles;
  double* pair_state = lds + kConsts + (size_t)per_block * group_doubles;  // [64 * kPairState] (two_wave only)
  int* pair_flags = reinterpret_cast<int*>(pair_state + 64 * kPairState);              // [0] all done, [1] two-sided evaluation, [2] moving start / masked segments
  double* x = base;
  double* gr = x + Sb;
  double* xn = gr + Sb;
  double* gn = xn + Sb;
  double* dir = gn + Sb;
  double* sm = dir + Sb;            // [M][Sb]
  double* ym = sm + kLbfgsM * Sb;   // [M][Sb]
  double* rho = ym + kLbfgsM * Sb;  // [M]
  double* tick_f = rho + kLbfgsM + 1;                         // rho[kLbfgsM] = scaling from the newest pair (fast path)
  int* tick_i = reinterpret_cast<int*>(tick_f + 2);
  double* vtx = tick_f + kTickState;  // [(Sb + 1) * kVtxLds]
  double* seg = vtx + (size_t)(Sb + 1) * kVtxLds + (kExtras ? kStartExtra : 0);  // [Sb * kSegLds], the moving-start extras in front
  // the start times are requested in front of the vertex staging (one trip to memory instead of two in a row)
  const double* t_src = GENERAL ? start_times : seg_times;
  double t_first = 0.0;
  if (active && wave == 0 && g < S) t_first = start_time(prm, t_src, pr, g);
  if (LEAN) {  // vtx = the evaluation area: dp of every segment; the path's eligibility decides who runs it
    // (the kernel of the shared half sweeps with free end slots runs every bin of a ragged plan, in groups of S + 4 lanes:
    // paths of fewer than four segments are left to the sweeping kernel behind it)
    const bool takes = stage_ps(mask, vals, pr.v0, S, Sb, vtx, g, G, active, (LEANSHARED && MASKED4) ? prm.ends_min_segments : 2, d, MASKED4,
                                LEANSHARED ? vtx + 4 * (size_t)Sb + 4 * ((size_t)Sb + 1) + kLeanPub : nullptr, LEANSHARED && MASKED4);
    if (active && g == 0) fallback[q] = takes ? 0 : 1;
    active = active && takes;
  } else {
    if (active && wave == 0) stage_vertices(mask, vals, pr.v0, S, vtx, g, G);
    __syncthreads();
    if (active && wave == 0) stage_segments(vtx, S, d, seg, g, G, kExtras);
  }

  // ---- start point; NLopt rejects a start below the lower bound (-> INVALID_ARGS)
  int ok = 1;
  if (active && wave == 0) {
    double t_sum = 0.0;
    auto take = [&](int i, double t) {
      x[i] = t;
      xn[i] = t;
      t_sum += t;
      if (t < kTimeLowerBound) ok = 0;
    };
    if (g < S) take(g, t_first);
    for (int i = g + G; i < S; i += G) take(i, start_time(prm, t_src, pr, i));
    // the total time the search starts from: what the final solve measures a runaway of the feasibility scaling against
    t_sum = group_sum(t_sum, G);
    if (g == 0 && prm.sum_t0) prm.sum_t0[pr.p] = t_sum;
  }
  bool bad = active && wave == 0 && !group_and(ok, G);
  __syncthreads();
  MRS_TG_PHASE_MARK(1);

  const int maxeval = prm.max_iterations;
  // nlopt maxtime (src/mrs_trajectory_generation.cpp:899): the constant-rate wall clock of the device, read once per
  // evaluation; a path that is still running when it passes stops with MAXTIME_REACHED (6) at its last evaluated point
  // ONE deadline per call, written to a device word by a one-thread kernel in front of the outer-loop launches: a workgroup
  // that starts late (batches of several residency rounds, the general kernel behind the lean one, the careful re-run) is
  // measured against the call's clock, not its own
  const long long t_deadline = prm.deadline ? *prm.deadline : 0ll;
  bool timed_out = false;
  // nlopt checks the evaluation count first, then the clock (nlopt_stop_evals, nlopt_stop_time)
  auto budget_spent = [&](int n) { return (maxeval > 0 && n >= maxeval) || timed_out; };
  auto budget_code = [&](int n) { return (maxeval > 0 && n >= maxeval) ? 5 : 6; };
  const bool single = Sb <= G;  // uniform per workgroup
  int ret = -1;
  bool done = !active || bad;
  bool pair_ok = false;
  unsigned pair_special = 0u;
  if (two_wave) {
    if (threadIdx.x == 0) {
      // every perturbation needs its own quad of lanes in one round: 4 (S + 1) <= 64
      // every segment must have a specialised step: a start-type (or masked) first one, an end-type (or masked) last
      // one, interior or masked ones in between
      auto kind_of = [&](int i) { return (int)seg[(size_t)i * kSegLds + 36]; };
      bool plain = DS == 4 && G == 64 && active && S >= 4 && 4 * (S + 1) <= 64 &&
                   (kind_of(0) == kSegStart || kind_of(0) == kSegStartState || kind_of(0) == kSegMasked) &&
                   (kind_of(S - 1) == kSegEnd || kind_of(S - 1) == kSegMasked);
      for (int i = 1; plain && i < S - 1; ++i) plain = kind_of(i) == kSegInterior || kind_of(i) == kSegMasked;
      unsigned special = 0u;
      for (int i = 0; plain && i < S; ++i)
        if (kind_of(i) == kSegMasked) special |= 1u << i;
      if (plain && kind_of(0) == kSegStartState) special |= 1u << 31;
      pair_flags[2] = (int)special;
      pair_flags[0] = done ? 1 : 0;
      pair_flags[1] = plain ? 1 : 0;
    }
    __syncthreads();
    pair_ok = pair_flags[1] != 0;
    pair_special = (unsigned)__builtin_amdgcn_readfirstlane(pair_flags[2]);
    if (wave == 1) {
      // partner wavefront: backward half sweep of every tick, then the tick's other barriers
      const int k = lane >> 2, dim0 = lane & 3;
      while (pair_flags[0] == 0) {
        if (pair_ok && k <= S) {
          Elim<1> st;
          if (pair_special & 0x7FFFFFFFu) half_sweep<true, 2>(seg, hc, S, S / 2, d, xn, k, dim0, pair_special, st);
          else half_sweep<true, 0>(seg, hc, S, S / 2, d, xn, k, dim0, 0u, st);  // a moving start is the forward half's business
          double* ps = pair_state + lane;  // element-major, see evaluate_pair
#pragma unroll
          for (int e = 0; e < 10; ++e) ps[e * 64] = st.Sm[e];
#pragma unroll
          for (int r = 0; r < kNB; ++r) ps[(10 + r) * 64] = st.y[r][0];
          ps[14 * 64] = st.qf;
          ps[15 * 64] = st.red;
        }
        __syncthreads();  // hand-over
        __syncthreads();  // after the evaluation
        __syncthreads();  // after accept / reject
        __syncthreads();  // after the next trial point (and the all-done flag)
      }
      return;
    }
  }
  int neval = 0, npairs = 0, head = 0;
  double f = 0.0, alpha = 1.0;
  bool first = true;
  if (g == 0) tick_i[4] = 0;  // set by an evaluation whose by-product cost failed the guard

  // hands in the result of the group's path: the last evaluated point and the stopping reason
  auto retire = [&]() {
    // a path whose cost failed the guard somewhere is handed to the careful re-run: listed, start times left in place
    bool listed = false;
    if (!CAREFUL && prm.careful_count && !bad && tick_i[4] != 0) {
      int okl = 0;
      if (g == 0) {
        const int idx = atomicAdd(prm.careful_count, 1);
        okl = idx < prm.careful_cap;
        if (okl) prm.careful_list[idx] = q;
      }
      listed = __shfl(okl, (int)(threadIdx.x & ~(unsigned)(G - 1)), 64) != 0;
    }
    // the path's offsets are looked up again rather than carried (as addresses, in registers or scratch) through the ticks
    int q_again = q;
    asm volatile("" : "+v"(q_again));
    const PathRef pe = path_at(b, q_again);
    if (!listed)
      for (int i = g; i < S; i += G) seg_times[pe.s0 + i] = x[i];
    if (g == 0) opt_status[pe.p] = bad ? -2 : ret;
  };
  // LEAN with a queue (uniform batches of more wavefronts than the device holds at once: the launch is as many workgroups as
  // are resident, and a lane group whose path has stopped takes the next unclaimed one): `busy` = the group holds a path
  // whose result has not been handed in
  const bool queued = LEAN && prm.queue_next != nullptr;
  bool busy = active;
  bool queue_empty = false;

  while (true) {
    if (queued) {
      // Without the queue a wavefront runs until the slowest of its four paths has stopped (1 to 10 evaluations, 3.6 on
      // average) and a launch of eight residency rounds pays that maximum eight times.
      if (busy && done) {
        retire();
        busy = false;
      }
      bool want = !busy && !queue_empty;
      while (__ballot(want) != 0ull) {
        if (want) {
          int qn = 0;
          if (g == 0) qn = atomicAdd(prm.queue_next, 1);
          qn = __shfl(qn, (int)(threadIdx.x & ~(unsigned)(G - 1)), 64);
          if (qn >= q_count) {
            queue_empty = true;
            want = false;
          } else {
            q = q_begin + qn;
            pr = path_at(b, q);
            S = pr.S;
            active = true;
            double t_new = (g < S) ? start_time(prm, seg_times, pr, g) : 0.0;
            const bool takes = stage_ps(mask, vals, pr.v0, S, Sb, vtx, g, G, true, (LEANSHARED && MASKED4) ? prm.ends_min_segments : 2, d, MASKED4,
                                        LEANSHARED ? vtx + 4 * (size_t)Sb + 4 * ((size_t)Sb + 1) + kLeanPub : nullptr, LEANSHARED && MASKED4);
            if (g == 0) fallback[q] = takes ? 0 : 1;
            if (takes) {
              int okn = 1;
              double t_sum = 0.0;
              auto take = [&](int i, double t) {
                x[i] = t;
                xn[i] = t;
                t_sum += t;
                if (t < kTimeLowerBound) okn = 0;
              };
              if (g < S) take(g, t_new);
              for (int i = g + G; i < S; i += G) take(i, start_time(prm, seg_times, pr, i));
              t_sum = group_sum(t_sum, G);
              if (g == 0 && prm.sum_t0) prm.sum_t0[pr.p] = t_sum;
              bad = !group_and(okn, G);
              neval = 0;
              npairs = 0;
              head = 0;
              f = 0.0;
              alpha = 1.0;
              ret = -1;
              if (g == 0) tick_i[4] = 0;
              busy = true;
              done = bad;
              want = false;
              if (bad) {  // a start below the lower bound: nothing to search, hand it in and look for another path
                retire();
                busy = false;
                want = true;
              }
            }
            // (a path the lean sweeps do not take stays flagged for the general kernel: the group asks again)
          }
        }
      }
      ps_wave_sync();
    }
    if (__ballot(!done) == 0ull) break;
    // park the scalar state (see kTickState)
    if (g == 0) {
      tick_f[0] = f;
      tick_f[1] = alpha;
      tick_i[0] = neval;
      tick_i[1] = npairs;
      tick_i[2] = head;
      tick_i[3] = ret;
    }

    // (1) one objective evaluation at the trial point.  The lane coordinates go through an empty asm so that the
    // dozens of lane predicates derived from them (k <= S, dim0 == 0, G >= 16, ...) are recomputed where they are
    // used: hoisted out of the loop as 64-bit lane masks they overflowed the SGPR file (116 spilled SGPRs, each use
    // two v_readlane_b32)
    {
      int Gv = G;
      asm volatile("" : "+v"(g), "+v"(S), "+v"(Gv));
      G = __builtin_amdgcn_readfirstlane(Gv);
    }
    double fn;
    if (LEAN) {
      // shared half sweeps when every path of the wavefront takes them (plain paths of 4 <= S <= G - 4 segments: S + 4 lanes)
      const bool shared_path = S >= 4 && S + 4 <= G;
      if (LEANSHARED)
        fn = evaluate_lean_shared<MASKED4, true, TWOPASS>(hc, vtx, vtx + 4 * (size_t)Sb + 4 * ((size_t)Sb + 1), S, Sb, d, xn, gn, g, G, !done, tick_i + 4);
      else if (!MASKED4 && prm.lean_shared == 2 && __ballot(!done && !shared_path) == 0ull)
        fn = evaluate_lean_shared<false, false>(hc, vtx, vtx + 4 * (size_t)Sb + 4 * ((size_t)Sb + 1), S, Sb, d, xn, gn, g, G, !done, tick_i + 4);
      else
        fn = evaluate_lean<MASKED4>(hc, vtx, S, Sb, d, xn, gn, g, G, !done, tick_i + 4);
    } else if (GENERAL) {
      fn = evaluate_general(mask, vals, pr.v0, S, d, xn, gn, g, !done, careful_ws, (size_t)gridDim.x * 64,
                            blockIdx.x * 64u + (unsigned)lane);
    } else if (CAREFUL) {
      fn = evaluate_careful(mask, vals, pr.v0, S, d, xn, gn, g, !done, careful_ws, (size_t)gridDim.x * 64,
                            blockIdx.x * 64u + (unsigned)lane);
    } else if (pair_ok) {
      fn = evaluate_pair(seg, hc, pair_state, S, d, xn, gn, g, !done, pair_special, tick_i + 4);
    } else {
      fn = evaluate_objective<DS, MASKED4>(vtx, seg, hc, S, d, xn, gn, g, G, !done, tick_i + 4);
      if (two_wave) __syncthreads();  // the partner's hand-over barrier
    }
    __syncthreads();
    f = tick_f[0];
    alpha = tick_f[1];
    neval = tick_i[0];
    npairs = tick_i[1];
    head = tick_i[2];
    ret = tick_i[3];
    first = neval == 0;
    timed_out = t_deadline != 0ll && (long long)wall_clock64() > t_deadline;
#ifdef MRS_TG_PHASE_CLOCKS
    if (neval < 12) MRS_TG_PHASE_MARK(6 + 2 * neval);  // after evaluation #neval
#endif
    bool new_dir = false;
    // Bookkeeping in registers when every lane of the group owns at most one element of the optimiser's vectors
    // (Sb <= G: every path but the very long ones): x, g, dir live in LDS between ticks (the evaluation needs the
    // registers) and are read / written once per tick; `xi_`, `gi_` carry them from the accept step to the direction.
    double xi_ = 0.0, gi_ = 0.0;
    const bool me = g < S;
    const int Gr = (G == 64 && Sb <= 16) ? 0 : G;  // reductions of the register-resident bookkeeping (group_sum)
    if (single && !done) {
      ++neval;
      xi_ = me ? x[g] : 0.0;
      gi_ = me ? gr[g] : 0.0;
      const double xni = me ? xn[g] : 0.0, gni = me ? gn[g] : 0.0;
      if (first) {
        f = fn;
        xi_ = xni;
        gi_ = gni;
        if (budget_spent(neval)) {
          ret = budget_code(neval);
          done = true;
        } else {
          new_dir = true;
        }
      } else {
        // slope and the curvature sums of the would-be pair in one batched reduction (the sums are wasted on a
        // rejected step, which is cheaper than three more reductions on an accepted one)
        const double si = xni - xi_, yi = gni - gi_;
        double red4[4] = {gi_ * si, si * yi, si * si, yi * yi};
#ifdef MRS_TG_PHASE_CLOCKS
        if (neval == 2) MRS_TG_PHASE_MARK(23);  // tick 1: vectors read, terms formed
#endif
        group_sum_n<4>(red4, Gr);
#ifdef MRS_TG_PHASE_CLOCKS
        if (neval == 2) MRS_TG_PHASE_MARK(24);  // tick 1: the four sums
#endif
        const double slope = red4[0];
        if (fn <= f + 1e-4 * slope) {
          int stop = 0;
          if (relstop(f, fn, prm.f_rel, prm.f_abs)) {
            stop = 3;
          } else {
            const int allx = me ? (relstop(xi_, xni, prm.x_rel, prm.x_abs) ? 1 : 0) : 1;
            if (group_and(allx, G)) stop = 4;
          }
          const double sy = red4[1], ss = red4[2], yy = red4[3];
          const bool budget_out = budget_spent(neval);
          int slot = -1;
#ifdef MRS_TG_PHASE_CLOCKS
          if (neval == 2) MRS_TG_PHASE_MARK(25);  // tick 1: stopping rules
#endif
          if (!stop && !budget_out && sy > 0.0 && sy * sy > 1e-20 * (ss * yy)) {
            if (npairs == kLbfgsM) {
              slot = head;  // overwrite the oldest pair
              head = (head + 1) % kLbfgsM;
            } else {
              slot = (head + npairs) % kLbfgsM;
              ++npairs;
            }
          }
          if (slot >= 0 && me) {
            sm[slot * Sb + g] = si;
            ym[slot * Sb + g] = yi;
          }
          if (slot >= 0 && g == 0) {
            rho[slot] = 1.0 / sy;
            rho[kLbfgsM] = sy / yy;  // the scaling of the initial Hessian, taken from the newest pair
          }
#ifdef MRS_TG_PHASE_CLOCKS
          if (neval == 2) MRS_TG_PHASE_MARK(26);  // tick 1: curvature pair stored
#endif
          xi_ = xni;
          gi_ = gni;
          f = fn;
          if (stop) {
            ret = stop;
            done = true;
          } else if (budget_out) {
            ret = budget_code(neval);
            done = true;
          } else {
            new_dir = true;
          }
        } else if (budget_spent(neval)) {
          xi_ = xni;  // budget ends on a rejected trial: the last evaluated point is what the reference keeps
          ret = budget_code(neval);
          done = true;
        } else {
          alpha *= 0.5;
          if (alpha < 1e-12) {
            xi_ = xni;
            ret = 4;
            done = true;
          }
        }
      }
      if (me) {
        x[g] = xi_;
        gr[g] = gi_;
      }
    } else if (!done) {
      ++neval;
      if (first) {
        f = fn;
        for (int i = g; i < S; i += G) {
          x[i] = xn[i];
          gr[i] = gn[i];
        }
        if (budget_spent(neval)) {
          ret = budget_code(neval);
          done = true;
        } else {
          new_dir = true;
        }
      } else {
        double slope = 0.0;
        for (int i = g; i < S; i += G) slope += gr[i] * (xn[i] - x[i]);
        slope = group_sum(slope, G);
        if (fn <= f + 1e-4 * slope) {
          // accepted step: stopping rules, curvature pair, move
          int stop = 0;
          if (relstop(f, fn, prm.f_rel, prm.f_abs)) {
            stop = 3;
          } else {
            int allx = 1;
            for (int i = g; i < S; i += G)
              if (!relstop(x[i], xn[i], prm.x_rel, prm.x_abs)) allx = 0;
            if (group_and(allx, G)) stop = 4;
          }
          double sy = 0.0, ss = 0.0, yy = 0.0;
          for (int i = g; i < S; i += G) {
            const double si = xn[i] - x[i], yi = gn[i] - gr[i];
            sy += si * yi;
            ss += si * si;
            yy += yi * yi;
          }
          sy = group_sum(sy, G);
          ss = group_sum(ss, G);
          yy = group_sum(yy, G);
          const bool budget_out = budget_spent(neval);
          int slot = -1;
          if (!stop && !budget_out && sy > 0.0 && sy * sy > 1e-20 * (ss * yy)) {
            if (npairs == kLbfgsM) {
              slot = head;  // overwrite the oldest pair
              head = (head + 1) % kLbfgsM;
            } else {
              slot = (head + npairs) % kLbfgsM;
              ++npairs;
            }
          }
          for (int i = g; i < S; i += G) {
            if (slot >= 0) {
              sm[slot * Sb + i] = xn[i] - x[i];
              ym[slot * Sb + i] = gn[i] - gr[i];
            }
            x[i] = xn[i];
            gr[i] = gn[i];
          }
          if (slot >= 0 && g == 0) rho[slot] = 1.0 / sy;
          f = fn;
          if (stop) {
            ret = stop;
            done = true;
          } else if (budget_out) {
            ret = budget_code(neval);
            done = true;
          } else {
            new_dir = true;
          }
        } else if (budget_spent(neval)) {
          // budget ends on a rejected trial: the last evaluated point is what the reference keeps
          for (int i = g; i < S; i += G) x[i] = xn[i];
          ret = budget_code(neval);
          done = true;
        } else {
          alpha *= 0.5;
          if (alpha < 1e-12) {
            for (int i = g; i < S; i += G) x[i] = xn[i];
            ret = 4;
            done = true;
          }
        }
      }
    }
    first = false;
#ifdef MRS_TG_PHASE_CLOCKS
    if (neval == 2) MRS_TG_PHASE_MARK(28);  // tick 1: accept step done
#endif
    __syncthreads();
#ifdef MRS_TG_PHASE_CLOCKS
    if (neval == 2) MRS_TG_PHASE_MARK(29);
#endif

    // (2) search direction: L-BFGS two-loop recursion, projected on the lower bound
    if (single) {
      double di = (me && !new_dir) ? dir[g] : 0.0;
      if (new_dir) {
        di = -gi_;
        if (npairs > 0) {
          // the pairs and their rho are read up front (independent of the recursion), the recursion runs in registers
          double sk[kLbfgsM], yk[kLbfgsM], rk[kLbfgsM], al[kLbfgsM];
#pragma unroll
          for (int k = 0; k < kLbfgsM; ++k) {
            const int id = (head + k) % kLbfgsM;
            const bool have = k < npairs;
            sk[k] = (have && me) ? sm[id * Sb + g] : 0.0;
            yk[k] = (have && me) ? ym[id * Sb + g] : 0.0;
            rk[k] = have ? rho[id] : 0.0;
            al[k] = 0.0;
          }
#ifdef MRS_TG_PHASE_CLOCKS
          if (neval == 2) MRS_TG_PHASE_MARK(27);  // tick 1: pairs read
#endif
#pragma unroll
          for (int k = kLbfgsM - 1; k >= 0; --k)
            if (k < npairs) {
              al[k] = rk[k] * group_sum(sk[k] * di, Gr);
              di -= al[k] * yk[k];
            }
          di *= rho[kLbfgsM];  // s^T y / y^T y of the newest pair, computed when it was stored
#pragma unroll
          for (int k = 0; k < kLbfgsM; ++k)
            if (k < npairs) {
              const double beta = rk[k] * group_sum(yk[k] * di, Gr);
              di += (al[k] - beta) * sk[k];
            }
        }
#ifdef MRS_TG_PHASE_CLOCKS
        if (neval == 2) MRS_TG_PHASE_MARK(31);  // tick 1: two-loop recursion done
#endif
        if (xi_ <= kTimeLowerBound && di < 0.0) di = 0.0;
        double red3[3] = {gi_ * di, xi_ * xi_, di * di};
        group_sum_n<3>(red3, Gr);
        double gd = red3[0];
        const double nx = red3[1];
        double nd = red3[2];
        if (!(gd < 0.0)) {
          // not a descent direction: projected steepest descent, forget the curvature pairs
          double v = -gi_;
          if (xi_ <= kTimeLowerBound && v < 0.0) v = 0.0;
          di = v;
          double red2[2] = {gi_ * v, v * v};
          group_sum_n<2>(red2, Gr);
          gd = red2[0];
          nd = red2[1];
          npairs = 0;
          head = 0;
          if (!(gd < 0.0)) {
            ret = 1;
            done = true;
          }
        }
        alpha = 1.0;
        if (!done && npairs == 0) {
          // first trial step of a restart moves x by at most 10 % in norm
          const double cap = 0.1 * sqrt(nx) / sqrt(nd);
          if (cap < alpha) alpha = cap;
        }
        if (me) dir[g] = di;
      }
      // (3) next trial point
      if (!done && me) {
        if (!new_dir) xi_ = x[g];
        xn[g] = fmax(xi_ + alpha * di, kTimeLowerBound);
      }
    } else {
    if (new_dir) {
      double al[kLbfgsM];
      for (int i = g; i < S; i += G) dir[i] = -gr[i];
      if (npairs > 0) {
        for (int k = npairs - 1; k >= 0; --k) {
          const int id = (head + k) % kLbfgsM;
          const double sd = group_dot(sm + id * Sb, dir, S, g, G);
          al[k] = rho[id] * sd;
          for (int i = g; i < S; i += G) dir[i] -= al[k] * ym[id * Sb + i];
        }
        const int nw = (head + npairs - 1) % kLbfgsM;
        const double sy = group_dot(sm + nw * Sb, ym + nw * Sb, S, g, G);
        const double yy = group_dot(ym + nw * Sb, ym + nw * Sb, S, g, G);
        const double gamma = sy / yy;
        for (int i = g; i < S; i += G) dir[i] *= gamma;
        for (int k = 0; k < npairs; ++k) {
          const int id = (head + k) % kLbfgsM;
          const double yd = group_dot(ym + id * Sb, dir, S, g, G);
          const double beta = rho[id] * yd;
          for (int i = g; i < S; i += G) dir[i] += (al[k] - beta) * sm[id * Sb + i];
        }
      }
      double gd = 0.0;
      for (int i = g; i < S; i += G) {
        if (x[i] <= kTimeLowerBound && dir[i] < 0.0) dir[i] = 0.0;
        gd += gr[i] * dir[i];
      }
      gd = group_sum(gd, G);
      if (!(gd < 0.0)) {
        // not a descent direction: projected steepest descent, forget the curvature pairs
        gd = 0.0;
        for (int i = g; i < S; i += G) {
          double v = -gr[i];
          if (x[i] <= kTimeLowerBound && v < 0.0) v = 0.0;
          dir[i] = v;
          gd += gr[i] * v;
        }
        gd = group_sum(gd, G);
        npairs = 0;
        head = 0;
        if (!(gd < 0.0)) {
          ret = 1;
          done = true;
        }
      }
      alpha = 1.0;
      if (!done && npairs == 0) {
        // first trial step of a restart moves x by at most 10 % in norm
        double nx = 0.0, nd = 0.0;
        for (int i = g; i < S; i += G) {
          nx += x[i] * x[i];
          nd += dir[i] * dir[i];
        }
        nx = group_sum(nx, G);
        nd = group_sum(nd, G);
        const double cap = 0.1 * sqrt(nx) / sqrt(nd);
        if (cap < alpha) alpha = cap;
      }
    }
    // (3) next trial point
    if (!done)
      for (int i = g; i < S; i += G) xn[i] = fmax(x[i] + alpha * dir[i], kTimeLowerBound);
    }
#ifdef MRS_TG_PHASE_CLOCKS
    if (neval == 2) MRS_TG_PHASE_MARK(30);  // tick 1: direction and trial point done
#endif
    if (two_wave) {
      const bool all_done = __ballot(!done) == 0ull;
      if (lane == 0) pair_flags[0] = all_done ? 1 : 0;
    }
    __syncthreads();
#ifdef MRS_TG_PHASE_CLOCKS
    if (neval >= 1 && neval < 13) MRS_TG_PHASE_MARK(5 + 2 * neval);  // end of the tick that made evaluation #neval-1
#endif
  }
  MRS_TG_PHASE_MARK(5);

  if (active && !queued) retire();
  if (queued && busy) retire();
}

// One dimension per lane (small batches): up to two wavefronts per path, and at most 256 VGPRs so that both sit on one
// SIMD.  Four dimensions per lane (large batches): one wavefront per 64 / G paths, registers as needed.
__global__ __launch_bounds__(128, 2) void optimize_split_kernel(BatchView b, NonlinearParams prm, BinTable bins,
                                                                const uint8_t* __restrict__ mask,
                                                                const double* __restrict__ vals,
                                                                double* __restrict__ seg_times,
                                                                int32_t* __restrict__ opt_status) {
  optimize_body<4>(b, prm, bins, mask, vals, seg_times, opt_status);
}

template <bool MASKED4>
__global__ __launch_bounds__(64) void optimize_compact_kernel(BatchView b, NonlinearParams prm, BinTable bins,
                                                              const uint8_t* __restrict__ mask,
                                                              const double* __restrict__ vals, double* __restrict__ seg_times,
                                                              int32_t* __restrict__ opt_status) {
  optimize_body<1, MASKED4>(b, prm, bins, mask, vals, seg_times, opt_status);
}

__global__ void set_queue_kernel(int32_t* __restrict__ queue, int32_t first_unclaimed) { *queue = first_unclaimed; }

__global__ void set_deadline_kernel(long long* __restrict__ deadline, long long budget_ticks) {
  *deadline = (long long)wall_clock64() + budget_ticks;
}

// closes a call's list of guarded paths: [2] = their number (for mrs_tg_plan_careful_count), [0] = 0 for the next call
__global__ void careful_close_kernel(int32_t* __restrict__ careful) {
  careful[2] = careful[0];
  careful[0] = 0;
}

// every bin in one launch, as the sweeping kernels; optimize_body's own bookkeeping (one vector element per lane in
// registers where the group is as wide as the path is long)
__global__ __launch_bounds__(64, MRS_TG_LEAN_WAVES) void optimize_lean_kernel(BatchView b, NonlinearParams prm, BinTable bins,
                                                              const uint8_t* __restrict__ mask, const double* __restrict__ vals,
                                                              double* __restrict__ seg_times, int32_t* __restrict__ opt_status,
                                                              int32_t* __restrict__ fallback) {
  optimize_body<1, false, false, true>(b, prm, bins, mask, vals, seg_times, opt_status, nullptr, fallback);
}

// every path of every bin has 4 <= S <= G - 4 segments: shared half sweeps only (see optimize_body)
__global__ __launch_bounds__(64, MRS_TG_LEAN_WAVES) void optimize_lean_shared_kernel(BatchView b, NonlinearParams prm, BinTable bins,
                                                              const uint8_t* __restrict__ mask, const double* __restrict__ vals,
                                                              double* __restrict__ seg_times, int32_t* __restrict__ opt_status,
                                                              int32_t* __restrict__ fallback) {
  optimize_body<1, false, false, true, false, true>(b, prm, bins, mask, vals, seg_times, opt_status, nullptr, fallback);
}

// ... with paths of 61 .. 121 segments in the launch: two passes of the shared evaluation per wavefront (see there).  Min-snap
// launches with such paths run this instantiation too (its table is the run-time order's): the min-snap body with the loop of
// passes around it needs 528 bytes of scratch per lane and is slower than this one
__global__ __launch_bounds__(64, MRS_TG_LEAN_WAVES) void optimize_lean_shared_ends_long_kernel(BatchView b, NonlinearParams prm, BinTable bins,
                                                              const uint8_t* __restrict__ mask, const double* __restrict__ vals,
                                                              double* __restrict__ seg_times, int32_t* __restrict__ opt_status,
                                                              int32_t* __restrict__ fallback) {
  optimize_body<1, true, false, true, false, true, true>(b, prm, bins, mask, vals, seg_times, opt_status, nullptr, fallback);
}

// shared half sweeps with free slots at the end vertices (objective orders below snap; evaluate_lean_shared<true>): every bin
// in groups of at least S + 4 lanes
__global__ __launch_bounds__(64, MRS_TG_LEAN_WAVES) void optimize_lean_shared_ends_kernel(BatchView b, NonlinearParams prm, BinTable bins,
                                                              const uint8_t* __restrict__ mask, const double* __restrict__ vals,
                                                              double* __restrict__ seg_times, int32_t* __restrict__ opt_status,
                                                              int32_t* __restrict__ fallback) {
  optimize_body<1, true, false, true, false, true>(b, prm, bins, mask, vals, seg_times, opt_status, nullptr, fallback);
}

// the end vertices may leave slots free (launches whose objective order is below snap: the masked step at the two ends of
// the sweep); one wavefront per SIMD -- at two it spills 96 registers and loses to the general kernel
__global__ __launch_bounds__(64) void optimize_lean_masked_kernel(BatchView b, NonlinearParams prm, BinTable bins,
                                                                  const uint8_t* __restrict__ mask,
                                                                  const double* __restrict__ vals,
                                                                  double* __restrict__ seg_times,
                                                                  int32_t* __restrict__ opt_status,
                                                                  int32_t* __restrict__ fallback) {
  optimize_body<1, true, false, true>(b, prm, bins, mask, vals, seg_times, opt_status, nullptr, fallback);
}

#if MRS_TG_WITH_CAREFUL
// The outer loop again, from the untouched start times, for the paths the fast kernels listed (a trial point whose
// by-product cost failed the guard): one path per workgroup, every evaluation through primal_cost_lane.
__global__ __launch_bounds__(64) void optimize_careful_kernel(BatchView b, NonlinearParams prm, const uint8_t* __restrict__ mask,
                                                              const double* __restrict__ vals, double* __restrict__ seg_times,
                                                              int32_t* __restrict__ opt_status, double* __restrict__ ws) {
  BinTable none{};
  optimize_body<4, false, true>(b, prm, none, mask, vals, seg_times, opt_status, ws);
}

#endif

// ---- paths with a position-free vertex (MRS_TG_FLAG_GENERAL_PATTERNS) ------------------------------------------------
// general[0] = number of such paths, general[4 + p] = 1 where path p is one, general[4 + n_paths + k] = position q of the
// k-th (in no particular order).  One thread per position.
__global__ __launch_bounds__(256) void general_list_kernel(BatchView b, const uint8_t* __restrict__ mask,
                                                           int32_t* __restrict__ general) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= b.n_paths) return;
  const PathRef pr = path_at(b, q);
  bool any = false;
  for (int v = 0; v <= pr.S; ++v) any = any || mask[(size_t)(pr.v0 + v) * kHalf] == 0;
  general[4 + pr.p] = any ? 1 : 0;
  if (any) general[4 + b.n_paths + atomicAdd(general, 1)] = q;
}

// the next launch of optimize_general_kernel takes the next `taken` entries of the list
__global__ void general_advance_kernel(int32_t* __restrict__ general, int taken) {
  general[0] = general[0] > taken ? general[0] - taken : 0;
}

// The outer loop for the listed paths: one path per workgroup, every evaluation through the 5 x 5-block solve.
__global__ __launch_bounds__(64) void optimize_general_kernel(BatchView b, NonlinearParams prm, const uint8_t* __restrict__ mask,
                                                              const double* __restrict__ vals, double* __restrict__ seg_times,
                                                              int32_t* __restrict__ opt_status, double* __restrict__ ws,
                                                              const double* __restrict__ start_times) {
  BinTable none{};
  optimize_body<4, false, true, false, true>(b, prm, none, mask, vals, seg_times, opt_status, ws, nullptr, start_times);
}

// per-segment maxima, one (k, group) per blockIdx.y: maxima[seg * 9 + 3 (k-1) + group]
__global__ __launch_bounds__(64, MRS_TG_MAXIMA_WAVES) void segment_maxima9_kernel(int n_segments, const double* __restrict__ coeffs,
                                                             const double* __restrict__ seg_times,
                                                             double* __restrict__ maxima) {
  // One lane per (segment, which).  A quad of lanes per polynomial, each a quarter of the grid (max_mag2's PARTS = 4), was
  // measured: 12.8 -> 11.1 us at 1024 x 10, 39 -> 64 us at 8192 x 10 (every lane repeats the polynomial's set-up) -- not kept.
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= n_segments) return;
  const int which = blockIdx.y;
  maxima[(size_t)s * 9 + which] = segment_maximum(coeffs + (size_t)s * kD * kN, seg_times[s], which);
}

// The maxima of a pipeline, where they only feed the feasibility scaling s_i = max(1, v / v_max, sqrt(a / a_max),
// cbrt(j / j_max)) (trajectory.cpp:625-642): an entry that cannot exceed its limit does not move s_i, whatever its exact value.
// A workgroup takes kMsSegs segments.  Phase A, one lane per (segment, dimension): Bernstein bounds on |q^(k)|, k = 1..3
// (bernstein_bound), combined over the dimensions of a group inside the lane quad; an entry whose bound (with a 1e-9 margin)
// is below its limit is written as that bound -- below the limit like the exact maximum, so the scaling is the same bits --
// every other one is queued in LDS.  Phase B: the queued searches, densely, one lane each: the horizontal group first (two
// dimensions per search), then the single dimensions, the derivative order as data (segment_maximum_any: the numbers of
// segment_maxima9_kernel).  On the bench's batches 81 % of the 9 entries per segment are settled in phase A (all of the jerk,
// most of the acceleration and of the vertical and heading velocity); 65536 x 10: 303 -> us.
constexpr int kMsSegs = 128;
constexpr int kMsThreads = 256;

__global__ __launch_bounds__(kMsThreads) void segment_maxima_scaling_kernel(BatchView b, const double* __restrict__ coeffs,
                                                                            const double* __restrict__ seg_times,
                                                                            const double* __restrict__ limits,
                                                                            double* __restrict__ maxima) {
  __shared__ unsigned short list_h[kMsSegs * 3], list_s[kMsSegs * 6];
  __shared__ int n_h, n_s;
  const int tid = threadIdx.x;
  const int seg0 = blockIdx.x * kMsSegs;
  if (tid == 0) {
    n_h = 0;
    n_s = 0;
  }
  __syncthreads();
  // ---- phase A
  for (int e = tid; e < kMsSegs * kD; e += kMsThreads) {
    const int ls = e >> 2, dim = e & 3, s = seg0 + ls;
    const bool live = s < b.n_segments;
    const int sc = live ? s : b.n_segments - 1;
    int p;
    if (b.uniform_S > 0) {
      p = sc / b.uniform_S;
    } else {
      int lo = 0, hi = b.n_paths;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (b.seg_offsets[mid] <= sc) lo = mid;
        else hi = mid;
      }
      p = lo;
    }
    const double T = seg_times[sc];
    const double* c = coeffs + ((size_t)sc * kD + dim) * kN;
    double cb[kN], tp = 1.0;
#pragma unroll
    for (int j = 0; j < kN; ++j) {
      cb[j] = c[j] * tp;
      tp *= T;
    }
    const double ti = 1.0 / T;
    double bd[3] = {bernstein_bound<1>(cb), bernstein_bound<2>(cb), bernstein_bound<3>(cb)};
    double sck = ti;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      // the bounds of the four dimensions, to every lane of the quad
      const double bx = dpp_move<0x00>(bd[k]), by = dpp_move<0x55>(bd[k]), bz = dpp_move<0xAA>(bd[k]), bh = dpp_move<0xFF>(bd[k]);
      // lane 0 of the quad settles the horizontal group, lane 2 the vertical one, lane 3 the heading
      const int grp = dim == 0 ? 0 : dim - 1;
      const double bound = (dim == 0 ? sqrt(bx * bx + by * by) : (dim == 2 ? bz : bh)) * sck * (1.0 + 1.0e-9);
      if (live && dim != 1) {
        const double lim = limits[(size_t)p * 9 + k * 3 + grp];
        if (bound <= lim) {
          maxima[(size_t)s * 9 + k * 3 + grp] = bound;
        } else {  // (also a bound that is not a number: the search then reports what the separate kernel reports)
          const unsigned short entry = (unsigned short)(ls | (k << 7) | (grp << 9));
          if (grp == 0) list_h[atomicAdd(&n_h, 1)] = entry;
          else list_s[atomicAdd(&n_s, 1)] = entry;
        }
      }
      sck *= ti;
    }
  }
  __syncthreads();
  // ---- phase B: wavefronts of horizontal searches, then wavefronts of single-dimension searches
  const int nh = n_h, nh_pad = (nh + 63) & ~63, total = nh_pad + n_s;
  for (int t = tid; t < total; t += kMsThreads) {
    if (t < nh_pad) {
      if (t < nh) {
        const int entry = list_h[t], ls = entry & 127, k = (entry >> 7) & 3, s = seg0 + ls;
        maxima[(size_t)s * 9 + k * 3] = segment_maximum_any<2, 1>(coeffs + (size_t)s * kD * kN, seg_times[s], k + 1, 0, 0);
      }
    } else {
      const int entry = list_s[t - nh_pad], ls = entry & 127, k = (entry >> 7) & 3, grp = entry >> 9, s = seg0 + ls;
      maxima[(size_t)s * 9 + k * 3 + grp] = segment_maximum_any<1, 1>(coeffs + (size_t)s * kD * kN, seg_times[s], k + 1, 1 + grp, 0);
    }
  }
}

// scaleSegmentTimesToMeetConstraints' per-segment step (trajectory.cpp:610-658): T <- T * max(1, v, sqrt a, cbrt j).
// Paths whose start the optimiser rejected are left alone (they never reach this step in the reference).
__global__ __launch_bounds__(256) void apply_scaling_kernel(BatchView b, const double* __restrict__ maxima,
                                                            const double* __restrict__ limits,
                                                            const int32_t* __restrict__ opt_status,
                                                            double* __restrict__ seg_times) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= b.n_segments) return;
  int p;
  if (b.uniform_S > 0) {
    p = idx / b.uniform_S;  // one division instead of log2(P) dependent loads of the search below (5 -> 3 us at 1024 x 10)
  } else {
    int lo = 0, hi = b.n_paths;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (b.seg_offsets[mid] <= idx) lo = mid;
      else hi = mid;
    }
    p = lo;
  }
  if (opt_status[p] == -2) return;
  seg_times[idx] = seg_times[idx] * violation_scaling(maxima + (size_t)idx * 9, limits + (size_t)p * 9);
}

// The runaway test of the final solve for the pipelines whose scaling is its own launch (the rows kernel does it in its
// tail): a path whose scaled total time exceeds MRS_TG_RUNAWAY_TIME_FACTOR times the total it started from leaves the
// pipeline with ROUNDOFF_LIMITED (-4) instead of the outer loop's stopping reason (include/mrs_tg.h).
__global__ __launch_bounds__(256) void runaway_kernel(BatchView b, const double* __restrict__ seg_times,
                                                      const double* __restrict__ sum_t0, int32_t* __restrict__ opt_status) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= b.n_paths) return;
  const PathRef pr = path_at(b, q);
  if (opt_status[pr.p] <= 0) return;
  double t = 0.0;
  for (int i = 0; i < pr.S; ++i) t += seg_times[pr.s0 + i];
  if (t > MRS_TG_RUNAWAY_TIME_FACTOR * sum_t0[pr.p]) opt_status[pr.p] = MRS_TG_STATUS_ROUNDOFF_LIMITED;
}

// J_d and the forward-difference gradient at the given times (parity-test building block)
template <int DS>
__global__ __launch_bounds__(64) void cost_gradient_kernel(BatchView b, int d, int G, int q_begin, int q_count, int Sb,
                                                           const uint8_t* __restrict__ mask,
                                                           const double* __restrict__ vals,
                                                           const double* __restrict__ seg_times,
                                                           double* __restrict__ cost, double* __restrict__ grad,
                                                           const int32_t* __restrict__ only_flagged) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  const int g = lane & (G - 1);
  const int g_shift = __builtin_ctz((unsigned)G);  // G is a power of two
  const int grp = lane >> g_shift;
  const int qi = blockIdx.x * (64 >> g_shift) + grp;
  bool active = qi < q_count;
  const PathRef pr = path_at(b, q_begin + (active ? qi : 0));
  if (only_flagged) active = active && only_flagged[q_begin + (qi < q_count ? qi : 0)] != 0;
  double* hc = lds;
  stage_block_constants(d, hc, lane, 64);
  constexpr bool kExtras = DS == 4;
  double* x = lds + kBlockConsts + (size_t)grp * gradient_lds_doubles(Sb, kExtras);
  double* gr = x + Sb;
  double* vtx = gr + Sb;
  double* seg = vtx + (size_t)(Sb + 1) * kVtxLds + (kExtras ? kStartExtra : 0);
  if (active) {
    for (int i = g; i < pr.S; i += G) x[i] = seg_times[pr.s0 + i];
    stage_vertices(mask, vals, pr.v0, pr.S, vtx, g, G);
  }
  __syncthreads();
  if (active) stage_segments(vtx, pr.S, d, seg, g, G, kExtras);
  __syncthreads();
  const double J = evaluate_objective<DS>(vtx, seg, hc, pr.S, d, x, gr, g, G, active);
  __syncthreads();
  if (active) {
    for (int i = g; i < pr.S; i += G) grad[pr.s0 + i] = gr[i];
    if (g == 0) cost[pr.p] = J;
  }
}

// ---------------------------------------------------------------------------------------------
// host side

// split the four dimensions over lanes while the batch is too small to fill the machine otherwise
static int dim_split_for(int n_paths, int max_S) {
  if (const char* e = std::getenv("MRS_TG_DIM_SPLIT_MAX_PATHS")) return n_paths <= std::atoi(e) ? 4 : 1;  // tuning knob
  // One lane per dimension pays while a path's (S + 1) x 4 lanes fit ONE wavefront: from 16 segments on the split kernel
  // walks a path in several passes and the lean kernel's lane groups are 2-3x faster at every batch size (whole pipeline,
  // scripts/measure_configs.py uniform<P>x<S>, profiles/round5_dim_split_crossover.txt: 256 x 16 0.378 vs 0.168 ms,
  // 2048 x 30 1.290 vs 0.429 ms, 1024 ragged 3..30 0.681 vs 0.240 ms).
  if (max_S > 15) return 1;
  // measured cross-over against the lean kernel of large batches (scripts/ps_step.py with MRS_TG_DIM_SPLIT_MAX_PATHS=0):
  // outer-loop kernel 2048 paths 122 vs 143 us, 3072 paths 160 vs 150 us (10 segments); 13 segments: 1024 paths 0.152 vs
  // 0.168 ms, 2048 paths 0.220 vs 0.192 ms (whole pipeline)
  return n_paths <= (max_S > 12 ? 1536 : 2560) ? 4 : 1;
}

static int group_for(int S, int ds) {
  int G = 4;
  while (G < (S + 1) * ds && G < 64) G <<= 1;
  return G;
}

// lanes per path when the long paths get the S + 4 lanes of the shared half sweeps: 13-15 and 29-30 segments move to the next
// group width.  (5-7 segments stay in groups of 8 and their one-sided sweeps: six steps against four do not pay for half
// the paths per wavefront -- 8192 x 6: 0.190 ms narrow, 0.197 ms wide; 65536 x 6: 0.861 vs 1.015 ms.)  These bins decide
// WHETHER a launch is small enough for wide groups (wide_blocks) and are what MRS_TG_LEAN_WIDE_ALL=0 launches; by default
// such a launch uses the bins of group_for_ends below (every path of two or more segments in S + 4 lanes: launch_nonlinear)
static int group_for_wide(int S) {
  int G = group_for(S, 1);
  if (S >= 13 && G < S + 4 && G < 64) G <<= 1;
  return G;
}

constexpr int kLeanTwoPassMaxS = 121;  // evaluate_lean_shared: 3 publishers + 61 other virtual lanes per pass, two passes
// lanes per path when EVERY path of kEndsMinSegments or more segments gets its S + 4 lanes (0: no group is wide enough).
// Two segments are enough for the shared half sweeps (one step per half); handing the 2- and 3-segment paths of a ragged
// batch to the sweeping kernel BEHIND the launch instead cost 8192 ragged paths 0.67 instead of 0.58 ms (d = 2).
// MRS_TG_ENDS_MIN_SEGMENTS: tuning / test knob, read once per process
static const int kEndsMinSegments = [] {
  const char* e = std::getenv("MRS_TG_ENDS_MIN_SEGMENTS");
  return e ? std::max(2, std::atoi(e)) : 2;
}();
static int group_for_ends(int S) {
  if (S < kEndsMinSegments) return group_for(S, 1);
  if (S + 4 > 64) return S <= kLeanTwoPassMaxS ? 64 : 0;  // a wavefront per path, two passes of the shared evaluation
  int G = 8;
  while (G < S + 4) G <<= 1;
  return G;
}

template <class GroupFor>
static void build_bins(std::vector<NonlinearBin>& bins, const std::vector<int32_t>& so, const std::vector<int32_t>& order,
                       GroupFor group_of) {
  bins.clear();
  const int P = (int)order.size();
  int q = 0;
  while (q < P) {
    const int p = order[q];
    const int S = so[p + 1] - so[p];
    NonlinearBin bin;
    bin.group = group_of(S);
    bin.q_begin = q;
    bin.max_S = S;  // sorted longest first: the first path of a bin is its longest
    int e = q;
    while (e < P && group_of(so[order[e] + 1] - so[order[e]]) == bin.group) ++e;
    bin.q_count = e - q;
    bin.min_S = so[order[e - 1] + 1] - so[order[e - 1]];  // ... and the last one its shortest
    bins.push_back(bin);
    q = e;
  }
}

int nonlinear_plan_build(NonlinearPlan& nl, const std::vector<int32_t>& so, const std::vector<int32_t>& order) {
  const int P = (int)order.size();
  nl.dim_split = dim_split_for(P, P > 0 ? so[order[0] + 1] - so[order[0]] : 0);  // (sorted longest first)
  const int ds = nl.dim_split;
  const int max_S = P > 0 ? so[order[0] + 1] - so[order[0]] : 0;
  build_bins(nl.bins, so, order, [ds](int S) { return group_for(S, ds); });
  // The lane-group layouts as well where a CALL may prefer them to the plan's dimension split (launch_nonlinear: 13-15
  // segments under an objective order below snap or with constrained slots / moving starts, whose paths the split kernel
  // hands to its general step): the plan is built before the options are known
  nl.regroup_possible = ds == 4 && max_S >= 13 && max_S <= 15;
  const bool groups = ds == 1 || nl.regroup_possible;
  nl.bins1.clear();
  if (nl.regroup_possible) build_bins(nl.bins1, so, order, [](int S) { return group_for(S, 1); });
  nl.wide_bins.clear();
  nl.wide_blocks = 0;
  if (groups) {
    build_bins(nl.wide_bins, so, order, [](int S) { return group_for_wide(S); });
    for (const NonlinearBin& bin : nl.wide_bins) nl.wide_blocks += (int)cdiv_u(bin.q_count, 64 / bin.group);
  }
  nl.ends_bins.clear();
  if (groups && P > 0 && group_for_ends(max_S) != 0)
    build_bins(nl.ends_bins, so, order, [](int S) { return group_for_ends(S); });
  return 0;
}

void nonlinear_plan_free(NonlinearPlan& nl) {
  if (nl.d_ws) (void)mrs_tg::pool_free(nl.d_ws);
  if (nl.d_opt_status) (void)mrs_tg::pool_free(nl.d_opt_status);
  if (nl.d_maxima) (void)mrs_tg::pool_free(nl.d_maxima);
  if (nl.d_sum_t0) (void)mrs_tg::pool_free(nl.d_sum_t0);
  nl.d_sum_t0 = nullptr;
  if (nl.d_queue) (void)mrs_tg::pool_free(nl.d_queue);
  nl.d_queue = nullptr;
  if (nl.d_general) (void)mrs_tg::pool_free(nl.d_general);
  nl.d_general = nullptr;
  if (nl.d_general_ws) (void)mrs_tg::pool_free(nl.d_general_ws);
  nl.d_general_ws = nullptr;
  if (nl.d_general_solve_ws) (void)mrs_tg::pool_free(nl.d_general_solve_ws);
  nl.d_general_solve_ws = nullptr;
  if (nl.d_general_t0) (void)mrs_tg::pool_free(nl.d_general_t0);
  nl.d_general_t0 = nullptr;
  if (nl.d_careful) (void)mrs_tg::pool_free(nl.d_careful);
  nl.d_careful = nullptr;
  if (nl.d_fallback) (void)mrs_tg::pool_free(nl.d_fallback);
  nl.d_fallback = nullptr;
  if (nl.d_careful_ws) (void)mrs_tg::pool_free(nl.d_careful_ws);
  nl.d_careful_ws = nullptr;
  nl.careful_ws_doubles = 0;
  if (nl.d_dfo_vec) (void)mrs_tg::pool_free(nl.d_dfo_vec);
  if (nl.d_dfo_f) (void)mrs_tg::pool_free(nl.d_dfo_f);
  if (nl.d_dfo_state) (void)mrs_tg::pool_free(nl.d_dfo_state);
  if (nl.d_dfo_fidx) (void)mrs_tg::pool_free(nl.d_dfo_fidx);
  if (nl.d_dfo_segcost) (void)mrs_tg::pool_free(nl.d_dfo_segcost);
  if (nl.d_dfo_seg_path) (void)mrs_tg::pool_free(nl.d_dfo_seg_path);
  if (nl.d_dfo_deadline) (void)mrs_tg::pool_free(nl.d_dfo_deadline);
  nl.d_dfo_deadline = nullptr;
  nl.d_dfo_fidx = nullptr;
  nl.d_dfo_segcost = nullptr;
  nl.d_dfo_seg_path = nullptr;
  nl.d_dfo_vec = nullptr;
  nl.d_dfo_f = nullptr;
  nl.d_dfo_state = nullptr;
  nl.d_ws = nullptr;
  nl.d_opt_status = nullptr;
  nl.d_maxima = nullptr;
  nl.ws_doubles = 0;
}


constexpr int kCarefulCap = 1024;  // paths per call that can be re-run with primal costs; further ones keep the fast result

bool careful_rerun_built() { return MRS_TG_WITH_CAREFUL != 0; }

hipError_t nonlinear_ensure_buffers(NonlinearPlan& nl, const BatchView& b) {
  hipError_t e;
  const size_t need = linear_workspace_doubles(b);
  if (nl.ws_doubles < need) {
    if (nl.d_ws) (void)mrs_tg::pool_free(nl.d_ws);
    nl.d_ws = nullptr;
    nl.ws_doubles = 0;
    if ((e = mrs_tg::pool_alloc(&nl.d_ws, need * sizeof(double))) != hipSuccess) return e;
    nl.ws_doubles = need;
  }
  if (!nl.d_opt_status && (e = mrs_tg::pool_alloc(&nl.d_opt_status, sizeof(int32_t) * (size_t)(b.n_paths > 0 ? b.n_paths : 1))) != hipSuccess)
    return e;
  if (!nl.d_maxima && (e = mrs_tg::pool_alloc(&nl.d_maxima, sizeof(double) * 9 * (size_t)(b.n_segments > 0 ? b.n_segments : 1))) != hipSuccess)
    return e;
  if (!nl.d_sum_t0 && (e = mrs_tg::pool_alloc(&nl.d_sum_t0, sizeof(double) * (size_t)(b.n_paths > 0 ? b.n_paths : 1))) != hipSuccess)
    return e;
  if (!nl.d_careful) {
    if ((e = mrs_tg::pool_alloc(&nl.d_careful, sizeof(int32_t) * (4 + kCarefulCap))) != hipSuccess) return e;
    if ((e = hipMemset(nl.d_careful, 0, sizeof(int32_t) * 4)) != hipSuccess) return e;  // once per plan: every call leaves the counter at zero
  }
  return hipSuccess;
}

// Lean sweeps for plain paths (optimize_lean_kernel): two wavefronts per SIMD where the general sweeping kernel has one.
// Objective orders below snap leave slots free at the end vertices of a rest-to-rest path: optimize_lean_masked_kernel takes
// those (masked step at the two ends of the sweep), at one wavefront per SIMD -- 256 + 74 registers; forced to two it spills
// 96 and loses to the general kernel (8192 x 10 random-walk paths at d = 2: 429 vs 365 us; at one wavefront 332 us).
static bool lean_applies(int dim_split) {
  if (const char* e = std::getenv("MRS_TG_LEAN")) return std::atoi(e) != 0;
  return dim_split == 1;
}

static hipError_t ensure_fallback(NonlinearPlan& nl, const BatchView& b) {
  if (nl.d_fallback) return hipSuccess;
  return mrs_tg::pool_alloc(&nl.d_fallback, sizeof(int32_t) * (size_t)(b.n_paths > 0 ? b.n_paths : 1));
}

// Buffers of the pipelines' route for paths with a position-free vertex, and the list / flags of this call's batch
// (general_list_kernel).  `outer_loop`: also the factor store of optimize_general_kernel's lanes and the copy of the start
// times; returns the number of paths one launch of that kernel takes in *cap.
hipError_t nonlinear_prepare_general(NonlinearPlan& nl, const BatchView& b, const uint8_t* mask, const double* seg_times,
                                  bool outer_loop, int* cap, hipStream_t stream) {
  hipError_t e;
  const size_t P = (size_t)b.n_paths;
  if (!nl.d_general && (e = mrs_tg::pool_alloc(&nl.d_general, sizeof(int32_t) * (4 + 2 * P))) != hipSuccess) return e;
  if (!nl.d_general_solve_ws &&
      (e = mrs_tg::pool_alloc(&nl.d_general_solve_ws, sizeof(double) * general_workspace_doubles(b))) != hipSuccess)
    return e;
  if (outer_loop) {
    // as many paths per launch as a 2 GB factor store holds (64 lanes x S vertices x kGenWs doubles each)
    const size_t per_path = (size_t)64 * (size_t)b.max_segments * kGenWs;
    const size_t n = std::min<size_t>(P, std::max<size_t>((size_t)16, ((size_t)1 << 28) / per_path));
    if (!nl.d_general_ws && (e = mrs_tg::pool_alloc(&nl.d_general_ws, sizeof(double) * per_path * n)) != hipSuccess) return e;
    if (!nl.d_general_t0 &&
        (e = mrs_tg::pool_alloc(&nl.d_general_t0, sizeof(double) * (size_t)std::max(b.n_segments, 1))) != hipSuccess)
      return e;
    if (!dry_run() &&
        (e = hipMemcpyAsync(nl.d_general_t0, seg_times, sizeof(double) * (size_t)b.n_segments, hipMemcpyDeviceToDevice, stream)) !=
            hipSuccess)
      return e;
    if (cap) *cap = (int)n;
  }
  if ((e = hipMemsetAsync(nl.d_general, 0, sizeof(int32_t) * 4, stream)) != hipSuccess) return e;
  MRS_TG_LAUNCH(general_list_kernel, dim3(cdiv_u(b.n_paths, 256)), dim3(256), 0, stream, b, mask, nl.d_general);
  return hipGetLastError();
}

hipError_t launch_nonlinear(NonlinearPlan& nl, const BatchView& b, const NonlinearParams& prm_in, const uint8_t* mask,
                            const double* vals, const double* limits, double* seg_times, double* coeffs,
                            int32_t* status, double* cost, hipStream_t stream, double sampling_dt, int sample_capacity,
                            int32_t* n_samples, double* samples, bool* sampled_out, bool general) {
  if (sampled_out) *sampled_out = false;
  if (b.n_paths == 0) return hipSuccess;
  hipError_t e = nonlinear_ensure_buffers(nl, b);
  if (e != hipSuccess) return e;
  // the start point is the estimate (prm_in.estimate_wp): computed by the outer-loop kernels themselves (start_time), by a
  // launch of its own in front of the kernels that read their start from a copy
  // (every outer-loop kernel but the one for position-free paths and the careful re-run, which start from a copy of the times)
  const bool estimate_in_kernel = prm_in.estimate_wp != nullptr && !general && prm_in.careful_cap == 0;
  if (prm_in.estimate_wp && !estimate_in_kernel &&
      (e = launch_estimate_times(b, prm_in.estimate_wp, prm_in.estimate_limits, seg_times, stream)) != hipSuccess)
    return e;
  // paths with a position-free vertex (the caller says there may be some): flagged and listed, their start times kept aside
  int general_cap = 0;
  if (general && (e = nonlinear_prepare_general(nl, b, mask, seg_times, true, &general_cap, stream)) != hipSuccess) return e;
  const int32_t* general_flag = general ? nl.d_general + 4 : nullptr;
  // 1. outer loop: every bin in one launch
  NonlinearParams prm = prm_in;
  static const int lean_shared = [] {
    // tuning / test knob, read once per process; 0: one-sided lean sweeps only, 1: shared half sweeps where a whole batch
    // takes them (optimize_lean_shared_kernel), 2 (default): also wave by wave inside the mixed kernel (ragged batches)
    const char* e = std::getenv("MRS_TG_LEAN_SHARED");
    return e == nullptr ? 2 : std::atoi(e);
  }();
  prm.lean_shared = lean_shared;
  prm.ends_min_segments = kEndsMinSegments;
  if (!estimate_in_kernel) prm.estimate_wp = prm.estimate_limits = nullptr;
  prm.sum_t0 = nl.d_sum_t0;
  prm.deadline = nullptr;
  if (prm_in.time_budget_ticks > 0) {
    if (!nl.d_dfo_deadline && (e = mrs_tg::pool_alloc(&nl.d_dfo_deadline, sizeof(long long))) != hipSuccess) return e;
    MRS_TG_LAUNCH(set_deadline_kernel, dim3(1), dim3(1), 0, stream, nl.d_dfo_deadline, prm_in.time_budget_ticks);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    prm.deadline = nl.d_dfo_deadline;
  }
  const bool careful = prm_in.careful_cap != 0;  // the caller asked for the careful re-run (MRS_TG_FLAG_CAREFUL_COST)
  int careful_cap = 0;
  if (careful) {
    // as many paths per call as a 2 GB factor store holds (64 lanes x S vertices x 30 doubles each), at most kCarefulCap
    const size_t per_path = (size_t)64 * (size_t)b.max_segments * ws_per_vertex<1>();
    careful_cap = (int)std::min<size_t>((size_t)kCarefulCap, std::max<size_t>((size_t)16, ((size_t)1 << 28) / per_path));
    careful_cap = std::min(careful_cap, b.n_paths);
    const size_t need = per_path * (size_t)careful_cap;
    if (nl.careful_ws_doubles < need) {
      if (nl.d_careful_ws) (void)mrs_tg::pool_free(nl.d_careful_ws);
      nl.d_careful_ws = nullptr;
      nl.careful_ws_doubles = 0;
      if ((e = mrs_tg::pool_alloc(&nl.d_careful_ws, need * sizeof(double))) != hipSuccess) return e;
      nl.careful_ws_doubles = need;
    }
    prm.careful_count = nl.d_careful;
    prm.careful_list = nl.d_careful + 4;
    prm.careful_cap = careful_cap;
  } else {
    prm.careful_count = nullptr;
    prm.careful_list = nullptr;
    prm.careful_cap = 0;
  }
  // per-dispatch timing (kernel family 2): one pair of events spans ALL outer-loop launches of the call -- the start event
  // rides on the first, the stop event on the last (the lean kernel, the general kernel for the paths it flagged, the
  // careful re-run)
  const KernelTimer kt = take_kernel_timer();
  // 1a. the lean kernel takes every plain path and flags the others for the sweeping kernel below
  // The split of the four dimensions over lanes is the PLAN's choice (batch size and longest path); this CALL goes back to
  // lane groups where the split kernel would hand its paths to the general step anyway and the groups are 10-25 % faster:
  // 13-15 segments (<= 1536 paths) under an objective order below snap, or when the caller says that vertices hold
  // constrained slots (stop_at waypoints: MRS_TG_FLAG_CONSTRAINED_SLOTS) or mrs_tg_solve_batch has seen a moving start in
  // its host copy of the values.  MRS_TG_REGROUP=0 switches it off.
  static const bool regroup_allowed = [] {
    const char* e = std::getenv("MRS_TG_REGROUP");
    return e == nullptr || std::atoi(e) != 0;
  }();
  const bool regroup = regroup_allowed && nl.regroup_possible &&
                       (prm_in.derivative < 4 || constrained_slots_hint() || moving_starts_hint());
  const int dim_split = regroup ? 1 : nl.dim_split;
  const std::vector<NonlinearBin>& plan_bins = regroup ? nl.bins1 : nl.bins;
  bool lean = lean_applies(dim_split) && (int)plan_bins.size() <= 5;
  auto plain_lds = [&](const NonlinearBin& bin) {
    return ((size_t)(64 / bin.group) * lean_group_doubles(bin.max_S) + 2 * kPsTable) * sizeof(double);
  };
  // Which lane groups.  The shared half sweeps halve the steps of an evaluation but need S + 4 lanes, the next group width
  // for 13-15 and 29-30 segments: half the paths per wavefront for (S/2 + 1)/S of the steps -- more instructions per path,
  // so a launch of many residency rounds (a saturated device: the kernel is bound by its instruction count) keeps the
  // narrow groups and their one-sided sweeps.  A launch of a few residency rounds lasts about as long as its slowest
  // wavefronts -- ten evaluations of the longest paths -- and there the wide groups win: whole pipeline, wide vs narrow,
  // 8192 ragged 0.57 vs 0.65 ms, 8192 x 14 0.395 vs 0.425, 16384 x 14 0.644 vs 0.674, 32768 x 30 3.33 vs 3.51,
  // 32768 ragged 1.87 vs 1.89; past that narrow: 65536 x 14 2.03 vs 1.97 ms (profiles/round5_wide_groups_ab.txt).
  // (MRS_TG_LEAN_WIDE=0 / 1 forces.)
  // (per device: a process may drive devices of different sizes or partitions, and the first call's figure is not theirs)
  int resident_waves = 256 * 4 * MRS_TG_LEAN_WAVES;
  {
    static std::mutex mu;
    static std::map<int, int> cus_of;  // device ordinal -> compute units
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
      std::lock_guard<std::mutex> lock(mu);
      auto it = cus_of.find(dev);
      if (it == cus_of.end()) {
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        it = cus_of.emplace(dev, cus).first;
      }
      resident_waves = it->second * 4 * MRS_TG_LEAN_WAVES;
    }
  }
  static const int wide_forced = [] {
    const char* e = std::getenv("MRS_TG_LEAN_WIDE");
    return e ? (std::atoi(e) != 0 ? 1 : 0) : -1;
  }();
  const bool lean_masked_order = prm_in.derivative < 4;
  bool wide = !nl.wide_bins.empty() && (int)nl.wide_bins.size() <= 5 && !lean_masked_order && prm.lean_shared != 0 &&
              (wide_forced >= 0 ? wide_forced == 1 : nl.wide_blocks <= 8 * resident_waves);
  if (wide)
    for (const NonlinearBin& bin : nl.wide_bins)
      if (plain_lds(bin) > 160 * 1024) wide = false;
  // objective orders below snap: the shared half sweeps with free end slots, every path in a group of S + 4 lanes at least
  // (MRS_TG_LEAN_SHARED=0: the one-sided masked sweeps of optimize_lean_masked_kernel, as until round 5)
  // (a min-snap launch whose caller says that interior vertices may hold constrained slots -- stop_at waypoints -- as well:
  // MRS_TG_FLAG_CONSTRAINED_SLOTS)
  bool ends_shared = (lean_masked_order || constrained_slots_hint()) && prm.lean_shared != 0 && !nl.ends_bins.empty() &&
                     (int)nl.ends_bins.size() <= 5;
  if (ends_shared)
    for (const NonlinearBin& bin : nl.ends_bins)
      if (plain_lds(bin) > 160 * 1024) ends_shared = false;
  // min-snap launches of a few residency rounds: EVERY path of two or more segments in a group of S + 4 lanes (the bins of the
  // free-end kernel), so that the kernel with only the shared half sweeps runs -- the one that takes moving starts; 5-7
  // segments then pay 3-4 % for their wider groups when they start at rest (MRS_TG_LEAN_WIDE_ALL=0: only 13-15 and 29-30
  // segments move up and a ragged batch runs the mixed kernel, whose one-sided sweeps leave moving starts to the sweeping kernel)
  static const bool wide_all = [] {
    const char* e = std::getenv("MRS_TG_LEAN_WIDE_ALL");
    return e == nullptr || std::atoi(e) != 0;
  }();
  const bool wide_shared_all = wide && wide_all && !nl.ends_bins.empty() && (int)nl.ends_bins.size() <= 5;
  const std::vector<NonlinearBin>& lean_bins = (ends_shared || wide_shared_all) ? nl.ends_bins : wide ? nl.wide_bins : plan_bins;
  if (lean)
    for (const NonlinearBin& bin : lean_bins)
      if (plain_lds(bin) > 160 * 1024) lean = false;
  if (lean) {
    if ((e = ensure_fallback(nl, b)) != hipSuccess) return e;
    BinTable bt{};
    bt.n = (int)lean_bins.size();
    size_t plds = 0;
    int blocks = 0;
    for (int i = 0; i < bt.n; ++i) {
      const NonlinearBin& bin = lean_bins[i];
      bt.group[i] = bin.group;
      bt.q_begin[i] = bin.q_begin;
      bt.q_count[i] = bin.q_count;
      bt.max_S[i] = bin.max_S;
      bt.block_begin[i] = blocks;
      blocks += (int)cdiv_u(bin.q_count, 64 / bin.group);
      plds = std::max(plds, plain_lds(bin));
    }
    // A uniform batch of more wavefronts than the device holds at once (two per SIMD): launch what is resident and let a
    // lane group whose path has stopped claim the next one (optimize_body, `queued`), instead of eight rounds of wavefronts
    // that each last as long as the slowest of their four paths.  65536 x 10: 780 -> us.
    prm.queue_next = nullptr;
    static const int resident_blocks = [] {
      if (const char* e = std::getenv("MRS_TG_LEAN_RESIDENT_BLOCKS")) return std::atoi(e);  // tuning knob; 0 = no queue
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      return cus * 4 * MRS_TG_LEAN_WAVES;
    }();
    if (bt.n == 1 && resident_blocks > 0 && blocks > resident_blocks) {
      if (!nl.d_queue && (e = mrs_tg::pool_alloc(&nl.d_queue, sizeof(int32_t) * 4)) != hipSuccess) return e;
      blocks = resident_blocks;
      MRS_TG_LAUNCH(set_queue_kernel, dim3(1), dim3(1), 0, stream, nl.d_queue, blocks * (64 / bt.group[0]));
      if ((e = hipGetLastError()) != hipSuccess) return e;
      prm.queue_next = nl.d_queue;
    }
    const bool lean_masked = prm.derivative < 4;  // rest-to-rest paths end on vertices with free slots
    // shared half sweeps (evaluate_lean_shared) where every path of every bin has its S + 4 lanes
    // (with a bin of paths shorter than four segments beside them the kernel that has both evaluations compiled in runs, the
    // short paths in wavefronts of their own next to the others: handing them to the compact kernel BEHIND the launch cost
    // 66 us on 8192 ragged paths, profiles/round5_wide_groups_ab.txt)
    bool lean_shared_only = prm.lean_shared != 0 && !lean_masked;
    for (const NonlinearBin& bin : lean_bins)  // (a bin of one-segment paths: the kernel leaves them to the sweeping kernel behind it)
      if (bin.max_S >= 2 && (bin.min_S < 2 || (bin.max_S + 4 > bin.group && !(bin.group == 64 && bin.max_S <= kLeanTwoPassMaxS))))
        lean_shared_only = false;
    bool long_paths = false;  // a bin whose paths have more half sweeps than a wavefront has lanes: the two-pass instantiations
    for (const NonlinearBin& bin : lean_bins)
      if (bin.max_S + 4 > bin.group && bin.group == 64) long_paths = true;
    const bool ends_long = long_paths && (ends_shared || lean_shared_only);
    if (plds > 64 * 1024 &&
        (e = hipFuncSetAttribute(ends_long ? (const void*)optimize_lean_shared_ends_long_kernel
                                 : ends_shared ? (const void*)optimize_lean_shared_ends_kernel
                                 : lean_masked ? (const void*)optimize_lean_masked_kernel
                                 : lean_shared_only ? (const void*)optimize_lean_shared_kernel : (const void*)optimize_lean_kernel,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds)) != hipSuccess)
      return e;
    if (ends_long)
      MRS_TG_LAUNCH_EXT(optimize_lean_shared_ends_long_kernel, dim3(blocks), dim3(64), plds, stream, kt.start, nullptr, 0, b, prm, bt,
                            mask, vals, seg_times, nl.d_opt_status, nl.d_fallback);
    else if (ends_shared)
      MRS_TG_LAUNCH_EXT(optimize_lean_shared_ends_kernel, dim3(blocks), dim3(64), plds, stream, kt.start, nullptr, 0, b, prm, bt,
                            mask, vals, seg_times, nl.d_opt_status, nl.d_fallback);
    else if (lean_masked)
      MRS_TG_LAUNCH_EXT(optimize_lean_masked_kernel, dim3(blocks), dim3(64), plds, stream, kt.start, nullptr, 0, b, prm, bt, mask,
                            vals, seg_times, nl.d_opt_status, nl.d_fallback);
    else if (lean_shared_only)
      MRS_TG_LAUNCH_EXT(optimize_lean_shared_kernel, dim3(blocks), dim3(64), plds, stream, kt.start, nullptr, 0, b, prm, bt, mask,
                            vals, seg_times, nl.d_opt_status, nl.d_fallback);
    else
      MRS_TG_LAUNCH_EXT(optimize_lean_kernel, dim3(blocks), dim3(64), plds, stream, kt.start, nullptr, 0, b, prm, bt, mask, vals,
                            seg_times, nl.d_opt_status, nl.d_fallback);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    prm.only_flagged = nl.d_fallback;
    prm.queue_next = nullptr;
  }
  {
    const hipEvent_t ev_start = lean ? nullptr : kt.start, ev_stop = (careful || general) ? nullptr : kt.stop;
    if (!lean && wave_kernel_applies(b, dim_split)) {
      // one wavefront per path, both directions of the two-sided evaluation in it (mrs_tg_wave.hip)
      if ((e = launch_optimize_wave(b, prm, mask, vals, seg_times, nl.d_opt_status, stream, ev_start, ev_stop)) != hipSuccess) return e;
    } else {
    BinTable bt{};
    bt.n = (int)plan_bins.size();
    if (bt.n > 5) return hipErrorInvalidValue;
    size_t lds_bytes = 0;
    int blocks = 0;
    bool all_single = true;
    for (int i = 0; i < bt.n; ++i) {
      const NonlinearBin& bin = plan_bins[i];
      const int per_block = 64 / bin.group;
      bt.group[i] = bin.group;
      bt.q_begin[i] = bin.q_begin;
      bt.q_count[i] = bin.q_count;
      bt.max_S[i] = bin.max_S;
      bt.block_begin[i] = blocks;
      blocks += (int)cdiv_u(bin.q_count, per_block);
      const size_t need = ((size_t)per_block * group_lds_doubles(bin.max_S, dim_split == 4) + kBlockConsts) * sizeof(double);
      if (need > lds_bytes) lds_bytes = need;
      if (bin.group != 64) all_single = false;
    }
    // one path per block and one dimension per lane: a partner wavefront runs the other half of every sweep
    const unsigned threads = (dim_split == 4 && all_single) ? 128u : 64u;
    if (threads == 128u) lds_bytes += (64 * kPairState + 2) * sizeof(double);  // hand-over area + flags
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    // objective order below snap: the end vertices of rest-to-rest paths keep free slots, so the large-batch kernel is
    // launched with the masked step compiled in (the min-snap instantiation stays free of it)
    const bool masked4 = prm.derivative < 4;
    if (lds_bytes > 64 * 1024) {
      const void* fn = dim_split == 4 ? (const void*)optimize_split_kernel
                       : masked4         ? (const void*)optimize_compact_kernel<true>
                                         : (const void*)optimize_compact_kernel<false>;
      e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      if (e != hipSuccess) return e;
    }
    if (dim_split == 4)
      MRS_TG_LAUNCH_EXT(optimize_split_kernel, dim3(blocks), dim3(threads), lds_bytes, stream, ev_start, ev_stop, 0, b, prm, bt,
                            mask, vals, seg_times, nl.d_opt_status);
    else if (masked4)
      MRS_TG_LAUNCH_EXT(optimize_compact_kernel<true>, dim3(blocks), dim3(64), lds_bytes, stream, ev_start, ev_stop, 0, b, prm,
                            bt, mask, vals, seg_times, nl.d_opt_status);
    else
      MRS_TG_LAUNCH_EXT(optimize_compact_kernel<false>, dim3(blocks), dim3(64), lds_bytes, stream, ev_start, ev_stop, 0, b, prm,
                            bt, mask, vals, seg_times, nl.d_opt_status);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
#if MRS_TG_WITH_CAREFUL
    if (careful) {
      const size_t clds = ((size_t)group_lds_doubles(b.max_segments, true) + kBlockConsts) * sizeof(double);
      if (clds > 160 * 1024) return hipErrorInvalidValue;
      if (clds > 64 * 1024 &&
          (e = hipFuncSetAttribute((const void*)optimize_careful_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)clds)) != hipSuccess)
        return e;
      MRS_TG_LAUNCH_EXT(optimize_careful_kernel, dim3(careful_cap), dim3(64), clds, stream, nullptr,
                            general ? nullptr : kt.stop, 0, b, prm, mask, vals, seg_times, nl.d_opt_status, nl.d_careful_ws);
      if ((e = hipGetLastError()) != hipSuccess) return e;
      MRS_TG_LAUNCH(careful_close_kernel, dim3(1), dim3(1), 0, stream, nl.d_careful);
      if ((e = hipGetLastError()) != hipSuccess) return e;
    }
#else
    if (careful) return hipErrorNotSupported;  // (mrs_tg_plan_solve refuses the flag before it gets here)
#endif
    // the paths none of the kernels above knows (what they wrote for them is overwritten here): the same search with the
    // 5 x 5-block evaluation, from the start times kept aside, `general_cap` listed paths per launch
    if (general) {
      const size_t glds = ((size_t)group_lds_doubles(b.max_segments, true) + kBlockConsts) * sizeof(double);
      if (glds > 160 * 1024) return hipErrorInvalidValue;
      if (glds > 64 * 1024 &&
          (e = hipFuncSetAttribute((const void*)optimize_general_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds)) != hipSuccess)
        return e;
      NonlinearParams gp = prm;
      gp.only_flagged = nullptr;
      gp.queue_next = nullptr;
      gp.careful_count = nl.d_general;
      gp.careful_cap = general_cap;
      for (int off = 0; off < b.n_paths; off += general_cap) {
        const bool last = off + general_cap >= b.n_paths;
        gp.careful_list = nl.d_general + 4 + b.n_paths + off;
        MRS_TG_LAUNCH_EXT(optimize_general_kernel, dim3(general_cap), dim3(64), glds, stream, nullptr, last ? kt.stop : nullptr,
                              0, b, gp, mask, vals, seg_times, nl.d_opt_status, nl.d_general_ws, nl.d_general_t0);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if (!last) {
          MRS_TG_LAUNCH(general_advance_kernel, dim3(1), dim3(1), 0, stream, nl.d_general, general_cap);
          if ((e = hipGetLastError()) != hipSuccess) return e;
        }
      }
    }
  }
  // 2-4 in one launch for the small batches whose wavefronts hold one path (solve_rows_pipeline_kernel, mrs_tg_rows.hip)
  if (!general && !quad_kernel_applies(b, b.n_paths, false) && rows_pipeline_applies(b)) {
    RowsTail tail;
    tail.maxima_in_launch = true;
    tail.limits = limits;
    tail.opt_status = nl.d_opt_status;
    tail.sum_t0 = prm.reference_status ? nullptr : nl.d_sum_t0;  // (no runaway test: MRS_TG_FLAG_REFERENCE_STATUS)
    tail.seg_times_out = seg_times;
    if (sampling_dt > 0.0 && n_samples != nullptr) {
      tail.sampling_dt = sampling_dt;
      tail.sample_capacity = sample_capacity;
      tail.n_samples = n_samples;
      tail.samples = samples;
      if (sampled_out) *sampled_out = true;
    }
    return launch_solve_rows(b, prm.derivative, mask, vals, seg_times, coeffs, status, cost, nl.d_opt_status, stream, tail);
  }
  // 2. trajectory of the last evaluated point (scaleSegmentTimesWithViolation works on poly_opt_'s state)
  if ((e = launch_solve_linear(b, prm.derivative, true, mask, vals, seg_times, nullptr, nullptr, nl.d_ws, coeffs, nullptr,
                               nullptr, nullptr, stream, prm.pos_wp)) != hipSuccess)
    return e;
  if (general && (e = launch_solve_general(b, prm.derivative, mask, vals, seg_times, nl.d_general_solve_ws, coeffs, nullptr, nullptr,
                                           stream, general_flag, nullptr)) != hipSuccess)
    return e;
  // 3. per-segment maxima and time scaling
  static const bool certified_maxima = [] {  // MRS_TG_MAXIMA_BOUNDS=0: every entry searched (tuning / test knob, read once)
    const char* e = std::getenv("MRS_TG_MAXIMA_BOUNDS");
    return e == nullptr || std::atoi(e) != 0;
  }();
  if (certified_maxima)
    MRS_TG_LAUNCH(segment_maxima_scaling_kernel, dim3(cdiv_u(b.n_segments, kMsSegs)), dim3(kMsThreads), 0, stream, b, coeffs,
                       seg_times, limits, nl.d_maxima);
  else
    MRS_TG_LAUNCH(segment_maxima9_kernel, dim3(cdiv_u(b.n_segments, 64), 9), dim3(64), 0, stream, b.n_segments,
                       coeffs, seg_times, nl.d_maxima);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  // 3b + 4 in one launch where the rows kernel applies: its staging pass scales the times of its own path, its tail samples
  const bool want_samples = sampling_dt > 0.0 && n_samples != nullptr && rows_tail_sampling_pays(b);
  if (!general && quad_kernel_applies(b, b.n_paths, false)) {  // saturated device; the caller's sampler follows
    RowsTail tail;
    tail.maxima = nl.d_maxima;
    tail.limits = limits;
    tail.opt_status = nl.d_opt_status;
    tail.sum_t0 = prm.reference_status ? nullptr : nl.d_sum_t0;  // (no runaway test: MRS_TG_FLAG_REFERENCE_STATUS)
    tail.seg_times_out = seg_times;
    tail.pos_wp = prm.pos_wp;
    return launch_solve_quad(b, prm.derivative, mask, vals, seg_times, coeffs, status, cost, nl.d_opt_status, nl.d_ws, stream, tail);
  }
  if (!general && rows_kernel_applies(b, want_samples)) {
    RowsTail tail;
    tail.maxima = nl.d_maxima;
    tail.limits = limits;
    tail.opt_status = nl.d_opt_status;
    tail.sum_t0 = prm.reference_status ? nullptr : nl.d_sum_t0;  // (no runaway test: MRS_TG_FLAG_REFERENCE_STATUS)
    tail.seg_times_out = seg_times;
    if (want_samples) {
      tail.sampling_dt = sampling_dt;
      tail.sample_capacity = sample_capacity;
      tail.n_samples = n_samples;
      tail.samples = samples;
      if (sampled_out) *sampled_out = true;
    }
    return launch_solve_rows(b, prm.derivative, mask, vals, seg_times, coeffs, status, cost, nl.d_opt_status, stream, tail);
  }
  MRS_TG_LAUNCH(apply_scaling_kernel, dim3(cdiv_u(b.n_segments, 256)), dim3(256), 0, stream, b, nl.d_maxima, limits,
                     nl.d_opt_status, seg_times);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if (!prm.reference_status) {
    MRS_TG_LAUNCH(runaway_kernel, dim3(cdiv_u(b.n_paths, 256)), dim3(256), 0, stream, b, seg_times, nl.d_sum_t0, nl.d_opt_status);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  // 4. updateSegmentTimes + solveLinear with the scaled times (nonlinear_impl.h:405-408), final status
  if ((e = launch_solve_linear(b, prm.derivative, true, mask, vals, seg_times, nullptr, nullptr, nl.d_ws, coeffs, status, cost,
                               nl.d_opt_status, stream)) != hipSuccess)
    return e;
  if (general)
    return launch_solve_general(b, prm.derivative, mask, vals, seg_times, nl.d_general_solve_ws, coeffs, status, cost, stream,
                                general_flag, nl.d_opt_status);
  return hipSuccess;
}

hipError_t launch_cost_gradient(NonlinearPlan& nl, const BatchView& b, int d, const uint8_t* mask, const double* vals,
                                const double* seg_times, double* cost, double* grad, hipStream_t stream) {
  hipError_t e;
  const int32_t* only = nullptr;
  for (const NonlinearBin& bin : nl.bins) {
    const int per_block = 64 / bin.group;
    const size_t lds_bytes = ((size_t)per_block * gradient_lds_doubles(bin.max_S, nl.dim_split == 4) + kBlockConsts) * sizeof(double);
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    const dim3 grid(cdiv_u(bin.q_count, per_block));
    if (nl.dim_split == 4)
      MRS_TG_LAUNCH(cost_gradient_kernel<4>, grid, dim3(64), lds_bytes, stream, b, d, bin.group, bin.q_begin,
                         bin.q_count, bin.max_S, mask, vals, seg_times, cost, grad, only);
    else
      MRS_TG_LAUNCH(cost_gradient_kernel<1>, grid, dim3(64), lds_bytes, stream, b, d, bin.group, bin.q_begin,
                         bin.q_count, bin.max_S, mask, vals, seg_times, cost, grad, only);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t launch_segment_maxima(const BatchView& b, const double* coeffs, const double* seg_times, double* maxima,
                                 hipStream_t stream) {
  if (b.n_segments == 0) return hipSuccess;
  MRS_TG_LAUNCH(segment_maxima9_kernel, dim3(cdiv_u(b.n_segments, 64), 9), dim3(64), 0, stream, b.n_segments, coeffs,
                     seg_times, maxima);
  return hipGetLastError();
}

}  // namespace mrs_tg
