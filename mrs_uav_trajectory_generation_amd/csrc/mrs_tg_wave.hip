// mrs_tg_wave.hip -- the Mellinger outer loop with ONE wavefront per path: small batches of paths with at most 12 segments
// (what a drop-in nodelet sends: one request, or a few hundred of them).
//
// Reference behaviour being reproduced (paths relative to /root/reference/):
//   include/eth_trajectory_generation/impl/polynomial_optimization_nonlinear_impl.h
//     :160-234  optimizeTimeMellingerOuterLoop      :257-333  getCostAndGradientMellinger
//     :617-649  objectiveFunctionTimeMellingerOuterLoop
// with the project's own projected L-BFGS in place of NLopt's LD_LBFGS (CPU twin oracle/mto_nonlinear.c).
//
// Why a kernel of its own.  An objective evaluation is the cost at S + 1 time vectors: x itself and, for every segment j,
// B' = max(x - h / (S - 1), 0.01) with component j replaced by x_j + h.  Seen from the two-sided elimination, whose halves
// meet at vertex m = S / 2, the left half sweep (segments 0 .. m-1) of EVERY vector that perturbs a segment of the right half
// is one and the same sweep over B', and the other way round: an evaluation has m + 2 distinct left halves (x, B', one per
// left segment) and S - m + 2 distinct right halves -- S + 4 half sweeps where optimize_split_kernel's two wavefronts run
// 2 (S + 1).  With one dimension per lane that is 4 (S + 4) <= 64 lanes for S <= 12: both directions of a path fit ONE
// wavefront, the dependent chain of an evaluation (S - m segment steps and the join) is issued once instead of twice, and the
// SIMD that held two wavefronts of one path holds two paths.  A lane's direction is data, not code: every lane runs
// FastStep's left-to-right forms on the block constants of ITS direction (stage_ps_tables: near / far swapped and the coupling
// transposed for a right-to-left lane) and on brackets read with its direction's offsets.  The join: the three half sweeps
// more than one vector needs (right half of x, right half of B', left half of B') are published in LDS, the S + 1 other lane
// quads add their partner's state to their own and factor the middle vertex.  The arithmetic of every half sweep and join is
// that of the two-wavefront evaluation, on the same numbers.
//
//   lane quads (quad = lane / 4, lane % 4 = dimension):
//     0: x, left | 1 .. m: vector k = quad, left | m + 1: B', left |
//     m + 2: x, right | m + 2 + r: vector k = m + r, right (r = 1 .. S - m) | S + 3: B', right
//
// The optimiser's bookkeeping is written for exactly this shape -- one path, vectors of <= 12 elements, element i in lane i
// of row 0 -- with every decision a scalar branch: no group width, no queue, no partner wavefront.  Paths the shared
// evaluation does not take (fewer than 4 segments, constraint patterns without a specialised step) run the same
// bookkeeping around the one-sided sweeps of evaluate_objective, behind a real call so that their general segment step does
// not weigh on the register allocation of the common path.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "mrs_tg_device.hpp"
#include "mrs_tg_estimate.hpp"
#include "mrs_tg_nl_common.hpp"
#include "mrs_tg_nonlinear.h"
#include "mrs_tg_sweep.hpp"
#include "mrs_tg_wave.h"

#ifndef MRS_TG_WAVE_WAVES
#define MRS_TG_WAVE_WAVES 1
#endif

namespace mrs_tg {

// LDS of a wavefront (doubles).  Everything is sized for kWaveMaxS, so every offset is a constant.
constexpr int kWvVec = 16;  // stride of the optimiser's vectors (>= kWaveMaxS, 128-byte rows)
constexpr int kWvX = 0, kWvG = kWvX + kWvVec, kWvXn = kWvG + kWvVec, kWvGn = kWvXn + kWvVec, kWvDir = kWvGn + kWvVec;
constexpr int kWvSm = kWvDir + kWvVec;                  // [kLbfgsM][kWvVec]
constexpr int kWvYm = kWvSm + kLbfgsM * kWvVec;         // [kLbfgsM][kWvVec]
constexpr int kWvRho = kWvYm + kLbfgsM * kWvVec;        // [kLbfgsM] 1 / s^T y | [kLbfgsM] s^T y / y^T y of the newest pair
constexpr int kWvFlags = kWvRho + kLbfgsM + 1;          // [0] an evaluation failed the guard (int)
constexpr int kWvTabs = kWvFlags + 1;                   // [2][kPsTable] direction tables
constexpr int kWvPub = kWvTabs + 2 * kPsTable;          // [16][12] hand-over area of the join
constexpr int kWvHc = kWvPub + 3 * 4 * 16;              // [kBlockConsts] block constants of the one-sided sweeps
constexpr int kWvVtx = kWvHc + kBlockConsts;            // [(kWaveMaxS + 1)][kVtxLds]
constexpr int kWvSeg = kWvVtx + (kWaveMaxS + 1) * kVtxLds + kStartExtra;  // the moving-start extras in front of the records
constexpr int kWvTotal = kWvSeg + kWaveMaxS * kSegLds;

// sum over lanes 0..15 (the vectors live in row 0; every other lane holds 0.0), delivered to every lane
template <int N>
__device__ __forceinline__ void row0_sum(double (&v)[N]) {
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
}

// a condition that is the same in every lane, as a scalar (v_cmp into an SGPR pair + s_cmp: no round trip through a VGPR)
__device__ __forceinline__ bool uniform(bool c) { return __ballot(c) != 0ull; }

// segment_powers with the objective order as three select masks (bits of d) instead of a branch tree per segment step:
// T^d = (d & 1 ? T : 1) (d & 2 ? T^2 : 1) (d & 4 ? T^4 : 1), the same products as segment_powers forms (times exact ones)
__device__ __forceinline__ void segment_powers_bits(double T, bool d1, bool d2, bool d4, double (&p2)[9]) {
  const double t2 = T * T;
  const double t4 = t2 * t2;
  const double td = ((d1 ? T : 1.0) * (d2 ? t2 : 1.0)) * (d4 ? t4 : 1.0);
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

// what a lane does in an evaluation; fixed once the path's segment count is known
struct WaveRole {
  int k;         // time vector of the lane's half sweep
  int nhalf;     // segments of the lane's half
  int i0, di;    // first segment and step of the sweep (left: 0, +1; right: S - 1, -1)
  int pub;       // offset of the slot the lane publishes to, -1: none
  int src;       // offset of the slot the lane's join reads, -1: the lane does not join
  bool left, valid;
};

__device__ __forceinline__ WaveRole wave_role(int S, int lane) {
  WaveRole w;
  const int m = S >> 1, nL = m + 2;
  const int quad = lane >> 2, dim = lane & 3;
  w.left = quad < nL;
  const int r = w.left ? quad : quad - nL;
  w.nhalf = w.left ? m : S - m;
  w.valid = quad < S + 4;
  const bool pure = r == w.nhalf + 1, base = r == 0;
  // B' on this half = any vector that perturbs a segment of the other half
  w.k = base ? 0 : pure ? (w.left ? S : 1) : (w.left ? r : m + r);
  w.i0 = w.left ? 0 : S - 1;
  w.di = w.left ? 1 : -1;
  // slots: 0 right half of x | 4 right half of B' | 8 left half of B'
  w.pub = !w.valid ? -1 : (pure ? (w.left ? 8 : 4) : ((base && !w.left) ? 0 : -1));
  w.src = (!w.valid || w.pub >= 0) ? -1 : (w.left ? (base ? 0 : 4) : 8);
  if (w.pub >= 0) w.pub += dim;
  if (w.src >= 0) w.src += dim;
  return w;
}

// SPECIAL: 0 plain path (start | interior ... | end), 1 the first segment starts from a moving state, 2 masked segments (and
// possibly a moving start); `special` as in half_sweep: bit i = segment i takes the masked step, bit 31 = moving start
template <int SPECIAL>
__device__ __forceinline__ double wave_evaluate(const double* lds, const WaveRole& w, int S, int d, unsigned special, int lane,
                                                double* grad, bool clocked = false) {
  const double* seg = lds + kWvSeg;
  const double* pt = lds + kWvXn;
  double* pub = const_cast<double*>(lds) + kWvPub;
  const int dim = lane & 3;
  // the block constants of the lane's direction
  double tab[kBlockConsts];
  {
    const double* tsrc = lds + kWvTabs + (w.left ? 0 : kPsTable);
#pragma unroll
    for (int e = 0; e < kBlockConsts; ++e) tab[e] = tsrc[e];
  }
#ifdef MRS_TG_PHASE_CLOCKS
  asm volatile("" ::"v"(tab[0]), "v"(tab[35]) : "memory");
  if (clocked) MRS_TG_PHASE_MARK(20);
#endif
  const int bracket_near = w.left ? 0 : kNB, bracket_far = w.left ? kNB : 0;
  const double corr = kGradStep / ((double)S - 1.0);
  const bool d1 = (d & 1) != 0, d2 = (d & 2) != 0, d4 = (d & 4) != 0;
  Elim<1> st;
  st.init();
  const int nmax = S - (S >> 1);
  const double* sr = seg + (size_t)w.i0 * kSegLds + dim * 9;
  const double* tp = pt + w.i0;
  int i = w.i0;
  for (int s = 0; s < nmax; ++s) {
    if (w.valid && s < w.nhalf) {
      double T = *tp;
      if (w.k > 0) T = (i == w.k - 1) ? T + kGradStep : fmax(T - corr, kTimeLowerBound);
      double p2[9];
      segment_powers_bits(T, d1, d2, d4, p2);
      FastStep<1> fast;
#pragma unroll
      for (int j = 0; j < kNB; ++j) {
        fast.w[0][j] = sr[bracket_near + j];
        fast.w[0][kNB + j] = sr[bracket_far + j];
      }
      fast.w[0][8] = sr[8];
      if (SPECIAL == 2 && ((special >> i) & 1u)) {
        const unsigned masks = (unsigned)seg[(size_t)i * kSegLds + 37];
        const unsigned ms = masks & 0xFu, me = masks >> 4;
        // a moving start with free slots beside its constrained values (kSegMaskedStartState): the start vertex's own
        // right-hand side in front of the masked step, vertex 1's and f^T H f behind it
        const bool moving_first = s == 0 && w.left && (special >> 31);
        const double* e = seg - kStartExtra + dim * kStartExtraDim;
        if (moving_first) {
#pragma unroll
          for (int r = 0; r < kNB; ++r) {
            double u = 0.0;
#pragma unroll
            for (int c = 1; c < kHalf; ++c) u = fma(e[24 + r * 4 + (c - 1)], p2[r + 1 + c], u);
            st.y[r][0] -= u;
          }
        }
        fast.template masked_t<false>(st, tab, p2, w.left ? ms : me, w.left ? me : ms);
        if (moving_first) {
#pragma unroll
          for (int r = 0; r < kNB; ++r) {
            double u = 0.0;
#pragma unroll
            for (int c = 1; c < kHalf; ++c) u = fma(e[r * 4 + (c - 1)], p2[r + 1 + c], u);
            st.y[r][0] -= ((me >> r) & 1u) ? u : 0.0;
          }
#pragma unroll
          for (int mm = 1; mm < 9; ++mm) st.qf = fma(e[16 + (mm - 1)], p2[mm], st.qf);
        }
      } else if (s == 0) {
        fast.template start_t<false>(st, tab, p2);
        if (SPECIAL != 0 && w.left && (special >> 31)) {  // the terms of the start vertex's derivative values (FastStep::start_state)
          const double* e = seg - kStartExtra + dim * kStartExtraDim;
#pragma unroll
          for (int r = 0; r < kNB; ++r) {
            double u = 0.0;
#pragma unroll
            for (int c = 1; c < kHalf; ++c) u = fma(e[r * 4 + (c - 1)], p2[r + 1 + c], u);
            st.y[r][0] -= u;
          }
#pragma unroll
          for (int mm = 1; mm < 9; ++mm) st.qf = fma(e[16 + (mm - 1)], p2[mm], st.qf);
        }
      } else {
        fast.template interior_t<false>(st, tab, p2);
      }
    }
    i += w.di;
    sr += w.di * kSegLds;
    tp += w.di;
  }
#ifdef MRS_TG_PHASE_CLOCKS
  asm volatile("" ::"v"(st.Sm[0]), "v"(st.red) : "memory");
  if (clocked) MRS_TG_PHASE_MARK(21);
#endif
  // hand-over: element e of (slot, dimension) at [e * 12 + slot + dimension] -- twelve consecutive doubles per store
  if (w.pub >= 0) {
    double* ps = pub + w.pub;
#pragma unroll
    for (int e = 0; e < 10; ++e) ps[e * 12] = st.Sm[e];
#pragma unroll
    for (int r = 0; r < kNB; ++r) ps[(10 + r) * 12] = st.y[r][0];
    ps[14 * 12] = st.qf;
    ps[15 * 12] = st.red;
  }
  ps_wave_sync();
  double Jk = 0.0, qfk = 0.0;
  if (w.src >= 0) {
    const double* ps = pub + w.src;
#pragma unroll
    for (int e = 0; e < 10; ++e) st.Sm[e] += ps[e * 12];
#pragma unroll
    for (int r = 0; r < kNB; ++r) st.y[r][0] += ps[(10 + r) * 12];
    st.qf += ps[14 * 12];
    st.red += ps[15 * 12];
    // free mask of the middle vertex = end mask of the segment in front of it
    if (SPECIAL == 2) FastStep<1>::apply_mask(st, (unsigned)seg[(size_t)((S >> 1) - 1) * kSegLds + 37] >> 4);
    FastStep<1> fs;
    double L[10], Linv[kNB], z[kNB][1];
    fs.factor(st, L, Linv, z);
    Jk = 0.5 * (st.qf - st.red);
    qfk = st.qf;
  }
#ifdef MRS_TG_PHASE_CLOCKS
  asm volatile("" ::"v"(Jk) : "memory");
  if (clocked) MRS_TG_PHASE_MARK(22);
#endif
  Jk += dpp_move<0xB1>(Jk);
  Jk += dpp_move<0x4E>(Jk);
  qfk += dpp_move<0xB1>(qfk);
  qfk += dpp_move<0x4E>(qfk);
  Jk = guarded_cost(Jk, qfk, w.k == 0);
  if (w.src >= 0 && Jk == kUnreliableCost) reinterpret_cast<int*>(const_cast<double*>(lds) + kWvFlags)[0] = 1;
  const double J0 = row_value(Jk, 0);  // lanes 0..3: the left half of x, joined with its right half
  if (w.src >= 0 && dim == 0 && w.k >= 1) grad[w.k - 1] = (Jk - J0) / kGradStep;
  return J0;
}

// any other path: the one-sided sweeps, four lanes per time vector (the 64-thread launch of optimize_split_kernel evaluates
// this way); a real call, see the head of the file
__device__ __attribute__((noinline)) double wave_evaluate_generic(double* lds, int S, int d, int lane) {
  int tripped = 0;
  const double J0 = evaluate_objective<4, false>(lds + kWvVtx, lds + kWvSeg, lds + kWvHc, S, d, lds + kWvXn, lds + kWvGn, lane, 64,
                                                 true, &tripped);
  if (tripped) reinterpret_cast<int*>(lds + kWvFlags)[0] = 1;
  return J0;
}

__global__ __launch_bounds__(64, MRS_TG_WAVE_WAVES) void optimize_wave_kernel(BatchView b, NonlinearParams prm,
                                                                                const uint8_t* __restrict__ mask,
                                                                                const double* __restrict__ vals,
                                                                                double* __restrict__ seg_times,
                                                                                int32_t* __restrict__ opt_status) {
  __shared__ double lds[kWvTotal];
  MRS_TG_PHASE_MARK(0);
  const int lane = threadIdx.x;
  const int q = blockIdx.x;
  const PathRef pr = path_at(b, q);
  const int S = __builtin_amdgcn_readfirstlane(pr.S);
  const int d = prm.derivative;
  double* x = lds + kWvX;
  double* gr = lds + kWvG;
  double* xn = lds + kWvXn;
  double* gn = lds + kWvGn;
  double* dir = lds + kWvDir;
  double* sm = lds + kWvSm;
  double* ym = lds + kWvYm;
  double* rho = lds + kWvRho;
  int* flags = reinterpret_cast<int*>(lds + kWvFlags);
  double* vtx = lds + kWvVtx;
  double* seg = lds + kWvSeg;
  const bool me = lane < S;

  // the start times are requested in front of the vertex staging (one trip to memory instead of two in a row)
  double t_first = 0.0;
  if (me)
    t_first = prm.estimate_wp ? estimate_segment_time(prm.estimate_wp + (size_t)(pr.v0 + lane) * 4, prm.estimate_limits + (size_t)pr.p * 9)
                              : seg_times[pr.s0 + lane];
  stage_ps_tables(d, lds + kWvTabs, lane, 64);
  stage_block_constants(d, lds + kWvHc, lane, 64);
  stage_vertices(mask, vals, pr.v0, S, vtx, lane, 64);
  if (lane == 0) flags[0] = 0;
  __syncthreads();
  stage_segments(vtx, S, d, seg, lane, 64, true);
  // start point; NLopt rejects a start below the lower bound (-> INVALID_ARGS)
  if (lane < kWvVec) {
    x[lane] = t_first;
    xn[lane] = t_first;
    gr[lane] = 0.0;
    gn[lane] = 0.0;
    dir[lane] = 0.0;
  }
  const bool bad = __ballot(me && t_first < kTimeLowerBound) != 0ull;
  {
    double t_sum[1] = {me ? t_first : 0.0};
    row0_sum<1>(t_sum);
    // the total time the search starts from: what the final solve measures a runaway of the feasibility scaling against
    if (lane == 0 && prm.sum_t0) prm.sum_t0[pr.p] = t_sum[0];
  }
  __syncthreads();
  MRS_TG_PHASE_MARK(1);

  // which evaluation takes the path: every segment needs a specialised step (a start-type or masked first one, an end-type
  // or masked last one, interior or masked ones in between), and 4 (S + 4) lanes
  int mode = -1;  // -1 generic, else SPECIAL of wave_evaluate
  unsigned special = 0u;
  if (S >= 4 && S <= kWaveMaxS) {
    int kind = kSegInterior;
    if (me) kind = (int)seg[(size_t)lane * kSegLds + 36];
    const bool first = lane == 0, last = lane == S - 1;
    const bool kind_ok = !me || kind == kSegMasked || (first ? (kind == kSegStart || kind == kSegStartState || kind == kSegMaskedStartState)
                                                             : last ? kind == kSegEnd : kind == kSegInterior);
    const unsigned long long masked = __ballot(me && (kind == kSegMasked || kind == kSegMaskedStartState));
    const bool moving = __ballot(first && (kind == kSegStartState || kind == kSegMaskedStartState)) != 0ull;
    if (__ballot(!kind_ok) == 0ull) {
      special = (unsigned)masked | (moving ? 1u << 31 : 0u);
      mode = masked ? 2 : moving ? 1 : 0;
    }
  }
  const WaveRole role = wave_role(S, lane);

  const int maxeval = prm.max_iterations;
  // the tolerances live in vector registers: left in SGPRs they are re-fetched from the kernel argument segment inside the
  // tick loop (the loop's lane masks crowd the scalar file), a scalar-cache round trip per use
  double f_rel = prm.f_rel, f_abs = prm.f_abs, x_rel = prm.x_rel, x_abs = prm.x_abs;
  asm volatile("" : "+v"(f_rel), "+v"(f_abs), "+v"(x_rel), "+v"(x_abs));
  // nlopt maxtime (src/mrs_trajectory_generation.cpp:899): ONE deadline per call, written to a device word in front of the
  // outer-loop launches; read once per evaluation
  const long long t_deadline = prm.deadline ? *prm.deadline : 0ll;
  bool timed_out = false;
  // nlopt checks the evaluation count first, then the clock (nlopt_stop_evals, nlopt_stop_time)
  auto budget_spent = [&](int n) { return (maxeval > 0 && n >= maxeval) || timed_out; };
  auto budget_code = [&](int n) { return (maxeval > 0 && n >= maxeval) ? 5 : 6; };

  // curvature pairs: slot k of sm / ym / rho holds the k-th oldest pair (slots >= npairs are read along and not used)
  if (lane < kWvVec) {
#pragma unroll
    for (int k = 0; k < kLbfgsM; ++k) sm[k * kWvVec + lane] = ym[k * kWvVec + lane] = 0.0;
    if (lane <= kLbfgsM) rho[lane] = 0.0;
  }
  int neval = 0, npairs = 0, ret = -1;
  double f = 0.0, alpha = 1.0;
  bool done = bad;
  while (!done) {
    // (1) one objective evaluation at the trial point xn: cost to every lane, gradient to gn
    double fn;
    if (mode == 0) fn = wave_evaluate<0>(lds, role, S, d, special, lane, gn, neval == 1);
    else if (mode == 1) fn = wave_evaluate<1>(lds, role, S, d, special, lane, gn);
    else if (mode == 2) fn = wave_evaluate<2>(lds, role, S, d, special, lane, gn);
    else fn = wave_evaluate_generic(lds, S, d, lane);
    __syncthreads();
    ++neval;
    timed_out = t_deadline != 0ll && (long long)wall_clock64() > t_deadline;
#ifdef MRS_TG_PHASE_CLOCKS
    if (neval <= 6) MRS_TG_PHASE_MARK(4 + 2 * neval);  // after evaluation #neval - 1
#endif
    // (2) accept / reject; element i of every vector in lane i.  Everything the tick may need is
    // requested from LDS here, in one batch.  Lanes S .. 15 hold zeros (nobody writes those words); lanes >= 16 read row 0's
    // words and compute along, but no sum (row 0 only), ballot (`me`) or store (`me`) looks at them.
    const int vl = lane & (kWvVec - 1);
    double xi = x[vl], gi = gr[vl];
    double xni = xn[vl], gni = gn[vl];
    double di = dir[vl];
    double sk[kLbfgsM], yk[kLbfgsM], rk[kLbfgsM];
#pragma unroll
    for (int k = 0; k < kLbfgsM; ++k) {
      sk[k] = sm[k * kWvVec + vl];
      yk[k] = ym[k * kWvVec + vl];
      rk[k] = rho[k];
    }
    double gamma = rho[kLbfgsM];  // s^T y / y^T y of the newest pair
    bool new_dir = false;
    if (neval == 1) {
      f = fn;
      xi = xni;
      gi = gni;
      if (budget_spent(neval)) {
        ret = budget_code(neval);
        done = true;
      } else {
        new_dir = true;
      }
    } else {
      // slope and the curvature sums of the would-be pair in one batched reduction
      const double si = xni - xi, yi = gni - gi;
      double red4[4] = {gi * si, si * yi, si * si, yi * yi};
#ifdef MRS_TG_PHASE_CLOCKS
    if (neval == 2) MRS_TG_PHASE_MARK(23);
#endif
      row0_sum<4>(red4);
#ifdef MRS_TG_PHASE_CLOCKS
    if (neval == 2) MRS_TG_PHASE_MARK(24);
#endif
      const double slope = red4[0];
      if (uniform(fn <= f + 1e-4 * slope)) {
        int stop = 0;
        if (uniform(relstop_flat(f, fn, f_rel, f_abs))) stop = 3;
        else if (__ballot(me & !relstop_flat(xi, xni, x_rel, x_abs)) == 0ull) stop = 4;
        const double sy = red4[1], ss = red4[2], yy = red4[3];
#ifdef MRS_TG_PHASE_CLOCKS
    if (neval == 2) MRS_TG_PHASE_MARK(25);
#endif
        const bool budget_out = budget_spent(neval);
        // curvature condition s^T y > 1e-10 |s| |y|, compared in squares (no square roots on the chain)
        if (!stop && !budget_out && uniform((sy > 0.0) & (sy * sy > 1e-20 * (ss * yy)))) {
          if (npairs == kLbfgsM) {  // forget the oldest pair: the others move down one slot, in LDS and in this tick's copy
#pragma unroll
            for (int k = 0; k + 1 < kLbfgsM; ++k) {
              sk[k] = sk[k + 1];
              yk[k] = yk[k + 1];
              rk[k] = rk[k + 1];
              if (me) {
                sm[k * kWvVec + lane] = sk[k];
                ym[k * kWvVec + lane] = yk[k];
              }
              if (lane == 0) rho[k] = rk[k];
            }
            npairs = kLbfgsM - 1;
          }
          // 1 / s^T y and the scaling of the initial Hessian s^T y / y^T y by refined reciprocals (~1 ulp; an IEEE division
          // is a chain of a dozen dependent operations)
          const double rho_new = rcp_refined(sy);
          gamma = sy * rcp_refined(yy);
#pragma unroll
          for (int k = 0; k < kLbfgsM; ++k)  // (a ladder of scalar branches: a run-time index would send the arrays to scratch)
            if (k == npairs) {
              sk[k] = si;
              yk[k] = yi;
              rk[k] = rho_new;
            }
          if (me) {
            sm[npairs * kWvVec + lane] = si;
            ym[npairs * kWvVec + lane] = yi;
          }
          if (lane == 0) {
            rho[npairs] = rho_new;
            rho[kLbfgsM] = gamma;
          }
          ++npairs;
        }
#ifdef MRS_TG_PHASE_CLOCKS
    if (neval == 2) MRS_TG_PHASE_MARK(26);
#endif
        xi = xni;
        gi = gni;
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
        xi = xni;  // budget ends on a rejected trial: the last evaluated point is what the reference keeps
        ret = budget_code(neval);
        done = true;
      } else {
        alpha *= 0.5;
        if (uniform(alpha < 1e-12)) {
          xi = xni;
          ret = 4;
          done = true;
        }
      }
    }
    if (me) {
      x[lane] = xi;
      gr[lane] = gi;
    }
#ifdef MRS_TG_PHASE_CLOCKS
    if (neval == 2) MRS_TG_PHASE_MARK(28);  // tick 1: accept step done
#endif
    // (3) search direction: L-BFGS two-loop recursion, projected on the lower bound
#ifdef MRS_TG_PHASE_CLOCKS
    if (neval == 2) MRS_TG_PHASE_MARK(29);
#endif
    if (new_dir) {
      di = -gi;
      if (npairs > 0) {
        double al[kLbfgsM];
#pragma unroll
        for (int k = 0; k < kLbfgsM; ++k) al[k] = 0.0;
#ifdef MRS_TG_PHASE_CLOCKS
    if (neval == 2) MRS_TG_PHASE_MARK(27);
#endif
#pragma unroll
        for (int k = kLbfgsM - 1; k >= 0; --k)
          if (k < npairs) {
            double t[1] = {sk[k] * di};
            row0_sum<1>(t);
            al[k] = rk[k] * t[0];
            di -= al[k] * yk[k];
          }
        di *= gamma;
#pragma unroll
        for (int k = 0; k < kLbfgsM; ++k)
          if (k < npairs) {
            double t[1] = {yk[k] * di};
            row0_sum<1>(t);
            const double beta = rk[k] * t[0];
            di += (al[k] - beta) * sk[k];
          }
      }
#ifdef MRS_TG_PHASE_CLOCKS
    if (neval == 2) MRS_TG_PHASE_MARK(31);
#endif
      if (xi <= kTimeLowerBound && di < 0.0) di = 0.0;
      double red3[3] = {gi * di, xi * xi, di * di};
      row0_sum<3>(red3);
      double gd = red3[0];
      const double nx = red3[1];
      double nd = red3[2];
      if (!uniform(gd < 0.0)) {
        // not a descent direction: projected steepest descent, forget the curvature pairs
        double v = -gi;
        if (xi <= kTimeLowerBound && v < 0.0) v = 0.0;
        di = v;
        double red2[2] = {gi * v, v * v};
        row0_sum<2>(red2);
        gd = red2[0];
        nd = red2[1];
        npairs = 0;
        if (!uniform(gd < 0.0)) {  // projected gradient vanishes
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
      if (me) dir[lane] = di;
    }
    // (4) next trial point
    if (!done && me) xn[lane] = fmax(xi + alpha * di, kTimeLowerBound);
#ifdef MRS_TG_PHASE_CLOCKS
    if (neval == 2) MRS_TG_PHASE_MARK(30);
#endif
    __syncthreads();
#ifdef MRS_TG_PHASE_CLOCKS
    if (neval <= 6) MRS_TG_PHASE_MARK(5 + 2 * neval);  // end of the tick of evaluation #neval - 1
#endif
  }
  MRS_TG_PHASE_MARK(5);

  // hand in the result: the last evaluated point and the stopping reason.  A path whose cost failed the guard somewhere is
  // handed to the careful re-run instead (MRS_TG_FLAG_CAREFUL_COST): listed, its start times left in place
  bool listed = false;
  if (prm.careful_count && !bad && flags[0] != 0) {
    int okl = 0;
    if (lane == 0) {
      const int idx = atomicAdd(prm.careful_count, 1);
      okl = idx < prm.careful_cap;
      if (okl) prm.careful_list[idx] = q;
    }
    listed = __builtin_amdgcn_readfirstlane(okl) != 0;
  }
  if (!listed && me) seg_times[pr.s0 + lane] = x[lane];
  if (lane == 0) opt_status[pr.p] = bad ? -2 : ret;
}

bool wave_kernel_applies(const BatchView& b, int dim_split) {
  static const bool on = [] {
    const char* e = std::getenv("MRS_TG_WAVE_KERNEL");  // tuning / test knob, read once per process
    return e == nullptr || std::atoi(e) != 0;
  }();
  return on && dim_split == 4 && b.n_paths > 0 && b.max_segments <= kWaveMaxS;
}

hipError_t launch_optimize_wave(const BatchView& b, const NonlinearParams& prm, const uint8_t* mask, const double* vals,
                                double* seg_times, int32_t* opt_status, hipStream_t stream, hipEvent_t ev_start,
                                hipEvent_t ev_stop) {
  MRS_TG_LAUNCH_EXT(optimize_wave_kernel, dim3((unsigned)b.n_paths), dim3(64), 0, stream, ev_start, ev_stop, 0, b, prm, mask,
                        vals, seg_times, opt_status);
  return hipGetLastError();
}

}  // namespace mrs_tg
