// mrs_tg_quad.hip -- the linear QP solve for SATURATED devices: four lanes per path, the factors of the block elimination in LDS.
//
// What it replaces: PolynomialOptimization::updateSegmentTimes + constructR + solveLinear +
// updateSegmentsFromCompactConstraints + computeCost for every path of a batch
// (/root/reference/include/eth_trajectory_generation/impl/polynomial_optimization_linear_impl.h:289-304, 311-373,
// 264-282, 128-141) -- the same contract as mrs_tg_rows.hip, chosen by the launcher once a launch carries more paths than
// the device has wavefront slots for the rows kernel.
//
// Why a second solve kernel.  solve_rows_kernel spends a whole wavefront on one or two paths (one lane per unknown, rank-one
// updates by DPP row broadcasts): the shortest dependent chain there is, ~1900 wavefront instructions per path -- 3 % of the
// lanes carry useful work.  With 65536 paths every SIMD sees 64 of them one after the other, and the solve is bound by that
// instruction count (202 us).  Here a quad of lanes owns a path (lane = dimension) and runs the block-tridiagonal Cholesky of
// the vertex chain sequentially: 16 paths per wavefront, ~5000 wavefront instructions for all of them -- ~300 per path.  What
// made the four-lanes-per-path kernel of round 1 slow was its factor store (L, W, z of every vertex, written for the back
// substitution) in global memory: 30 doubles per vertex and LANE, 250 MB out and back in at 65536 x 10.  Here the store is in
// LDS, once per PATH (the four lanes of a quad hold the same L: lane 0 writes it, every lane its own z; W is formed again in
// the back substitution instead of being kept): 26 doubles per vertex, 31 KB per wavefront at 10 segments, five wavefronts
// per CU.  Barely more than one wavefront per SIMD -- a step's latencies are exposed, so positions are requested three
// segments ahead -- and still a multiple of the rows kernel's throughput.
//
//   prologue  segment times -> LDS (the feasibility scaling of a Mellinger pipeline's last solve applied on the way);
//             is every path of the wavefront "plain" (start | interior ... | end: position-only interior vertices, fully
//             constrained ends at rest)?
//   plain     forward: per segment near block + factor + W + Schur complement, all from the exact unit-time constants times
//             powers of T (as FastStep::interior_t); backward: x_v = L^-T (z - W x_{v+1}), c = A^-1 d segment by segment,
//             cost 0.5 c^T Q c as the reference computes it
//   other     the general masked step of mrs_tg_solve.hpp::solve_path (factors in the plan's global workspace)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "mrs_tg_device.hpp"
#include "mrs_tg_launch.h"
#include "mrs_tg_nl_common.hpp"
#include "mrs_tg_solve.hpp"

// wavefronts per SIMD the kernels are compiled for.  The single launch keeps its 262 registers (one wavefront per SIMD; LDS
// would allow 1.25): at 65536 x 10 the two-wavefront build is 8 % slower (137 vs 127 us, 20 B of scratch).  The grouped launch
// is compiled for two (256 VGPRs + 12 B): a host that keeps several dispatches in flight then has 1280 instead of 1024
// wavefront slots for them -- grouped headline over 200 steps 485 -> 539-556 M/s, over 20 steps 339 -> 345-357 with ten
// steps per dispatch and twenty batches in flight.
#ifndef MRS_TG_QUAD_WAVES
#define MRS_TG_QUAD_WAVES 1
#endif
#ifndef MRS_TG_QUAD_EXP
#define MRS_TG_QUAD_EXP 0
#endif
#ifndef MRS_TG_QUAD_GROUP_WAVES
#define MRS_TG_QUAD_GROUP_WAVES 2
#endif
// The two-sided group kernel: two as well.  Its 18.5 KB of LDS per wavefront (10 segments) let a CU hold eight wavefronts -- two
// per SIMD -- whatever the register budget says, so a build for three (168 registers: the plain path fits, the general step
// spills) only pays for the registers: 25.9 us per dispatch against 24.4, the region 67-69 us against 63 (same box, alternating
// runs; profiles/round6_duo_group_waves_ab.txt).  One wavefront of this kernel keeps its SIMD's FP64 pipe about half busy (an
// FMA on three vector registers issues every 8.7 clocks from one wavefront: profiles/round6_issue_latencies.txt).
#ifndef MRS_TG_DUO_GROUP_WAVES
#define MRS_TG_DUO_GROUP_WAVES 2
#endif

namespace mrs_tg {

constexpr int kQdPaths = 16;  // paths per wavefront
// LDS record of (vertex, path): L off-diagonal [6] | 1 / diag L [4] | z [4 rows][4 dimensions].  W = L^-1 E is NOT kept: the
// back substitution forms W x_{v+1} as L^-1 (E x_{v+1}) from the segment's time again (~25 instructions more per vertex,
// 16 doubles less per record: five wavefronts per CU instead of three at 10 segments)
constexpr int kQdL = 0, kQdLinv = 6, kQdZ = 10, kQdRec = 26;

// ends: records for the two end vertices as well (solve_quad_body<WP, true>: end vertices with free slots are eliminated)
__host__ __device__ constexpr size_t quad_lds_doubles(int Smax, bool ends = false) {
  return (size_t)(ends ? Smax + 1 : (Smax > 1 ? Smax - 1 : 1)) * kQdRec * kQdPaths + (size_t)Smax * kQdPaths  // records | times
         + (ends ? (size_t)(Smax + 1) * 2 : 0)  // ends: the free mask of every (vertex, path), one byte each
#if MRS_TG_QUAD_EXP == 5
         + kQdPaths  // per path: first segment and segment count (ints)
#endif
      ;
}

// T^(m + 1 - 2d), m = 0..8, the objective order as select masks (no branch tree per segment)
__device__ __forceinline__ void quad_powers(double T, bool d1, bool d2, bool d4, double (&p2)[9]) {
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

// a pair of coefficients (16 bytes).  MRS_TG_COEFF_NT=1 (experiment build): as a streaming store -- measured in round 6 after
// the samples and the assembled blocks had gained from it: here it is 2.5-3 x SLOWER (headline dispatch 26 -> 73-81 us, 65536 x 10
// 131 -> 335-339 us): a lane owns 80 consecutive bytes, so a store instruction writes 64 separate 16-byte pieces, and without the
// L2 to combine them every piece is a partial-line write to HBM.  Streaming pays where an instruction writes whole lines.
#ifndef MRS_TG_COEFF_NT
#define MRS_TG_COEFF_NT 0
#endif
__device__ __forceinline__ void store_coeff_pair(double2* dst, double a, double b) {
#if MRS_TG_COEFF_NT
  typedef double coeff_pair __attribute__((ext_vector_type(2)));
  coeff_pair v;
  v.x = a, v.y = b;
  __builtin_nontemporal_store(v, reinterpret_cast<coeff_pair*>(dst));
#else
  *dst = make_double2(a, b);
#endif
}

__device__ __forceinline__ void quad_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ double quad_sum(double v) {  // over the four lanes of a quad, to all of them
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  return v;
}

// WP: vertex positions from the compact waypoint array (MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS).  A template parameter, not a
// run-time stride: with the stride in a register the value-array path lost 13 % (65536 x 10: 129 -> 147 us, same box)
// ENDS: the two end vertices may leave derivative slots free (rest-to-rest paths under an objective order below snap: jerk and
// / or snap are unknowns there).  Such an end vertex is eliminated like an interior one -- its block is the near (far) part of
// its only segment with the identity in the rows and columns of the constrained slots and zero in place of their reciprocal
// pivots, so that they contribute nothing to W, z and x -- and has a record of its own: records for vertices 0 .. S instead
// of 1 .. S - 1.  A fully constrained end takes the plain path's step.  Without ENDS such paths take the general step.
// MOVING: a START vertex whose constrained derivatives carry non-zero values (a path that starts from a moving state) stays on
// the fast road -- the single-launch kernels; the grouped launch (the headline's kernel) is compiled without it: the lines
// below cost it 16 % (330 -> 300 M/s over 20 steps, 556 -> 469 over 200, same box) although no wavefront of it ever takes them
template <bool WP, bool ENDS = false, bool MOVING = false>
__device__ __forceinline__ void solve_quad_body(const BatchView& b, int d, const uint8_t* __restrict__ mask,
                                                const double* __restrict__ vals, const double* seg_times,
                                                double* __restrict__ coeffs, int32_t* __restrict__ status,
                                                double* __restrict__ cost, const int32_t* __restrict__ status_in, double* ws,
                                                const RowsTail& tail, int block, const double* __restrict__ pos_wp) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x, pl = lane >> 2, dim = lane & 3;
  const int q = block * kQdPaths + pl;
  const bool active = q < b.n_paths;
  const PathRef pr = path_at(b, active ? q : b.n_paths - 1);
  const int S = pr.S;
  const int Smax = b.max_segments;
  double* rec0 = lds;
  double* tbuf = lds + (size_t)(ENDS ? Smax + 1 : (Smax > 1 ? Smax - 1 : 1)) * kQdRec * kQdPaths;  // [segment][path]
  unsigned char* vmask = reinterpret_cast<unsigned char*>(tbuf + (size_t)Smax * kQdPaths);        // ENDS: [vertex][path] free masks
#if MRS_TG_QUAD_EXP == 5
  const int n_rec = Smax > 1 ? Smax - 1 : 1;
  int* pinfo = reinterpret_cast<int*>(tbuf + (size_t)Smax * kQdPaths);
  if (dim == 0) {
    pinfo[pl] = pr.s0;
    pinfo[kQdPaths + pl] = active ? S : 0;
  }
#endif
  // ---- prologue: times (scaled, for the last solve of a Mellinger pipeline), plainness of the path
  const bool scaling = tail.maxima != nullptr;
  double t_sum = 0.0;
  bool ok = S >= 2, pos_ok = true;
  unsigned fm_first = 0u, fm_last = 0u;  // ENDS: free slots of the two end vertices (bit r: derivative r + 1)
  bool moving_path = false;              // the start vertex's constrained derivatives carry non-zero values (lane 0 of the quad knows)
  if (active) {
    const int opt_st = scaling ? tail.opt_status[pr.p] : 0;
    for (int i = dim; i < S; i += 4) {
      double T = seg_times[pr.s0 + i];
      if (scaling) {  // scaleSegmentTimesToMeetConstraints (trajectory.cpp:625-657), then the solve at the scaled times
        if (opt_st != -2) T *= violation_scaling(tail.maxima + (size_t)(pr.s0 + i) * 9, tail.limits + (size_t)pr.p * 9);
        tail.seg_times_out[pr.s0 + i] = T;
      }
      tbuf[i * kQdPaths + pl] = T;
      t_sum += T;
    }
    for (int v = dim; v <= S; v += 4) {
      const uint8_t* mrow = mask + (size_t)(pr.v0 + v) * kHalf;
      const bool end = v == 0 || v == S;
      unsigned fixed = 0;
#pragma unroll
      for (int k = 0; k < kHalf; ++k) fixed |= (mrow[k] != 0 ? 1u : 0u) << k;
      pos_ok = pos_ok && (fixed & 1u);
      if (end) {
        const double* vrow = vals + (size_t)(pr.v0 + v) * kHalf * kD;
        double nz = 0.0;
        // (the values of unconstrained slots are not read by anybody: only those of constrained ones count.  Every value is
        // LOADED unconditionally, sixteen requests back to back -- written as a conditional expression around the load they
        // became four branches with a trip to memory each: 65536 x 10 132 -> 147 us, the grouped headline -16 %)
        double av[kHalf * kD];
#pragma unroll
        for (int e = kD; e < kHalf * kD; ++e) av[e] = fabs(vrow[e]);
#pragma unroll
        for (int k = 1; k < kHalf; ++k)
#pragma unroll
          for (int e = 0; e < kD; ++e) nz += ((fixed >> k) & 1u) ? av[k * kD + e] : 0.0;
        // a START vertex whose constrained derivatives carry non-zero values (a path that starts from a moving state) stays on
        // this road: the values are right-hand-side terms of the first step (below); the end vertex must be at rest
        if (MOVING && v == 0) {
          moving_path = nz != 0.0;
          nz = 0.0;
        }
        if (ENDS) {
          ok = ok && (fixed & 1u) && nz == 0.0;
          if (v == 0) fm_first = (~fixed >> 1) & 0xFu;
          if (v == S) fm_last = (~fixed >> 1) & 0xFu;
        } else {
          ok = ok && fixed == 0x1Fu && nz == 0.0;
        }
      } else if (ENDS) {
        // an interior vertex may hold slots constrained to ZERO (a stop_at vertex: velocity = acceleration = jerk = 0): it is
        // eliminated with the identity in those slots; its values are looked at only when it has such slots
        bool zero = true;
        if (fixed != 0x1u) {
          const double* vrow = vals + (size_t)(pr.v0 + v) * kHalf * kD;
          double av[kHalf * kD];
#pragma unroll
          for (int e = kD; e < kHalf * kD; ++e) av[e] = fabs(vrow[e]);
          double nz = 0.0;
#pragma unroll
          for (int k = 1; k < kHalf; ++k)
#pragma unroll
            for (int e = 0; e < kD; ++e) nz += ((fixed >> k) & 1u) ? av[k * kD + e] : 0.0;
          zero = nz == 0.0;
        }
        ok = ok && (fixed & 1u) && zero;
      } else {
        ok = ok && fixed == 0x1u;
      }
      if (ENDS) vmask[v * kQdPaths + pl] = (unsigned char)((~fixed >> 1) & 0xFu);
    }
  }
  if (ENDS) {  // vertex 0 was read by lane 0 of the quad, vertex S by lane S mod 4
    fm_first = (unsigned)__shfl((int)fm_first, lane & ~3, 64);
    fm_last = (unsigned)__shfl((int)fm_last, (lane & ~3) + (S & 3), 64);
  }
  t_sum = quad_sum(t_sum);
  const bool any_moving = MOVING && __ballot(active && moving_path) != 0ull;  // (wave-uniform: a wavefront of paths at rest skips every line of it)
  if (any_moving) moving_path = __shfl((int)moving_path, lane & ~3, 64) != 0;
  const bool plain_wave = __ballot(active && !ok) == 0ull;
  const unsigned long long pos_bad = __ballot(active && !pos_ok);
  const bool path_pos_ok = ((pos_bad >> (lane & ~3)) & 0xFull) == 0ull;
  quad_wave_sync();

  double my_cost = 0.0;
  if (plain_wave) {
    // ---- the exact unit-time constants of this objective order (uniform: scalar loads)
    const double (*hb)[kN] = c_hbar[d];
    double cNear[10], cCpl[kNB][kNB], cFar[10], cN[kNB], cF[kNB];
#pragma unroll
    for (int r = 0; r < kNB; ++r) {
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        cNear[tri(r, c)] = hb[kSlot0 + r][kSlot0 + c];
        cFar[tri(r, c)] = hb[kHalf + kSlot0 + r][kHalf + kSlot0 + c];
      }
#pragma unroll
      for (int c = 0; c < kNB; ++c) cCpl[r][c] = hb[kSlot0 + r][kHalf + kSlot0 + c];
      cN[r] = hb[kSlot0 + r][0];
      cF[r] = hb[kHalf + kSlot0 + r][0];
    }
    const bool d1 = (d & 1) != 0, d2 = (d & 2) != 0, d4 = (d & 4) != 0;
    const int Sw = __builtin_amdgcn_readfirstlane(S);  // (a wavefront's paths may differ in length: ragged batches)
    int Smx = Sw;
    {  // longest path of the wavefront
      int s = active ? S : 0;
      for (int off = 32; off >= 4; off >>= 1) s = max(s, __shfl_xor(s, off, 64));
      Smx = __builtin_amdgcn_readfirstlane(s);
    }
    // position of vertex v, this dimension: pv[v * pstride].  From the caller's value array that is 8 bytes out of every 160
    // (and whole cache lines come along: 137 MB read for 45 MB of inputs at 65536 x 10); under
    // MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS from the compact [vertex][4] waypoint array, whose every byte is used
    const double* pv = WP ? pos_wp + (size_t)pr.v0 * kD + dim : vals + (size_t)pr.v0 * kHalf * kD + dim;
    constexpr size_t pstride = WP ? (size_t)kD : (size_t)(kHalf * kD);
    // ---- forward: block Cholesky over the vertex chain
    double Sm[10], y[kNB];
    // positions are requested three segments ahead: with less than one wavefront per SIMD nothing else hides a trip to memory,
    // and a segment step is shorter than one
    auto pos = [&](int v) { return (active && v <= S) ? pv[(size_t)v * pstride] : 0.0; };
    double p_cur = pos(0), p_nxt = pos(1), p_a2 = pos(2), p_a3 = pos(3);
    for (int i = 0; i < Smx; ++i) {
      const bool on = active && i < S;
      const double T = on ? tbuf[i * kQdPaths + pl] : 1.0;
      const double p_a4 = pos(i + 4);
      double p2[9];
      quad_powers(T, d1, d2, d4, p2);
      const double dp = p_cur - p_nxt;
      // a moving start: this dimension's constrained derivative values of the start vertex (zero in free slots), from the
      // caller's value array -- read here and again for the first segment's coefficients, not kept in between
      double f0[kNB] = {0.0, 0.0, 0.0, 0.0};
      if (any_moving && i == 0 && active && moving_path) {
        const double* vrow0 = vals + (size_t)pr.v0 * kHalf * kD + dim;
#pragma unroll
        for (int r = 0; r < kNB; ++r) f0[r] = vrow0[(r + 1) * kD];
#pragma unroll
        for (int r = 0; r < kNB; ++r) f0[r] = (!ENDS || !((fm_first >> r) & 1u)) ? f0[r] : 0.0;
      }
      if (on) {
        // ENDS: the free mask of the vertex this step eliminates: the start vertex's (0: fully constrained, nothing to
        // eliminate) or vertex i's (0xF unless it is a stop_at vertex)
        const unsigned fmv = !ENDS ? 0xFu : (i == 0 ? fm_first : (unsigned)vmask[i * kQdPaths + pl]);
        const bool free_start = ENDS && i == 0 && fm_first != 0u;
        const bool masked_step = ENDS && (i == 0 ? fm_first != 0u : fmv != 0xFu);
        if (i == 0 && !free_start) {  // the start vertex is fully constrained: the state moves to vertex 1
#pragma unroll
          for (int r = 0; r < kNB; ++r) {
#pragma unroll
            for (int c = 0; c <= r; ++c) Sm[tri(r, c)] = cFar[tri(r, c)] * p2[r + c + 2];
            y[r] = -((cF[r] * p2[r + 1]) * dp);
          }
        } else {
          // vertex i: its block and right-hand side are complete with this segment's near part
          // (a start vertex with free slots: the near part IS its block -- the identity in the constrained slots)
#pragma unroll
          for (int r = 0; r < kNB; ++r) {
#pragma unroll
            for (int c = 0; c <= r; ++c)
              Sm[tri(r, c)] = free_start ? cNear[tri(r, c)] * p2[r + c + 2] : fma(cNear[tri(r, c)], p2[r + c + 2], Sm[tri(r, c)]);
            y[r] = free_start ? -((cN[r] * p2[r + 1]) * dp) : fma(-(cN[r] * p2[r + 1]), dp, y[r]);
          }
          if (ENDS && free_start && any_moving) {  // the start vertex's own unknowns see its constrained values
#pragma unroll
            for (int r = 0; r < kNB; ++r)
#pragma unroll
              for (int c = 0; c < kNB; ++c)
                y[r] = fma(-(cNear[r >= c ? tri(r, c) : tri(c, r)] * p2[r + c + 2]), f0[c], y[r]);
          }
          double rs[kNB];  // ENDS: 0 in place of the reciprocal pivot of a constrained slot
#pragma unroll
          for (int r = 0; r < kNB; ++r) rs[r] = (masked_step && !((fmv >> r) & 1u)) ? 0.0 : 1.0;
          if (masked_step) {
#pragma unroll
            for (int r = 0; r < kNB; ++r)
#pragma unroll
              for (int c = 0; c <= r; ++c)
                if (!((fmv >> r) & 1u) || !((fmv >> c) & 1u)) Sm[tri(r, c)] = (r == c) ? 1.0 : 0.0;
          }
          double L[10], Linv[kNB], z[kNB];
#pragma unroll
          for (int c = 0; c < kNB; ++c) {
            double dsum = Sm[tri(c, c)];
#pragma unroll
            for (int m = 0; m < c; ++m) dsum = fma(-L[tri(c, m)], L[tri(c, m)], dsum);
            const double inv = rsqrt_refined(dsum);
            Linv[c] = ENDS ? inv * rs[c] : inv;
#pragma unroll
            for (int r = c + 1; r < kNB; ++r) {
              double t = Sm[tri(r, c)];
#pragma unroll
              for (int m = 0; m < c; ++m) t = fma(-L[tri(r, m)], L[tri(c, m)], t);
              L[tri(r, c)] = t * inv;
            }
          }
#pragma unroll
          for (int r = 0; r < kNB; ++r) {
            double t = y[r];
#pragma unroll
            for (int m = 0; m < r; ++m) t = fma(-L[tri(r, m)], z[m], t);
            z[r] = t * Linv[r];
          }
          double* rec = rec0 + (size_t)(ENDS ? i : i - 1) * kQdRec * kQdPaths + pl;
          if (dim == 0) {
            rec[(kQdL + 0) * kQdPaths] = L[tri(1, 0)];
            rec[(kQdL + 1) * kQdPaths] = L[tri(2, 0)];
            rec[(kQdL + 2) * kQdPaths] = L[tri(2, 1)];
            rec[(kQdL + 3) * kQdPaths] = L[tri(3, 0)];
            rec[(kQdL + 4) * kQdPaths] = L[tri(3, 1)];
            rec[(kQdL + 5) * kQdPaths] = L[tri(3, 2)];
#pragma unroll
            for (int r = 0; r < kNB; ++r) rec[(kQdLinv + r) * kQdPaths] = Linv[r];
          }
#pragma unroll
          for (int r = 0; r < kNB; ++r) rec[(kQdZ + r * kD + dim) * kQdPaths] = z[r];
          // W = L^-1 E, then the Schur complement and right-hand side of vertex i + 1 (the last vertex: only when it has unknowns)
          if (i < S - 1 || (ENDS && fm_last != 0u)) {
            double W[kNB][kNB];
#pragma unroll
            for (int c = 0; c < kNB; ++c)
#pragma unroll
              for (int r = 0; r < kNB; ++r) {
                double t = cCpl[r][c] * p2[r + c + 2];
#pragma unroll
                for (int m = 0; m < r; ++m) t = fma(-L[tri(r, m)], W[m][c], t);
                W[r][c] = t * Linv[r];
              }
#pragma unroll
            for (int r = 0; r < kNB; ++r) {
#pragma unroll
              for (int c = 0; c <= r; ++c) {
                double t = cFar[tri(r, c)] * p2[r + c + 2];
#pragma unroll
                for (int m = 0; m < kNB; ++m) t = fma(-W[m][r], W[m][c], t);
                Sm[tri(r, c)] = t;
              }
              double t = -((cF[r] * p2[r + 1]) * dp);
#pragma unroll
              for (int m = 0; m < kNB; ++m) t = fma(-W[m][r], z[m], t);
              y[r] = t;
            }
          }
        }
        if (any_moving && i == 0) {  // vertex 1's right-hand side: - sum_c E[c][r] T^(r+c+2-2d) f_c
#pragma unroll
          for (int r = 0; r < kNB; ++r)
#pragma unroll
            for (int c = 0; c < kNB; ++c) y[r] = fma(-(cCpl[c][r] * p2[r + c + 2]), f0[c], y[r]);
        }
        if (ENDS && i == S - 1 && fm_last != 0u) {
          // the end vertex has unknowns: its block (the far part of the last segment, less what vertex S - 1 took) is complete
#pragma unroll
          for (int r = 0; r < kNB; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c)
              if (!((fm_last >> r) & 1u) || !((fm_last >> c) & 1u)) Sm[tri(r, c)] = (r == c) ? 1.0 : 0.0;
          double L[10], Linv[kNB], z[kNB];
#pragma unroll
          for (int c = 0; c < kNB; ++c) {
            double dsum = Sm[tri(c, c)];
#pragma unroll
            for (int m = 0; m < c; ++m) dsum = fma(-L[tri(c, m)], L[tri(c, m)], dsum);
            const double inv = rsqrt_refined(dsum);
            Linv[c] = ((fm_last >> c) & 1u) ? inv : 0.0;
#pragma unroll
            for (int r = c + 1; r < kNB; ++r) {
              double t = Sm[tri(r, c)];
#pragma unroll
              for (int m = 0; m < c; ++m) t = fma(-L[tri(r, m)], L[tri(c, m)], t);
              L[tri(r, c)] = t * inv;
            }
          }
#pragma unroll
          for (int r = 0; r < kNB; ++r) {
            double t = y[r];
#pragma unroll
            for (int m = 0; m < r; ++m) t = fma(-L[tri(r, m)], z[m], t);
            z[r] = t * Linv[r];
          }
          double* rec = rec0 + (size_t)S * kQdRec * kQdPaths + pl;
          if (dim == 0) {
            rec[(kQdL + 0) * kQdPaths] = L[tri(1, 0)];
            rec[(kQdL + 1) * kQdPaths] = L[tri(2, 0)];
            rec[(kQdL + 2) * kQdPaths] = L[tri(2, 1)];
            rec[(kQdL + 3) * kQdPaths] = L[tri(3, 0)];
            rec[(kQdL + 4) * kQdPaths] = L[tri(3, 1)];
            rec[(kQdL + 5) * kQdPaths] = L[tri(3, 2)];
#pragma unroll
            for (int r = 0; r < kNB; ++r) rec[(kQdLinv + r) * kQdPaths] = Linv[r];
          }
#pragma unroll
          for (int r = 0; r < kNB; ++r) rec[(kQdZ + r * kD + dim) * kQdPaths] = z[r];
        }
      }
      p_cur = p_nxt;
      p_nxt = p_a2;
      p_a2 = p_a3;
      p_a3 = p_a4;
    }
    quad_wave_sync();  // (lane 0 of a quad wrote L and W for the other three)
    // ---- backward: x_v = L^-T (z - W x_{v+1}); coefficients and cost of segment v from x_v, x_{v+1}
    double xn[kNB] = {0.0, 0.0, 0.0, 0.0};  // the last vertex is fully constrained (ENDS: or solved below, when it has unknowns)
    // (positions again, requested three vertices ahead of their use; a path shorter than the wavefront's longest joins late.
    // Keeping a copy of the positions in LDS instead -- no global load in this loop, so no s_waitcnt vmcnt(0) in front of
    // every step's stores -- was measured in round 5: no change at 65536 x 10, and the 5.6 KB it adds per wavefront cost
    // the grouped launch its fifth wavefront per CU: HISTORY.md)
    auto posb = [&](int v) { return (active && v >= 0 && v <= S) ? pv[(size_t)v * pstride] : 0.0; };
    double p_end = posb(S), p_b0 = posb(Smx - 1), p_b1 = posb(Smx - 2), p_b2 = posb(Smx - 3);
    for (int v = Smx - 1; v >= 0; --v) {
      const bool on = active && v < S;
      double x[kNB] = {0.0, 0.0, 0.0, 0.0};
      const double p_start = p_b0;
      p_b0 = p_b1;
      p_b1 = p_b2;
      p_b2 = posb(v - 3);
      if (ENDS && active && v == S - 1 && fm_last != 0u) {  // x_S = L^-T z of the end vertex (zero in its constrained slots)
        const double* rec = rec0 + (size_t)S * kQdRec * kQdPaths + pl;
        const double l10 = rec[(kQdL + 0) * kQdPaths], l20 = rec[(kQdL + 1) * kQdPaths], l21 = rec[(kQdL + 2) * kQdPaths],
                     l30 = rec[(kQdL + 3) * kQdPaths], l31 = rec[(kQdL + 4) * kQdPaths], l32 = rec[(kQdL + 5) * kQdPaths];
        const double i0 = rec[(kQdLinv + 0) * kQdPaths], i1 = rec[(kQdLinv + 1) * kQdPaths], i2 = rec[(kQdLinv + 2) * kQdPaths],
                     i3 = rec[(kQdLinv + 3) * kQdPaths];
        double t[kNB];
#pragma unroll
        for (int r = 0; r < kNB; ++r) t[r] = rec[(kQdZ + r * kD + dim) * kQdPaths];
        xn[3] = t[3] * i3;
        xn[2] = fma(-l32, xn[3], t[2]) * i2;
        xn[1] = fma(-l31, xn[3], fma(-l21, xn[2], t[1])) * i1;
        xn[0] = fma(-l30, xn[3], fma(-l20, xn[2], fma(-l10, xn[1], t[0]))) * i0;
      }
      if (on && (v >= 1 || (ENDS && fm_first != 0u))) {
        const double* rec = rec0 + (size_t)(ENDS ? v : v - 1) * kQdRec * kQdPaths + pl;
        double t[kNB];
#pragma unroll
        for (int r = 0; r < kNB; ++r) t[r] = rec[(kQdZ + r * kD + dim) * kQdPaths];
        const double l10 = rec[(kQdL + 0) * kQdPaths], l20 = rec[(kQdL + 1) * kQdPaths], l21 = rec[(kQdL + 2) * kQdPaths],
                     l30 = rec[(kQdL + 3) * kQdPaths], l31 = rec[(kQdL + 4) * kQdPaths], l32 = rec[(kQdL + 5) * kQdPaths];
        const double i0 = rec[(kQdLinv + 0) * kQdPaths], i1 = rec[(kQdLinv + 1) * kQdPaths], i2 = rec[(kQdLinv + 2) * kQdPaths],
                     i3 = rec[(kQdLinv + 3) * kQdPaths];
        // t = z - W x_{v+1},  W x = L^-1 (E x),  E[r][c] = HBAR[1+r][6+c] T_v^(r+c+2-2d)  (x_S = 0 unless the end vertex has unknowns)
        if (v < S - 1 || (ENDS && fm_last != 0u)) {
          double pw[9];
          quad_powers(tbuf[v * kQdPaths + pl], d1, d2, d4, pw);
          double u[kNB];
#pragma unroll
          for (int r = 0; r < kNB; ++r) {
            u[r] = 0.0;
#pragma unroll
            for (int c = 0; c < kNB; ++c) u[r] = fma(cCpl[r][c] * pw[r + c + 2], xn[c], u[r]);
          }
          const double w0 = u[0] * i0;
          const double w1 = fma(-l10, w0, u[1]) * i1;
          const double w2 = fma(-l21, w1, fma(-l20, w0, u[2])) * i2;
          const double w3 = fma(-l32, w2, fma(-l31, w1, fma(-l30, w0, u[3]))) * i3;
          t[0] -= w0;
          t[1] -= w1;
          t[2] -= w2;
          t[3] -= w3;
        }
        x[3] = t[3] * i3;
        x[2] = fma(-l32, x[3], t[2]) * i2;
        x[1] = fma(-l31, x[3], fma(-l21, x[2], t[1])) * i1;
        x[0] = fma(-l30, x[3], fma(-l20, x[2], fma(-l10, x[1], t[0]))) * i0;
      }
      if (any_moving && v == 0 && active && moving_path) {  // the start vertex's constrained derivative values
        const double* vrow0 = vals + (size_t)pr.v0 * kHalf * kD + dim;
#pragma unroll
        for (int r = 0; r < kNB; ++r) {
          const double fr = vrow0[(r + 1) * kD];
          x[r] += (!ENDS || !((fm_first >> r) & 1u)) ? fr : 0.0;
        }
      }
      if (on) {
        const double T = tbuf[v * kQdPaths + pl];
        // c_k = T^-k sum_j ABAR_INV[k][j] T^(j%5) d_j ; cb_k = c_k T^k is what the cost needs
        const double dv[kN] = {p_start, x[0], x[1], x[2], x[3], p_end, xn[0], xn[1], xn[2], xn[3]};
        double w[kHalf];
        w[0] = 1.0;
#pragma unroll
        for (int k = 1; k < kHalf; ++k) w[k] = w[k - 1] * T;
        double db[kN], cb[kN], c[kN];
#pragma unroll
        for (int j = 0; j < kN; ++j) db[j] = dv[j] * w[j % kHalf];
        const double ti = 1.0 / T;
        double tik = 1.0;
#pragma unroll
        for (int k = 0; k < kN; ++k) {
          double s = 0.0;
          if (k < kHalf) {
            s = c_abar_inv[k][k] * db[k];  // the upper half of ABAR_INV is diag(1/k!)
          } else {
#pragma unroll
            for (int j = 0; j < kN; ++j) s += c_abar_inv[k][j] * db[j];
          }
          cb[k] = s;
          c[k] = s * tik;
          tik *= ti;
        }
        double p2[9];
        quad_powers(T, d1, d2, d4, p2);  // p2[0] = T^(1 - 2d)
        my_cost = fma(cost_quadratic_form_d(d, cb), p2[0], my_cost);
        // the coefficients leave as five 16-byte pieces per lane (a lane owns 80 consecutive bytes).  Staging them through
        // the LDS of the consumed vertex records so that 16 lanes store 256 consecutive bytes of one path was built and
        // measured in round 5: 65536 x 10 129 -> 156 us -- the LDS round trip and its two wavefront syncs per step cost a
        // lone wavefront more than the 4x fewer write requests return (HISTORY.md)
        double2* out = reinterpret_cast<double2*>(coeffs + ((size_t)(pr.s0 + v) * kD + dim) * kN);
#if MRS_TG_QUAD_EXP == 5   // experiment: staged through the LDS of the consumed vertex records (256 B of one path per 16 lanes)
        (void)out;
        if ((v > 0 ? v - 1 : 0) + 2 <= n_rec) {
          double2* st = reinterpret_cast<double2*>(rec0 + (size_t)(v > 0 ? v - 1 : 0) * kQdRec * kQdPaths + pl * (kN * kD) + dim * kN);
#pragma unroll
          for (int k = 0; k < kN; k += 2) st[k / 2] = make_double2(c[k], c[k + 1]);
        } else {
          double2* o5 = reinterpret_cast<double2*>(coeffs + ((size_t)(pr.s0 + v) * kD + dim) * kN);
#pragma unroll
          for (int k = 0; k < kN; k += 2) o5[k / 2] = make_double2(c[k], c[k + 1]);
        }
#elif MRS_TG_QUAD_EXP == 1   // experiment: the coefficients are computed and not stored
#pragma unroll
        for (int k = 0; k < kN; ++k) asm volatile("" ::"v"(c[k]));
        (void)out;
#elif MRS_TG_QUAD_EXP == 3  // experiment: the same pieces into a 5 KB window per wavefront (L2 hits, no HBM write traffic)
        double2* o3 = reinterpret_cast<double2*>(coeffs + (size_t)block * 640 + lane * 10);
#pragma unroll
        for (int k = 0; k < kN; k += 2) o3[k / 2] = make_double2(c[k], c[k + 1]);
        (void)out;
#elif MRS_TG_QUAD_EXP == 4  // experiment: perfectly coalesced (wrong) layout: instruction k writes 64 consecutive 16-byte pieces
        double2* o4 = reinterpret_cast<double2*>(coeffs + ((size_t)block * Smax + v) * 640);
#pragma unroll
        for (int k = 0; k < kN; k += 2) o4[(k / 2) * 64 + lane] = make_double2(c[k], c[k + 1]);
        (void)out;
#elif MRS_TG_QUAD_EXP == 2  // experiment: streaming stores
#pragma unroll
        for (int k = 0; k < kN; k += 2) {
          __builtin_nontemporal_store(c[k], reinterpret_cast<double*>(out) + k);
          __builtin_nontemporal_store(c[k + 1], reinterpret_cast<double*>(out) + k + 1);
        }
#else
#pragma unroll
        for (int k = 0; k < kN; k += 2) store_coeff_pair(out + k / 2, c[k], c[k + 1]);
#endif
      }
#if MRS_TG_QUAD_EXP == 5
      if ((v > 0 ? v - 1 : 0) + 2 <= n_rec) {  // (uniform)
        const double* stage = rec0 + (size_t)(v > 0 ? v - 1 : 0) * kQdRec * kQdPaths;
        quad_wave_sync();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int P = 4 * g + (lane >> 4), piece = lane & 15;
          const int s0P = pinfo[P], SP = pinfo[kQdPaths + P];
          if (v < SP) {
            const double2 val = *reinterpret_cast<const double2*>(stage + P * (kN * kD) + piece * 2);
            *reinterpret_cast<double2*>(coeffs + ((size_t)(s0P + v) * (kN * kD) + piece * 2)) = val;
          }
        }
        if (on) {
          const double2 val = *reinterpret_cast<const double2*>(stage + pl * (kN * kD) + 32 + dim * 2);
          *reinterpret_cast<double2*>(coeffs + ((size_t)(pr.s0 + v) * (kN * kD) + 32 + dim * 2)) = val;
        }
      }
#endif
#pragma unroll
      for (int r = 0; r < kNB; ++r) xn[r] = x[r];
      if (on) p_end = p_start;
    }
  } else if (active) {
    // ---- any other constraint pattern: the general masked step, factors in the plan's global workspace
    bool pok = true;
    BlockSource none{nullptr, nullptr, 0, 0};
    // (the times of this path, contiguous: the scaled ones were written to seg_times_out above, by this quad)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    const double* times = (scaling ? tail.seg_times_out : seg_times) + pr.s0;
    my_cost = solve_path<1, true>(mask, vals, pr.v0, S, d, times, dim, none, ws, (size_t)b.n_paths * 4,
                                  (unsigned)q * 4u + (unsigned)dim, coeffs + (size_t)pr.s0 * kD * kN, pok);
  }
  my_cost = quad_sum(my_cost);
  if (active && dim == 0) {
    if (cost) cost[pr.p] = my_cost;
    if (status) {
      int st = merge_status(path_pos_ok, status_in, pr.p);
      if (tail.sum_t0 != nullptr && st > 0 && t_sum > MRS_TG_RUNAWAY_TIME_FACTOR * tail.sum_t0[pr.p]) st = MRS_TG_STATUS_ROUNDOFF_LIMITED;
      status[pr.p] = st;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The TWO-SIDED variant for launches that leave SIMDs idle (round 6).  A dispatch of 10240 paths is 640 quad wavefronts for
// 1024 SIMDs, each walking the chain of S - 1 vertices forward and back with its latencies exposed (VALU issue 19.7 %, wait
// share 45 %: profiles/round5_pmc_sq_solve_quad_group.json).  Here EIGHT lanes own a path -- side x dimension -- and a
// wavefront holds 8 paths: side 0 eliminates vertices 1 .. m - 1 from the start, side 1 eliminates S - 1 .. m + 1 from the
// end, the two meet at the middle vertex m = (S + 1) / 2: twice the wavefronts, half the dependent chain, the same LDS per CU.
//
// Side 1 runs the SAME instructions on the time-reversed path.  Reversing time maps a min-derivative problem onto itself:
// p~(t) = p(T - t) has derivatives (-1)^k p^(k) at the vertices, the cost is unchanged, and the unit-time constants are those
// of the forward problem.  So side 1 sees local vertex v = original vertex S - v, local segment i = original segment
// S - 1 - i, and its unknowns are the original ones with the odd derivatives negated (x~ = D x, D = diag(-1, +1, -1, +1) over
// the slots velocity .. snap).  At the join each side adds the other's Schur complement and right-hand side of the middle
// vertex -- D S D and D y, exact sign flips -- and both solve the same 4 x 4 system (bit-identical up to those signs); then
// each substitutes back through its own half and forms the coefficients of its own segments in FORWARD orientation from the
// end-point derivatives (side 1: start and end swapped, odd derivatives negated back).
// Plain paths only (position-only interior vertices, fully constrained ends; MOVING: a start vertex in motion, side 0's first
// step as in solve_quad_body); a wavefront with any other path takes the general masked step on the lanes of side 0.
// The results differ from solve_quad_body's in the last bits (another elimination order): same tolerances against the
// oracle's 113-bit route (tests/test_gpu_headline_kernel.py).
// Memory: every global load in the prologue, in one round; the position constraints, the times and the paths' indices wait in
// LDS; no register is carried across the loops that the compiler would spill ("What bounds the two-sided
// kernel": loads and stores share one in-order counter on gfx950, so a load or a scratch reload in the loops waits for the
// coefficient stores in flight).  Min-snap only (the launchers send nothing else; the ABI sends paths of up to 15 segments).
constexpr int kDuoPaths = 8;  // paths per wavefront

__host__ __device__ constexpr size_t duo_lds_doubles(int Smax) {
  return (size_t)(Smax > 1 ? Smax - 1 : 1) * kQdRec * kDuoPaths + (size_t)Smax * kDuoPaths +  // records | times
         (size_t)(Smax + 1) * kD * kDuoPaths + kDuoPaths / 2;                                   // | position constraints | path indices
}

#ifdef MRS_TG_DUO_STAMPS  // experiment builds (build.py --variant): the shader clock at the phase boundaries of each wavefront
__device__ unsigned long long g_duo_stamps[2048 * 16];
#define DUO_STAMP(k) (duo_stamp[k] = __builtin_readcyclecounter())
#else
#define DUO_STAMP(k)
#endif

// The lane's index, formed where it is needed: under a register budget of three wavefronts per SIMD the compiler spills the
// values it derived from threadIdx.x in the prologue (the path's slot, the side, the dimension, LDS byte offsets) to scratch and
// reloads them inside the elimination loops -- vector-memory loads, which on gfx950 wait for the coefficient stores in flight
// (one counter for both).  Two instructions here instead; valid while all 64 lanes are enabled (wave-uniform control flow).
// (TAG: two statements with different tags are different instructions to the compiler, which therefore cannot merge the tails
// of two branches that end in them -- see duo_finish)
template <int TAG = 0>
__device__ __forceinline__ int lane_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0 ; %1" : "=v"(l) : "i"(TAG));
  return l;
}

// the value of lane ^ 4 (the other side of the same path and dimension): ds_swizzle in bit-mask mode, no address register
__device__ __forceinline__ double xor4(double v) {
  constexpr int kSwap4 = (4 << 10) | 0x1F;  // and 0x1f, or 0, xor 4
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)u, kSwap4);
  const unsigned hi = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)(u >> 32), kSwap4);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// The epilogue of solve_duo_body: cost and status of the eight paths.  Written once per branch (TAG) instead of once behind
// both: behind the merge the compiler has to assume the general step's loads in flight and waits -- on gfx950 for all but the
// last few of the plain path's coefficient STORES as well (one counter), a store's round trip before the last two stores go out.
template <int TAG>
__device__ __forceinline__ void duo_finish(double my_cost, bool active, unsigned long long pos_bad, const int* pidx,
                                           double* __restrict__ cost, int32_t* __restrict__ status,
                                           const int32_t* __restrict__ status_in, const RowsTail& tail, double t_sum) {
  my_cost = quad_sum(my_cost);
  my_cost += xor4(my_cost);
  const int lane_late = lane_now<TAG + 1>();  // (the slot's address is formed HERE, not carried -- and spilled -- from the prologue)
  if (active && (lane_late & 7) == 0) {
    const bool path_pos_ok = ((pos_bad >> (lane_late & ~7)) & 0xFFull) == 0ull;
    const int p_out = pidx[lane_late >> 3];
    if (cost) cost[p_out] = my_cost;
    if (status) {
      int st = merge_status(path_pos_ok, status_in, p_out);
      if (tail.sum_t0 != nullptr && st > 0 && t_sum > MRS_TG_RUNAWAY_TIME_FACTOR * tail.sum_t0[p_out]) st = MRS_TG_STATUS_ROUNDOFF_LIMITED;
      status[p_out] = st;
      asm volatile("; epilogue %0" ::"i"(TAG));  // (keeps the two copies' tails apart: see above)
    }
  }
}

template <bool WP, bool MOVING = false>
__device__ __forceinline__ void solve_duo_body(const BatchView& b, int d, const uint8_t* __restrict__ mask,
                                               const double* __restrict__ vals, const double* seg_times,
                                               double* __restrict__ coeffs, int32_t* __restrict__ status,
                                               double* __restrict__ cost, const int32_t* __restrict__ status_in, double* ws,
                                               const RowsTail& tail, int block, const double* __restrict__ pos_wp) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x, pl = lane >> 3, side = (lane >> 2) & 1, dim = lane & 3, l8 = lane & 7;
  const int q = block * kDuoPaths + pl;
  const bool active = q < b.n_paths;
  const PathRef pr = path_at(b, active ? q : b.n_paths - 1);
  const int S = pr.S;
  const int Smax = b.max_segments;
  double* rec0 = lds;
  double* tbuf = lds + (size_t)(Smax > 1 ? Smax - 1 : 1) * kQdRec * kDuoPaths;  // [segment][path]
#ifdef MRS_TG_DUO_STAMPS
  unsigned long long duo_stamp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  DUO_STAMP(0);
  // ---- prologue: times (scaled, for the last solve of a Mellinger pipeline), plainness of the path, the position constraints
  // into LDS.  Every global load of the kernel is issued HERE, in one round (branch-free, indices clamped into the path): on
  // gfx950 loads and stores share one in-order counter (vmcnt), so a load inside the elimination loops makes each iteration
  // wait until the previous iteration's coefficient stores are acknowledged -- measured with the phase clocks of
  // -DMRS_TG_DUO_STAMPS: 3500 clocks per backward step, of which the arithmetic is a third (HISTORY.md, round 6).
  const bool scaling = tail.maxima != nullptr;
  double* pbuf = tbuf + (size_t)Smax * kDuoPaths;  // [vertex][dimension][path]
  int* pidx = reinterpret_cast<int*>(pbuf + (size_t)(Smax + 1) * kD * kDuoPaths);  // the paths' indices, for the epilogue: a register
  if (l8 == 0) pidx[pl] = pr.p;                                                     // held that long is spilled to scratch, and the
                                                                                    // reload is a vector-memory load (see above)
  DUO_STAMP(1);
  const double* pbase = WP ? pos_wp + (size_t)pr.v0 * kD : vals + (size_t)pr.v0 * kHalf * kD;
  constexpr int pstride = WP ? kD : kHalf * kD;
  const int n_pos = (S + 1) * kD;
  double t_sum = 0.0;
  bool ok = S >= 2, pos_ok = true;
  bool moving_path = false;
  auto mask_bits = [&](int v) {
    const uint8_t* mrow = mask + (size_t)(pr.v0 + v) * kHalf;
    unsigned fixed = 0;
#pragma unroll
    for (int k = 0; k < kHalf; ++k) fixed |= (mrow[k] != 0 ? 1u : 0u) << k;
    return fixed;
  };
  // the end vertices: even lanes of the eight read the start vertex's mask and constrained values, odd lanes the end vertex's
  // (lanes 0 and 1 decide)
  const int v_end = (l8 & 1) ? S : 0;
  const unsigned f_end = mask_bits(v_end);
  double av[kHalf * kD];
  {
    const double* vrow = vals + (size_t)(pr.v0 + v_end) * kHalf * kD;
#pragma unroll
    for (int e = kD; e < kHalf * kD; ++e) av[e] = fabs(vrow[e]);
  }
  const int opt_st = (scaling && active) ? tail.opt_status[pr.p] : 0;
  auto trip = [&](int base) {               // sixteen segments and vertices of every path: two of each per lane
    const int ia = base + l8, ib = ia + 8;
    double Ta = seg_times[pr.s0 + min(ia, S - 1)], Tb = seg_times[pr.s0 + min(ib, S - 1)];
    const unsigned fa = mask_bits(min(ia, S)), fb = mask_bits(min(ib, S));
    double pe[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ic = min(base * kD + k * 8 + l8, n_pos - 1);
      pe[k] = pbase[(size_t)(ic >> 2) * pstride + (ic & 3)];
    }
#ifdef MRS_TG_DUO_STAMPS
    if (base == 0) DUO_STAMP(13);
#endif
    if (active) {
      if (scaling) {  // scaleSegmentTimesToMeetConstraints (trajectory.cpp:625-657), then the solve at the scaled times
        if (ia < S) {
          if (opt_st != -2) Ta *= violation_scaling(tail.maxima + (size_t)(pr.s0 + ia) * 9, tail.limits + (size_t)pr.p * 9);
          tail.seg_times_out[pr.s0 + ia] = Ta;
        }
        if (ib < S) {
          if (opt_st != -2) Tb *= violation_scaling(tail.maxima + (size_t)(pr.s0 + ib) * 9, tail.limits + (size_t)pr.p * 9);
          tail.seg_times_out[pr.s0 + ib] = Tb;
        }
      }
      if (ia < S) {
        tbuf[ia * kDuoPaths + pl] = Ta;
        t_sum += Ta;
      }
      if (ib < S) {
        tbuf[ib * kDuoPaths + pl] = Tb;
        t_sum += Tb;
      }
      if (ia >= 1 && ia < S) {  // interior vertices: the position, nothing else
        ok = ok && fa == 0x1u;
        pos_ok = pos_ok && (fa & 1u);
      }
      if (ib < S) {
        ok = ok && fb == 0x1u;
        pos_ok = pos_ok && (fb & 1u);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int idx = base * kD + k * 8 + l8;
        if (idx < n_pos) pbuf[idx * kDuoPaths + pl] = pe[k];
      }
    }
  };
  // (the first trip is straight-line code -- all there is for paths of up to 15 segments: inside a loop the compiler has to
  // assume loads of the previous trip in flight and waits for the end vertices' values before it issues the positions' loads)
  trip(0);
  for (int base = 16; base <= Smax; base += 16) trip(base);
  if (active && l8 < 2) {  // the end vertices: fully constrained, at rest (MOVING: the start vertex may be in motion)
    double nz = 0.0;
#pragma unroll
    for (int k = 1; k < kHalf; ++k)
#pragma unroll
      for (int e = 0; e < kD; ++e) nz += ((f_end >> k) & 1u) ? av[k * kD + e] : 0.0;
    if (MOVING && l8 == 0) {
      moving_path = nz != 0.0;
      nz = 0.0;
    }
    ok = ok && f_end == 0x1Fu && nz == 0.0;
    pos_ok = pos_ok && (f_end & 1u);
  }
  DUO_STAMP(2);
  t_sum = quad_sum(t_sum);
  t_sum += xor4(t_sum);
  const bool any_moving = MOVING && __ballot(active && moving_path) != 0ull;
  if (any_moving) moving_path = __shfl((int)moving_path, lane & ~7, 64) != 0 && side == 0;  // (vertex 0 was read by lane 0 of the eight)
  const bool plain_wave = __ballot(active && !ok) == 0ull;
  const unsigned long long pos_bad = __ballot(active && !pos_ok);
  quad_wave_sync();
  DUO_STAMP(3);

  double my_cost = 0.0;
  if (plain_wave) {
    // (min-snap only: the launchers send no other objective order here -- with the order a constant the unit-time blocks are
    // literals instead of scalar loads the first step waits for, and one cost form is compiled instead of five)
    constexpr int kOrder = kHalf - 1;
    const double (*hb)[kN] = c_hbar[kOrder];
    double cNear[10], cCpl[kNB][kNB], cFar[10], cN[kNB], cF[kNB];
#pragma unroll
    for (int r = 0; r < kNB; ++r) {
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        cNear[tri(r, c)] = hb[kSlot0 + r][kSlot0 + c];
        cFar[tri(r, c)] = hb[kHalf + kSlot0 + r][kHalf + kSlot0 + c];
      }
#pragma unroll
      for (int c = 0; c < kNB; ++c) cCpl[r][c] = hb[kSlot0 + r][kHalf + kSlot0 + c];
      cN[r] = hb[kSlot0 + r][0];
      cF[r] = hb[kHalf + kSlot0 + r][0];
    }
    constexpr bool d1 = (kOrder & 1) != 0, d2 = (kOrder & 2) != 0, d4 = (kOrder & 4) != 0;
    // this side's chain: local vertices 0 .. M (0: the path's end on this side, M: the middle vertex), local segments 0 .. M - 1
    const int m_mid = (S + 1) >> 1;
    // (wave-uniform trip count of both loops: the longest side the batch can hold -- a reduction over the wavefront's own paths
    // costs four cross-lane exchanges and their waits; ragged batches are sorted by length, a wavefront's paths differ by little)
    const int Mmx = (Smax + 1) >> 1;
    const int m_mid_ = m_mid;
    struct LaneIds { int pl, side, dim; };
    auto ids_now = [&]() {  // (see lane_now)
      const int l = lane_now();
      return LaneIds{l >> 3, (l >> 2) & 1, l & 3};
    };
    auto oseg = [&](int sd, int i) { return sd ? S - 1 - i : i; };   // original index of local segment i
    auto overt = [&](int sd, int v) { return sd ? S - v : v; };      // original index of local vertex v
    auto side_len = [&](int sd) { return sd ? S - m_mid_ : m_mid_; };
    auto pos = [&](const LaneIds& id, int v) {  // the position constraint of local vertex v (0 outside this side's chain)
      const int Ms = side_len(id.side);
      const int vc = min(max(v, 0), Ms);
      const double r = pbuf[(overt(id.side, vc) * kD + id.dim) * kDuoPaths + id.pl];
      return (active && v >= 0 && v <= Ms) ? r : 0.0;
    };
    // ---- forward: this side's block Cholesky towards the middle
    double Sm[10], y[kNB];
#pragma unroll
    for (int e = 0; e < 10; ++e) Sm[e] = 0.0;
#pragma unroll
    for (int r = 0; r < kNB; ++r) y[r] = 0.0;
    double p_cur = pos(ids_now(), 0);
    DUO_STAMP(4);
    for (int i = 0; i < Mmx; ++i) {
      const LaneIds id = ids_now();
      const bool on = active && i < side_len(id.side);
      const double T = on ? tbuf[oseg(id.side, i) * kDuoPaths + id.pl] : 1.0;
      const double p_nxt = pos(id, i + 1);
      double p2[9];
      quad_powers(T, d1, d2, d4, p2);
      const double dp = p_cur - p_nxt;
      double f0[kNB] = {0.0, 0.0, 0.0, 0.0};
      if (any_moving && i == 0 && active && moving_path) {  // (side 0 only: moving_path is false on side 1)
        const double* vrow0 = vals + (size_t)pr.v0 * kHalf * kD + id.dim;
#pragma unroll
        for (int r = 0; r < kNB; ++r) f0[r] = vrow0[(r + 1) * kD];
      }
      if (on) {
        if (i == 0) {  // the end vertex is fully constrained: the state moves to local vertex 1
#pragma unroll
          for (int r = 0; r < kNB; ++r) {
#pragma unroll
            for (int c = 0; c <= r; ++c) Sm[tri(r, c)] = cFar[tri(r, c)] * p2[r + c + 2];
            y[r] = -((cF[r] * p2[r + 1]) * dp);
          }
          if (any_moving) {  // local vertex 1's right-hand side: - sum_c E[c][r] T^(r+c+2-2d) f_c
#pragma unroll
            for (int r = 0; r < kNB; ++r)
#pragma unroll
              for (int c = 0; c < kNB; ++c) y[r] = fma(-(cCpl[c][r] * p2[r + c + 2]), f0[c], y[r]);
          }
        } else {
          // local vertex i: its block and right-hand side are complete with this segment's near part
#pragma unroll
          for (int r = 0; r < kNB; ++r) {
#pragma unroll
            for (int c = 0; c <= r; ++c) Sm[tri(r, c)] = fma(cNear[tri(r, c)], p2[r + c + 2], Sm[tri(r, c)]);
            y[r] = fma(-(cN[r] * p2[r + 1]), dp, y[r]);
          }
          double L[10], Linv[kNB], z[kNB];
#pragma unroll
          for (int c = 0; c < kNB; ++c) {
            double dsum = Sm[tri(c, c)];
#pragma unroll
            for (int mm = 0; mm < c; ++mm) dsum = fma(-L[tri(c, mm)], L[tri(c, mm)], dsum);
            const double inv = rsqrt_refined(dsum);
            Linv[c] = inv;
#pragma unroll
            for (int r = c + 1; r < kNB; ++r) {
              double t = Sm[tri(r, c)];
#pragma unroll
              for (int mm = 0; mm < c; ++mm) t = fma(-L[tri(r, mm)], L[tri(c, mm)], t);
              L[tri(r, c)] = t * inv;
            }
          }
#pragma unroll
          for (int r = 0; r < kNB; ++r) {
            double t = y[r];
#pragma unroll
            for (int mm = 0; mm < r; ++mm) t = fma(-L[tri(r, mm)], z[mm], t);
            z[r] = t * Linv[r];
          }
          double* rec = rec0 + (size_t)(overt(id.side, i) - 1) * kQdRec * kDuoPaths + id.pl;
          if (id.dim == 0) {
            rec[(kQdL + 0) * kDuoPaths] = L[tri(1, 0)];
            rec[(kQdL + 1) * kDuoPaths] = L[tri(2, 0)];
            rec[(kQdL + 2) * kDuoPaths] = L[tri(2, 1)];
            rec[(kQdL + 3) * kDuoPaths] = L[tri(3, 0)];
            rec[(kQdL + 4) * kDuoPaths] = L[tri(3, 1)];
            rec[(kQdL + 5) * kDuoPaths] = L[tri(3, 2)];
#pragma unroll
            for (int r = 0; r < kNB; ++r) rec[(kQdLinv + r) * kDuoPaths] = Linv[r];
          }
#pragma unroll
          for (int r = 0; r < kNB; ++r) rec[(kQdZ + r * kD + id.dim) * kDuoPaths] = z[r];
          // W = L^-1 E, then the Schur complement and right-hand side of local vertex i + 1 (always an unknown: the next one
          // of this side, or the middle vertex)
          double W[kNB][kNB];
#pragma unroll
          for (int c = 0; c < kNB; ++c)
#pragma unroll
            for (int r = 0; r < kNB; ++r) {
              double t = cCpl[r][c] * p2[r + c + 2];
#pragma unroll
              for (int mm = 0; mm < r; ++mm) t = fma(-L[tri(r, mm)], W[mm][c], t);
              W[r][c] = t * Linv[r];
            }
#pragma unroll
          for (int r = 0; r < kNB; ++r) {
#pragma unroll
            for (int c = 0; c <= r; ++c) {
              double t = cFar[tri(r, c)] * p2[r + c + 2];
#pragma unroll
              for (int mm = 0; mm < kNB; ++mm) t = fma(-W[mm][r], W[mm][c], t);
              Sm[tri(r, c)] = t;
            }
            double t = -((cF[r] * p2[r + 1]) * dp);
#pragma unroll
            for (int mm = 0; mm < kNB; ++mm) t = fma(-W[mm][r], z[mm], t);
            y[r] = t;
          }
        }
      }
      p_cur = p_nxt;
#ifdef MRS_TG_DUO_STAMPS
      if (i == 0) DUO_STAMP(5);
      if (i == 1) DUO_STAMP(6);
#endif
    }
    DUO_STAMP(7);
    // ---- the join: the middle vertex's block and right-hand side are this side's part plus the other side's, which arrives in
    // the other orientation: D S D and D y with D = diag(-1, +1, -1, +1) -- exact sign flips, so both sides solve the same
    // system and x~ = D x bit for bit
    double xn[kNB];
    {
#pragma unroll
      for (int r = 0; r < kNB; ++r) {
#pragma unroll
        for (int c = 0; c <= r; ++c) {
          const double other = xor4(Sm[tri(r, c)]);
          Sm[tri(r, c)] += ((r + c) & 1) ? -other : other;
        }
        const double oy = xor4(y[r]);
        y[r] += (r & 1) ? oy : -oy;   // slot r is derivative r + 1: odd derivatives (r = 0, 2) change sign
      }
      double L[10], Linv[kNB], z[kNB];
#pragma unroll
      for (int c = 0; c < kNB; ++c) {
        double dsum = Sm[tri(c, c)];
#pragma unroll
        for (int mm = 0; mm < c; ++mm) dsum = fma(-L[tri(c, mm)], L[tri(c, mm)], dsum);
        const double inv = rsqrt_refined(dsum);
        Linv[c] = inv;
#pragma unroll
        for (int r = c + 1; r < kNB; ++r) {
          double t = Sm[tri(r, c)];
#pragma unroll
          for (int mm = 0; mm < c; ++mm) t = fma(-L[tri(r, mm)], L[tri(c, mm)], t);
          L[tri(r, c)] = t * inv;
        }
      }
#pragma unroll
      for (int r = 0; r < kNB; ++r) {
        double t = y[r];
#pragma unroll
        for (int mm = 0; mm < r; ++mm) t = fma(-L[tri(r, mm)], z[mm], t);
        z[r] = t * Linv[r];
      }
      xn[3] = z[3] * Linv[3];
      xn[2] = fma(-L[tri(3, 2)], xn[3], z[2]) * Linv[2];
      xn[1] = fma(-L[tri(3, 1)], xn[3], fma(-L[tri(2, 1)], xn[2], z[1])) * Linv[1];
      xn[0] = fma(-L[tri(3, 0)], xn[3], fma(-L[tri(2, 0)], xn[2], fma(-L[tri(1, 0)], xn[1], z[0]))) * Linv[0];
    }
    quad_wave_sync();  // (lane 0 of a side wrote L for the other three)
    DUO_STAMP(8);
    // ---- backward through this side's half: x_v = L^-T (z - W x_{v+1}); coefficients and cost of local segment v
    double p_end;
    {
      const LaneIds id = ids_now();
      p_end = pos(id, side_len(id.side));
    }
    for (int v = Mmx - 1; v >= 0; --v) {
      const LaneIds id = ids_now();
      const int side = id.side, dim = id.dim, pl = id.pl;  // (this iteration's: see lane_now)
      const bool on = active && v < side_len(side);
      double x[kNB] = {0.0, 0.0, 0.0, 0.0};
      const double p_start = pos(id, v);
      if (on && v >= 1) {
        const double* rec = rec0 + (size_t)(overt(side, v) - 1) * kQdRec * kDuoPaths + pl;
        double t[kNB];
#pragma unroll
        for (int r = 0; r < kNB; ++r) t[r] = rec[(kQdZ + r * kD + dim) * kDuoPaths];
        const double l10 = rec[(kQdL + 0) * kDuoPaths], l20 = rec[(kQdL + 1) * kDuoPaths], l21 = rec[(kQdL + 2) * kDuoPaths],
                     l30 = rec[(kQdL + 3) * kDuoPaths], l31 = rec[(kQdL + 4) * kDuoPaths], l32 = rec[(kQdL + 5) * kDuoPaths];
        const double i0 = rec[(kQdLinv + 0) * kDuoPaths], i1 = rec[(kQdLinv + 1) * kDuoPaths], i2 = rec[(kQdLinv + 2) * kDuoPaths],
                     i3 = rec[(kQdLinv + 3) * kDuoPaths];
        {  // t = z - W x_{v+1},  W x = L^-1 (E x)  (local vertex v + 1 is an unknown for every v <= M - 1)
          double pw[9];
          quad_powers(tbuf[oseg(side, v) * kDuoPaths + pl], d1, d2, d4, pw);
          double u[kNB];
#pragma unroll
          for (int r = 0; r < kNB; ++r) {
            u[r] = 0.0;
#pragma unroll
            for (int c = 0; c < kNB; ++c) u[r] = fma(cCpl[r][c] * pw[r + c + 2], xn[c], u[r]);
          }
          const double w0 = u[0] * i0;
          const double w1 = fma(-l10, w0, u[1]) * i1;
          const double w2 = fma(-l21, w1, fma(-l20, w0, u[2])) * i2;
          const double w3 = fma(-l32, w2, fma(-l31, w1, fma(-l30, w0, u[3]))) * i3;
          t[0] -= w0;
          t[1] -= w1;
          t[2] -= w2;
          t[3] -= w3;
        }
        x[3] = t[3] * i3;
        x[2] = fma(-l32, x[3], t[2]) * i2;
        x[1] = fma(-l31, x[3], fma(-l21, x[2], t[1])) * i1;
        x[0] = fma(-l30, x[3], fma(-l20, x[2], fma(-l10, x[1], t[0]))) * i0;
      }
      if (any_moving && v == 0 && active && moving_path) {  // the start vertex's constrained derivative values (side 0)
        const double* vrow0 = vals + (size_t)pr.v0 * kHalf * kD + dim;
#pragma unroll
        for (int r = 0; r < kNB; ++r) x[r] += vrow0[(r + 1) * kD];
      }
      if (on) {
        const double T = tbuf[oseg(side, v) * kDuoPaths + pl];
        // the segment's end-point derivatives in FORWARD orientation: side 0 as they are, side 1 with start and end swapped
        // and the odd derivatives negated back
        const double s1 = side ? -1.0 : 1.0;
        const double a0 = side ? p_end : p_start, b0 = side ? p_start : p_end;
        double da[kNB], dbv[kNB];
#pragma unroll
        for (int r = 0; r < kNB; ++r) {
          const double sg = (r & 1) ? 1.0 : s1;
          da[r] = (side ? xn[r] : x[r]) * sg;
          dbv[r] = (side ? x[r] : xn[r]) * sg;
        }
        const double dv[kN] = {a0, da[0], da[1], da[2], da[3], b0, dbv[0], dbv[1], dbv[2], dbv[3]};
        double w[kHalf];
        w[0] = 1.0;
#pragma unroll
        for (int k = 1; k < kHalf; ++k) w[k] = w[k - 1] * T;
        double db[kN], cb[kN], c[kN];
#pragma unroll
        for (int j = 0; j < kN; ++j) db[j] = dv[j] * w[j % kHalf];
        const double ti = 1.0 / T;
        double tik = 1.0;
#pragma unroll
        for (int k = 0; k < kN; ++k) {
          double sacc = 0.0;
          if (k < kHalf) {
            sacc = c_abar_inv[k][k] * db[k];  // the upper half of ABAR_INV is diag(1/k!)
          } else {
#pragma unroll
            for (int j = 0; j < kN; ++j) sacc += c_abar_inv[k][j] * db[j];
          }
          cb[k] = sacc;
          c[k] = sacc * tik;
          tik *= ti;
        }
        double p2[9];
        quad_powers(T, d1, d2, d4, p2);  // p2[0] = T^(1 - 2d)
        my_cost = fma(cost_quadratic_form<kOrder>(cb), p2[0], my_cost);
        double2* out = reinterpret_cast<double2*>(coeffs + ((size_t)(pr.s0 + oseg(side, v)) * kD + dim) * kN);
#pragma unroll
        for (int k = 0; k < kN; k += 2) store_coeff_pair(out + k / 2, c[k], c[k + 1]);
      }
      if (on) {  // (a side shorter than the wavefront's longest joins late: until then xn is the middle vertex's solution)
#pragma unroll
        for (int r = 0; r < kNB; ++r) xn[r] = x[r];
        p_end = p_start;
      }
#ifdef MRS_TG_DUO_STAMPS
      if (v == Mmx - 1) DUO_STAMP(9);
#endif
    }
    DUO_STAMP(10);
    duo_finish<0>(my_cost, active, pos_bad, pidx, cost, status, status_in, tail, t_sum);
  } else {
    if (active && side == 0) {
    // ---- any other constraint pattern: the general masked step on the four lanes of side 0, factors in the plan's workspace
    bool pok = true;
    BlockSource none{nullptr, nullptr, 0, 0};
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    const double* times = (scaling ? tail.seg_times_out : seg_times) + pr.s0;
    my_cost = solve_path<1, true>(mask, vals, pr.v0, S, d, times, dim, none, ws, (size_t)b.n_paths * 4,
                                  (unsigned)q * 4u + (unsigned)dim, coeffs + (size_t)pr.s0 * kD * kN, pok);
    }
    duo_finish<2>(my_cost, active, pos_bad, pidx, cost, status, status_in, tail, t_sum);
  }
#ifdef MRS_TG_DUO_STAMPS
  DUO_STAMP(11);
  __builtin_amdgcn_s_waitcnt(0);  // (every store acknowledged)
  DUO_STAMP(12);
  if (lane == 0 && blockIdx.x < 2048)
    for (int k = 0; k < 16; ++k) g_duo_stamps[blockIdx.x * 16 + k] = duo_stamp[k];
#endif
}

#ifdef MRS_TG_DUO_STAMPS
extern "C" int mrs_tg_debug_duo_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_duo_stamps), sizeof(unsigned long long) * 2048 * 16);
}
#endif

template <bool WP>
__global__ __launch_bounds__(64, MRS_TG_QUAD_WAVES) void solve_duo_kernel(BatchView b, int d, const uint8_t* __restrict__ mask,
                                                       const double* __restrict__ vals, const double* seg_times,
                                                       double* __restrict__ coeffs, int32_t* __restrict__ status,
                                                       double* __restrict__ cost, const int32_t* __restrict__ status_in,
                                                       double* ws, RowsTail tail) {
  solve_duo_body<WP, true>(b, d, mask, vals, seg_times, coeffs, status, cost, status_in, ws, tail, (int)blockIdx.x, tail.pos_wp);
}

template <bool WP>
__global__ __launch_bounds__(64, MRS_TG_DUO_GROUP_WAVES) void solve_duo_group_kernel(BatchView b, int d, RowsGroup g, double* ws,
                                                                                    size_t ws_batch_doubles, int blocks_per_batch) {
  const int j = __builtin_amdgcn_readfirstlane((int)blockIdx.x / blocks_per_batch);
  solve_duo_body<WP>(b, d, g.mask[j], g.vals[j], g.seg_times[j], g.coeffs[j], g.status[j], g.cost[j], nullptr,
                     ws + (size_t)j * ws_batch_doubles, RowsTail(), (int)blockIdx.x - j * blocks_per_batch, g.pos_wp[j]);
}

template <bool WP, bool ENDS = false>
__global__ __launch_bounds__(64, MRS_TG_QUAD_WAVES) void solve_quad_kernel(BatchView b, int d, const uint8_t* __restrict__ mask,
                                                        const double* __restrict__ vals, const double* seg_times,
                                                        double* __restrict__ coeffs, int32_t* __restrict__ status,
                                                        double* __restrict__ cost, const int32_t* __restrict__ status_in,
                                                        double* ws, RowsTail tail) {
  solve_quad_body<WP, ENDS, true>(b, d, mask, vals, seg_times, coeffs, status, cost, status_in, ws, tail, (int)blockIdx.x, tail.pos_wp);
}

// several batches of ONE plan in one launch (as solve_rows_group_kernel): workgroups [j * blocks_per_batch, ...) solve batch j
// (ws: one factor store of ws_batch_doubles per batch, for wavefronts that take the general step)
// (WP: every batch of the group states that its positions are its waypoints)
template <bool WP, bool ENDS = false>
__global__ __launch_bounds__(64, MRS_TG_QUAD_GROUP_WAVES) void solve_quad_group_kernel(BatchView b, int d, RowsGroup g, double* ws, size_t ws_batch_doubles,
                                                              int blocks_per_batch) {
  const int j = __builtin_amdgcn_readfirstlane((int)blockIdx.x / blocks_per_batch);
  solve_quad_body<WP, ENDS>(b, d, g.mask[j], g.vals[j], g.seg_times[j], g.coeffs[j], g.status[j], g.cost[j], nullptr,
                            ws + (size_t)j * ws_batch_doubles, RowsTail(), (int)blockIdx.x - j * blocks_per_batch, g.pos_wp[j]);
}

// ---------------------------------------------------------------------------------------------
// launcher

static constexpr size_t kQuadLdsBudget = 80 * 1024;  // at least two wavefronts per CU

// MRS_TG_QUAD_MIN_PATHS: paths per launch from which the quad kernel takes over (tuning / test knob, read once per process)
static long long quad_min_paths() {
  static const long long v = [] {
    const char* e = std::getenv("MRS_TG_QUAD_MIN_PATHS");
    return e ? std::atoll(e) : 6144ll;
  }();
  return v;
}

// MRS_TG_QUAD_ENDS=0: the general step for paths whose end vertices leave slots free, as until round 5 (read once per process)
static bool quad_ends_allowed() {
  static const bool v = [] {
    const char* e = std::getenv("MRS_TG_QUAD_ENDS");
    return e == nullptr || std::atoi(e) != 0;
  }();
  return v;
}

// The two-sided kernel takes a launch whose quad wavefronts (16 paths each) would leave SIMDs idle or barely covered: fewer than
// 1.25 per SIMD.  MRS_TG_DUO=0: never, =1: whenever the pattern allows (tuning / test knob).
static bool duo_pays(long long paths_in_launch) {
  if (const char* e = std::getenv("MRS_TG_DUO")) return std::atoi(e) != 0;  // (read at every call: tests run both kernels)
  static const int cus = [] {  // (of the first device used: the threshold is a tuning figure, results do not depend on it)
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n;
  }();
  return (paths_in_launch + kQdPaths - 1) / kQdPaths < (long long)cus * 4 * 5 / 4;
}

bool quad_kernel_applies(const BatchView& b, long long paths_in_launch, bool with_sampling) {
  if (b.n_paths == 0 || with_sampling) return false;
  if (paths_in_launch < quad_min_paths()) return false;
  return quad_lds_doubles(b.max_segments) * sizeof(double) <= kQuadLdsBudget;
}

hipError_t launch_solve_quad(const BatchView& b, int d, const uint8_t* mask, const double* vals, const double* seg_times,
                             double* coeffs, int32_t* status, double* cost, const int32_t* status_in, double* ws,
                             hipStream_t stream, const RowsTail& tail) {
  // objective orders below snap leave jerk and / or snap free at the end vertices of a rest-to-rest path: the instantiation
  // that eliminates such end vertices (two more records per path in LDS), while its LDS fits; MRS_TG_QUAD_ENDS=0: the general
  // step for those paths, as until round 5
  const bool ends = quad_ends_allowed() && (d < 4 || constrained_slots_hint()) &&
                    quad_lds_doubles(b.max_segments, true) * sizeof(double) <= kQuadLdsBudget;
  const bool wp = tail.pos_wp != nullptr;
  if (!ends && !(d < 4) && duo_pays(b.n_paths)) {  // few wavefronts: eight lanes per path, the chain cut in the middle
    const size_t lds_duo = duo_lds_doubles(b.max_segments) * sizeof(double);
    if (lds_duo > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(wp ? (const void*)solve_duo_kernel<true> : (const void*)solve_duo_kernel<false>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)kQuadLdsBudget);
      if (e != hipSuccess) return e;
    }
    const unsigned grid_duo = (unsigned)((b.n_paths + kDuoPaths - 1) / kDuoPaths);
    if (wp)
      MRS_TG_LAUNCH_TIMED(solve_duo_kernel<true>, dim3(grid_duo), dim3(64), lds_duo, stream, b, d, mask, vals, seg_times, coeffs, status,
                          cost, status_in, ws, tail);
    else
      MRS_TG_LAUNCH_TIMED(solve_duo_kernel<false>, dim3(grid_duo), dim3(64), lds_duo, stream, b, d, mask, vals, seg_times, coeffs, status,
                          cost, status_in, ws, tail);
    return hipGetLastError();
  }
  const size_t lds_bytes = quad_lds_doubles(b.max_segments, ends) * sizeof(double);
  const void* fn = ends ? (wp ? (const void*)solve_quad_kernel<true, true> : (const void*)solve_quad_kernel<false, true>)
                        : (wp ? (const void*)solve_quad_kernel<true> : (const void*)solve_quad_kernel<false>);
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kQuadLdsBudget);
    if (e != hipSuccess) return e;
  }
  const unsigned grid = (unsigned)((b.n_paths + kQdPaths - 1) / kQdPaths);
  if (ends && wp)
    MRS_TG_LAUNCH_TIMED_T2(solve_quad_kernel, true, true, dim3(grid), dim3(64), lds_bytes, stream, b, d, mask, vals, seg_times, coeffs,
                        status, cost, status_in, ws, tail);
  else if (ends)
    MRS_TG_LAUNCH_TIMED_T2(solve_quad_kernel, false, true, dim3(grid), dim3(64), lds_bytes, stream, b, d, mask, vals, seg_times, coeffs,
                        status, cost, status_in, ws, tail);
  else if (wp)
    MRS_TG_LAUNCH_TIMED(solve_quad_kernel<true>, dim3(grid), dim3(64), lds_bytes, stream, b, d, mask, vals, seg_times, coeffs, status,
                        cost, status_in, ws, tail);
  else
    MRS_TG_LAUNCH_TIMED(solve_quad_kernel<false>, dim3(grid), dim3(64), lds_bytes, stream, b, d, mask, vals, seg_times, coeffs, status,
                        cost, status_in, ws, tail);
  return hipGetLastError();
}

hipError_t launch_solve_quad_group(const BatchView& b, int d, const RowsGroup& g, double* ws, hipStream_t stream) {
  if (g.n < 1 || g.n > kRowsGroupMax) return hipErrorInvalidValue;
  const bool ends = quad_ends_allowed() && (d < 4 || constrained_slots_hint()) &&
                    quad_lds_doubles(b.max_segments, true) * sizeof(double) <= kQuadLdsBudget;
  const size_t lds_bytes = quad_lds_doubles(b.max_segments, ends) * sizeof(double);
  bool wp = true;
  for (int j = 0; j < g.n; ++j) wp = wp && g.pos_wp[j] != nullptr;
  if (!ends && !(d < 4) && duo_pays((long long)b.n_paths * g.n)) {
    const size_t lds_duo = duo_lds_doubles(b.max_segments) * sizeof(double);
    if (lds_duo > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(wp ? (const void*)solve_duo_group_kernel<true> : (const void*)solve_duo_group_kernel<false>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)kQuadLdsBudget);
      if (e != hipSuccess) return e;
    }
    const int per_batch_duo = (b.n_paths + kDuoPaths - 1) / kDuoPaths;
    const dim3 grid_duo((unsigned)(per_batch_duo * g.n));
    const size_t wsd_duo = linear_workspace_doubles(b);
    if (wp)
      MRS_TG_LAUNCH_TIMED(solve_duo_group_kernel<true>, grid_duo, dim3(64), lds_duo, stream, b, d, g, ws, wsd_duo, per_batch_duo);
    else
      MRS_TG_LAUNCH_TIMED(solve_duo_group_kernel<false>, grid_duo, dim3(64), lds_duo, stream, b, d, g, ws, wsd_duo, per_batch_duo);
    return hipGetLastError();
  }
  if (lds_bytes > 64 * 1024) {
    const void* fn = ends ? (wp ? (const void*)solve_quad_group_kernel<true, true> : (const void*)solve_quad_group_kernel<false, true>)
                          : (wp ? (const void*)solve_quad_group_kernel<true> : (const void*)solve_quad_group_kernel<false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kQuadLdsBudget);
    if (e != hipSuccess) return e;
  }
  const int per_batch = (b.n_paths + kQdPaths - 1) / kQdPaths;
  const dim3 grid((unsigned)(per_batch * g.n));
  const size_t wsd = linear_workspace_doubles(b);
  if (ends && wp)
    MRS_TG_LAUNCH_TIMED_T2(solve_quad_group_kernel, true, true, grid, dim3(64), lds_bytes, stream, b, d, g, ws, wsd, per_batch);
  else if (ends)
    MRS_TG_LAUNCH_TIMED_T2(solve_quad_group_kernel, false, true, grid, dim3(64), lds_bytes, stream, b, d, g, ws, wsd, per_batch);
  else if (wp)
    MRS_TG_LAUNCH_TIMED(solve_quad_group_kernel<true>, grid, dim3(64), lds_bytes, stream, b, d, g, ws, wsd, per_batch);
  else
    MRS_TG_LAUNCH_TIMED(solve_quad_group_kernel<false>, grid, dim3(64), lds_bytes, stream, b, d, g, ws, wsd, per_batch);
  return hipGetLastError();
}

}  // namespace mrs_tg
