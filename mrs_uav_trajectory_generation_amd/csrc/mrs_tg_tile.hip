// mrs_tg_tile.hip -- linear QP solve, phase-split over a tile of paths (the latency-optimised K2).
//
// The per-path solve has two kinds of work:
//   * work that is independent per (segment, dimension): u = H [f_i; f_{i+1}] (the right-hand-side and
//     f^T H f contributions), masking of the 4x4 blocks, and at the end c = A^-1 d;
//   * an inherently serial chain over the vertices: block Cholesky / forward substitution / back
//     substitution (PolynomialOptimization::solveLinear,
//     /root/reference/include/eth_trajectory_generation/impl/polynomial_optimization_linear_impl.h:341-373).
// One-lane-per-path kernels run both on the serial lane.  Here a workgroup owns a tile of TP paths, stages
// everything in LDS and runs
//   A0  one lane per vertex:               constraints -> f, free masks                     (parallel)
//   A1  one lane per (segment, dimension): blocks (from HBM or from T) -> u and masked blocks, added into the
//       records of the segment's two end vertices (LDS atomic adds)                          (parallel)
//   B   eight lanes per path (direction x dimension): two-sided elimination of the vertex chain on the
//       LDS-resident 4x4 blocks, meeting at the middle vertex                                  (serial, S/2 deep)
//   C   one lane per (segment, dimension): c = A^-1 d -> global coefficients                (parallel)
// so the serial lanes execute only the chain (~1/3 of the instructions of the one-lane kernel).
// Wavefront 0 of the 256-thread workgroup owns A0 and B (and, beside phase C, the cost and status of its paths);
// wavefronts 1-3 ("workers") own A1 and C.  The workers' global loads never wait for a phase boundary: the blocks (or
// segment time) of a worker's first A1 item are requested before A0 runs and the A^-1 rows of its first C item right
// after the barrier that starts B, and the barriers order LDS traffic only, so both latencies hide behind the other
// wavefront's phase.  Tiles are numbered so that each XCD owns a contiguous range of them.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "mrs_tg_device.hpp"
#include "mrs_tg_solve.hpp"

namespace mrs_tg {

// LDS record sizes (doubles).  A vertex's diagonal block D_v = Hee(segment v-1) + Hss(segment v) and right-hand side
// Y_v = uend(v-1) + ustart(v) are accumulated into the vertex record by the segment lanes of phase A1 (LDS atomic
// adds of two terms each: commutative, so bit-reproducible); phase B reads them from the slots it then overwrites
// with L and z.  (An earlier layout kept the four pieces in the segment records and summed them inside the serial
// chain: 11.4 KB per path instead of 7.5 KB, one workgroup per CU instead of two at eight paths per tile.)
constexpr int kSegRec = 16 + 4;            // coupling block EM (as its consumer reads it), qf[dim]
constexpr int kVtxRec = 42 + 20 + 2 + 2;   // D[10] -> L[10] | W[16] | Y[4][4] -> z[4][4] | d[5][4] | free bits, flags | pad
// (the pad takes the record off a multiple of 16 doubles = the 32-bank period of ds_read2_b64 / ds_write: the two
// directions of a path work on different vertices and met on the same banks at every access of phase B --
// SQ_LDS_BANK_CONFLICT was 27 % of the kernel's LDS cycles, profiles/round1_pmc_sq_linear_step.csv)
constexpr int kHandOver = 10 + 16;         // per path: the backward direction's Schur update for the middle vertex

// Per-path LDS stride.  Both record sizes are multiples of 8 doubles, so the unpadded stride put every path of a
// tile on the same banks ((a/4) mod 32 for ds_read2_b64 / ds_write*): the 16-lane groups of phase B (two paths x two
// directions) then hit one bank with up to four distinct addresses.  A stride of 4 (mod 16) doubles shifts each
// path by 8 banks.
__host__ __device__ constexpr int tile_path_doubles(int S) {
  const int base = S * kSegRec + (S + 1) * kVtxRec + kHandOver;
  return base + ((4 - base % 16) + 16) % 16;
}


constexpr int kTileThreads = 256;
constexpr int kTileWorkers = kTileThreads - 64;

// Two workgroups per CU (eight wavefronts, two per SIMD) need <= 256 VGPRs.  The blocks variant sits just below that
// (248); the bound keeps it there -- a change that cost 14 more registers silently took the larger batches back to one
// workgroup per CU (8192 x 10 linear 61 -> 76 us).
template <bool FUSED>
__global__ __launch_bounds__(kTileThreads, 2) void solve_tile_kernel(BatchView b, int d, int TP, int Smax,
                                                                  const uint8_t* __restrict__ mask,
                                                                  const double* __restrict__ vals,
                                                                  const double* __restrict__ seg_times,
                                                                  const double* __restrict__ Hblk,
                                                                  const double* __restrict__ Ablk,
                                                                  double* __restrict__ coeffs, int32_t* __restrict__ status,
                                                                  double* __restrict__ cost,
                                                                  const int32_t* __restrict__ status_in) {
  extern __shared__ double lds[];
  __shared__ int s_S[16], s_s0[16], s_v0[16], s_p[16];
  const int tid = threadIdx.x;
  const int wtid = tid - 64;  // worker index; negative on wavefront 0
  MRS_TG_PHASE_MARK(0);
  // Workgroups are dispatched to the eight XCDs round-robin by blockIdx, and each XCD has its own L2.  In the blocks
  // variant a 128-byte line of an SoA block row (one entry, 16 consecutive paths) is shared by the neighbouring tiles,
  // so neighbouring tiles go to the same XCD: XCD x owns a contiguous range of tiles.
  const int q0 = xcd_contiguous_index(blockIdx.x, gridDim.x) * TP;
  const int n_here = min(TP, b.n_paths - q0);
  const int PS = tile_path_doubles(Smax);
  const size_t P = (size_t)b.n_paths;

  if (tid < n_here) {
    const PathRef pr = path_at(b, q0 + tid);
    s_S[tid] = pr.S;
    s_s0[tid] = pr.s0;
    s_v0[tid] = pr.v0;
    s_p[tid] = pr.p;
  }
  auto seg_rec = [&](int t, int i) { return lds + (size_t)t * PS + (size_t)i * kSegRec; };
  auto vtx_rec = [&](int t, int v) { return lds + (size_t)t * PS + (size_t)Smax * kSegRec + (size_t)v * kVtxRec; };
  auto hand_over = [&](int t) { return lds + (size_t)t * PS + (size_t)Smax * kSegRec + (size_t)(Smax + 1) * kVtxRec; };
  __syncthreads();
  MRS_TG_PHASE_MARK(1);

  // item = (segment i, path t of the tile, dimension): the unit of the parallel phases A1 and C
  const int n_items = n_here * Smax * kD;
  // q / n_here for q < 8192 without an integer division (~40 instructions each on this hardware, and every item of
  // A1 and C started with two of them): floor(q * ceil(2^16 / n) / 2^16) is exact while q * n < 2^16
  const unsigned inv_n = (65536u + (unsigned)n_here - 1u) / (unsigned)n_here;  // once per workgroup, wave-uniform
  auto div_n = [&](int q) { return (int)(((unsigned)q * inv_n) >> 16); };
  auto item_valid = [&](int item, int& t, int& i, int& dim) {
    dim = item % kD;
    const int q = item / kD;  // (segment, path) index: path fastest
    i = div_n(q);
    t = q - i * n_here;
    return item < n_items && i < s_S[t];
  };

  // A0's own global loads go out first (one vertex per thread; further rounds, if any, follow below): they are the
  // shortest path to the next barrier, and queued behind the workers' block prefetch they used to arrive last
  // (phase A0 of the blocks variant: 5100 -> cycles measured in scripts/tile_phases.hip)
  const int n_vertex_items = n_here * (Smax + 1);
  double f_first[kHalf][kD];
  unsigned fb_first = 0;
  bool pos_first = false;
  const int v_first = div_n(tid), t_first = tid - v_first * n_here;
  const bool vertex_first = tid < n_vertex_items && v_first <= s_S[t_first];
  if (vertex_first) fb_first = load_vertex<kD>(mask, vals, s_v0[t_first] + v_first, 0, f_first, pos_first);

  // workers: request the inputs of their first A1 item now; they arrive while A0 runs
  double Hs[kSym10];
  double T_first = 1.0;
  {
    int t, i, dim;
    if (wtid >= 0 && item_valid(wtid, t, i, dim)) {
      if (FUSED) {
        T_first = seg_times[s_s0[t] + i];
      } else {
        const BlockSource src{Hblk, Ablk, P, q0 + t};
        load_H_blocks32(src, i, Hs);
      }
    }
  }

  // ---- A0: vertex constraints -> d (constrained values, 0 where free), free bits; the vertex accumulators and the
  // hand-over area of every path are zeroed (phase B reads the latter even when nothing was handed over)
  for (int e = tid; e < n_here * kHandOver; e += kTileThreads) hand_over(e / kHandOver)[e % kHandOver] = 0.0;
  auto store_vertex = [&](int t, int v, const double (&f)[kHalf][kD], unsigned fb, bool pos_fixed) {
    double* r = vtx_rec(t, v);
    // accumulators of phase A1
#pragma unroll
    for (int e = 0; e < 10; ++e) r[e] = 0.0;
#pragma unroll
    for (int e = 0; e < kNB * kD; ++e) r[26 + e] = 0.0;
#pragma unroll
    for (int k = 0; k < kHalf; ++k)
#pragma unroll
      for (int dd = 0; dd < kD; ++dd) r[42 + k * kD + dd] = f[k][dd];
    r[62] = (double)fb;
    r[63] = pos_fixed ? 1.0 : 0.0;
  };
  if (vertex_first) store_vertex(t_first, v_first, f_first, fb_first, pos_first);
  for (int item = tid + kTileThreads; item < n_vertex_items; item += kTileThreads) {
    const int v = div_n(item), t = item - v * n_here;
    if (v > s_S[t]) continue;
    double f[kHalf][kD];
    bool pos_fixed;
    const unsigned fb = load_vertex<kD>(mask, vals, s_v0[t] + v, 0, f, pos_fixed);
    store_vertex(t, v, f, fb, pos_fixed);
  }
  lds_barrier();  // not __syncthreads(): the workers' block loads stay in flight across it
  MRS_TG_PHASE_MARK(2);

  // ---- A1 (workers): per (segment, dimension): u = H [f_i; f_{i+1}], masked 4x4 blocks
  if (wtid >= 0) {
    for (int item = wtid; item < n_items; item += kTileWorkers) {
      int t, i, dim;
      if (!item_valid(item, t, i, dim)) continue;
      if (FUSED) {
        hessian_from_time((item == wtid) ? T_first : seg_times[s_s0[t] + i], d, Hs);
      } else if (item != wtid) {
        const BlockSource src{Hblk, Ablk, P, q0 + t};
        load_H_blocks32(src, i, Hs);
      }
      MRS_TG_PHASE_MARK_T(16, 64);
      const double* vs = vtx_rec(t, i);
      const double* ve = vtx_rec(t, i + 1);
      const unsigned free_s = (unsigned)vs[62], free_e = (unsigned)ve[62];
      double f[kN];
#pragma unroll
      for (int k = 0; k < kHalf; ++k) {
        f[k] = vs[42 + k * kD + dim];
        f[kHalf + k] = ve[42 + k * kD + dim];
      }
      double* rec = seg_rec(t, i);
      double* acc_s = const_cast<double*>(vs);  // vertex records of the two ends (accumulators)
      double* acc_e = const_cast<double*>(ve);
      double qf = 0.0;
#pragma unroll
      for (int a = 0; a < kN; ++a) {
        double u = 0.0;
#pragma unroll
        for (int c = 0; c < kN; ++c) u += Hs[sym10(a, c)] * f[c];
        qf += f[a] * u;
        if (a >= kSlot0 && a < kHalf)
          lds_add(acc_s + 26 + (a - kSlot0) * kD + dim, ((free_s >> (a - kSlot0)) & 1u) ? u : 0.0);
        if (a >= kHalf + kSlot0)
          lds_add(acc_e + 26 + (a - kHalf - kSlot0) * kD + dim, ((free_e >> (a - kHalf - kSlot0)) & 1u) ? u : 0.0);
      }
      rec[16 + dim] = qf;
      MRS_TG_PHASE_MARK_T(17, 64);
      if (dim == 0) {
        // masked blocks D_s H D_e with D = diag(free bits) as 0.0 / 1.0 factors: two multiplications per entry
        // instead of bit tests and selects (this lane works alone here, the other three of its quad wait)
        const bool transposed = i >= s_S[t] / 2;
        double ms[kNB], me[kNB];
#pragma unroll
        for (int r = 0; r < kNB; ++r) {
          ms[r] = (double)((free_s >> r) & 1u);
          me[r] = (double)((free_e >> r) & 1u);
        }
#pragma unroll
        for (int r = 0; r < kNB; ++r) {
#pragma unroll
          for (int c = 0; c <= r; ++c) {
            lds_add(acc_s + tri(r, c), (ms[r] * ms[c]) * Hs[sym10(kSlot0 + r, kSlot0 + c)]);
            lds_add(acc_e + tri(r, c), (me[r] * me[c]) * Hs[sym10(kHalf + kSlot0 + r, kHalf + kSlot0 + c)]);
          }
          // the coupling block is stored the way its consumer reads it: segments left of the middle vertex belong to
          // the forward direction of phase B (E[r][c]), the others to the backward direction (E transposed)
#pragma unroll
          for (int c = 0; c < kNB; ++c) {
            const double val = (ms[r] * me[c]) * Hs[sym10(kSlot0 + r, kHalf + kSlot0 + c)];
            rec[transposed ? c * kNB + r : r * kNB + c] = val;
          }
        }
      }
    }
  }
  MRS_TG_PHASE_MARK_T(18, 64);
  lds_barrier();
  MRS_TG_PHASE_MARK(3);
  // workers: request the A^-1 rows of their first C item now -- after the barrier that starts phase B (issuing the 55
  // loads takes a worker ~2600 cycles, which used to sit in front of that barrier); they arrive while wavefront 0 runs B
  AinvRows ar_first;
  if (!FUSED) {
    int t, i, dim;
    if (wtid >= 0 && item_valid(wtid, t, i, dim)) {
      const BlockSource src{Hblk, Ablk, P, q0 + t};
      load_A_blocks32(src, i, ar_first);
    }
  }
  MRS_TG_PHASE_MARK_T(19, 64);

  double red_both = 0.0;  // sum |z|^2 of this lane's dimension, both directions (phase B -> C)
  // ---- B: the vertex chain, eight lanes per path: lane = (direction, dimension).
  // Two-sided ("twisted") block elimination: direction 0 eliminates vertices 0, 1, ... from the left,
  // direction 1 eliminates S, S-1, ... from the right, both towards the middle vertex m = S/2; then the
  // middle block (which receives a Schur update from each side) is solved and the two halves are
  // back-substituted outwards.  Same minimiser, same cost 0.5 (qf - sum |z|^2), half the dependent steps.
  if (tid < n_here * 2 * kD) {
    const int t = tid / (2 * kD), dir = (tid / kD) & 1, dim = tid % kD;
    const int S = s_S[t];
    const int m = S / 2;                       // middle vertex
    const int len = dir ? (S - m) : m;         // vertices this side eliminates: [0, m) or (m, S]
    double Wp[kNB][kNB], zp[kNB];
    double L[10], Linv[kNB], z[kNB], W[kNB][kNB];
    double red = 0.0;
#pragma unroll
    for (int r = 0; r < kNB; ++r) {
      zp[r] = 0.0;
#pragma unroll
      for (int c = 0; c < kNB; ++c) Wp[r][c] = 0.0;
    }
    // One elimination step at vertex v.  INNER: there is a next vertex towards the middle to couple with;
    // MIDDLE: v is the middle vertex, whose record holds the other side's Schur update (zero if there is none).
    // The inputs of a step (StepIn) are loaded unconditionally up front (absent neighbours read the zero record, the
    // first step subtracts the zero-initialised Wp / zp), so the step has one LDS round trip and no divergent branches.
    // (Loading them one step ahead into a second register set changed nothing: the step is bound by its ~220
    // in-order instructions, not by the LDS latency.)
    struct StepIn {
      double D[10], Y[kNB], E[kNB][kNB];
      unsigned fb;
    };
    auto load_step = [&](int v, StepIn& in, auto inner_tag) {
      constexpr bool INNER = decltype(inner_tag)::value;
      const double* vr = vtx_rec(t, v);
      in.fb = (unsigned)vr[62];
#pragma unroll
      for (int e = 0; e < 10; ++e) in.D[e] = vr[e];
#pragma unroll
      for (int r = 0; r < kNB; ++r) in.Y[r] = vr[26 + r * kD + dim];
      if (INNER) {
        // coupling towards the middle: direction 0 reads E~_v[r][c] of segment v, direction 1 E~_{v-1} transposed of
        // segment v - 1 -- phase A1 stored it that way
        const double* cb = seg_rec(t, dir ? v - 1 : v);
#pragma unroll
        for (int r = 0; r < kNB; ++r)
#pragma unroll
          for (int c = 0; c < kNB; ++c) in.E[r][c] = cb[r * kNB + c];
      }
    };
    auto eliminate = [&](int v, const StepIn& in, auto inner_tag, auto middle_tag) {
      constexpr bool INNER = decltype(inner_tag)::value, MIDDLE = decltype(middle_tag)::value;
      double* vr = vtx_rec(t, v);
      const unsigned fb = in.fb;
      double Sm[10], y[kNB];
#pragma unroll
      for (int e = 0; e < 10; ++e) Sm[e] = in.D[e];
#pragma unroll
      for (int r = 0; r < kNB; ++r) y[r] = -in.Y[r];
      if (MIDDLE) {
        const double* ho = hand_over(t);
#pragma unroll
        for (int e = 0; e < 10; ++e) Sm[e] -= ho[e];
#pragma unroll
        for (int r = 0; r < kNB; ++r) y[r] -= ho[10 + r * kD + dim];
      }
#pragma unroll
      for (int r = 0; r < kNB; ++r) {
#pragma unroll
        for (int c = 0; c <= r; ++c) {
          double s = Sm[tri(r, c)];
#pragma unroll
          for (int k = 0; k < kNB; ++k) s -= Wp[k][r] * Wp[k][c];
          Sm[tri(r, c)] = s;
        }
        double s = y[r];
#pragma unroll
        for (int k = 0; k < kNB; ++k) s -= Wp[k][r] * zp[k];
        y[r] = s;
        if (!((fb >> r) & 1u)) Sm[tri(r, r)] = 1.0;  // constrained slot: identity row (its off-diagonals are masked to 0)
      }
#pragma unroll
      for (int c = 0; c < kNB; ++c) {
        double dsum = Sm[tri(c, c)];
#pragma unroll
        for (int k = 0; k < c; ++k) dsum -= L[tri(c, k)] * L[tri(c, k)];
        const double inv = rsqrt_refined(dsum);
        L[tri(c, c)] = dsum * inv;
        Linv[c] = inv;
#pragma unroll
        for (int r = c + 1; r < kNB; ++r) {
          double s = Sm[tri(r, c)];
#pragma unroll
          for (int k = 0; k < c; ++k) s -= L[tri(r, k)] * L[tri(c, k)];
          L[tri(r, c)] = s * inv;
        }
      }
#pragma unroll
      for (int r = 0; r < kNB; ++r) {
        double s = y[r];
#pragma unroll
        for (int k = 0; k < r; ++k) s -= L[tri(r, k)] * z[k];
        z[r] = s * Linv[r];
        red += z[r] * z[r];
      }
      if (INNER) {
        // row by row, the four columns side by side: four independent chains for the scheduler
#pragma unroll
        for (int r = 0; r < kNB; ++r)
#pragma unroll
          for (int c = 0; c < kNB; ++c) {
            double s = in.E[r][c];
#pragma unroll
            for (int k = 0; k < r; ++k) s -= L[tri(r, k)] * W[k][c];
            W[r][c] = s * Linv[r];
          }
      }
      // keep the factors for the backward sweep (reciprocal diagonal; written by the dim-0 lane)
      if (dim == 0) {
#pragma unroll
        for (int e = 0; e < 10; ++e) vr[e] = L[e];
#pragma unroll
        for (int r = 0; r < kNB; ++r) vr[tri(r, r)] = Linv[r];
        if (INNER) {
#pragma unroll
          for (int r = 0; r < kNB; ++r)
#pragma unroll
            for (int c = 0; c < kNB; ++c) vr[10 + r * kNB + c] = W[r][c];
        }
      }
#pragma unroll
      for (int r = 0; r < kNB; ++r) vr[26 + r * kD + dim] = z[r];
      if (INNER) {
#pragma unroll
        for (int r = 0; r < kNB; ++r) {
          zp[r] = z[r];
#pragma unroll
          for (int c = 0; c < kNB; ++c) Wp[r][c] = W[r][c];
        }
      }
    };
    using std::true_type;
    using std::false_type;

    // A fully constrained end vertex (every rest-to-rest path has two) has nothing to eliminate: its block is the
    // identity, z = 0, W = 0.  When that holds for every lane of the wavefront the chains start one vertex further in,
    // which takes one of the S/2 + 1 dependent steps off the phase.
    int s_first = 0;
    {
      const unsigned end_fb = (len > 0) ? (unsigned)vtx_rec(t, dir ? S : 0)[62] : 0xFu;
      if (__ballot(end_fb != 0u) == 0ull) s_first = 1;
    }
    MRS_TG_PHASE_MARK(10);
    for (int s = s_first; s < len; ++s) {
      const int v = dir ? (S - s) : s;
      StepIn in;
      load_step(v, in, true_type{});
      eliminate(v, in, true_type{}, false_type{});
    }
    MRS_TG_PHASE_MARK(11);
    // direction 1 hands its Schur update W^T W and right-hand-side update W^T z for the middle vertex over
    // through that vertex's (still unused) record
    if (dir == 1 && len > 0) {
      double* vm = hand_over(t);
      if (dim == 0) {
#pragma unroll
        for (int r = 0; r < kNB; ++r)
#pragma unroll
          for (int c = 0; c <= r; ++c) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < kNB; ++k) s += Wp[k][r] * Wp[k][c];
            vm[tri(r, c)] = s;
          }
      }
#pragma unroll
      for (int r = 0; r < kNB; ++r) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < kNB; ++k) s += Wp[k][r] * zp[k];
        vm[10 + r * kD + dim] = s;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    MRS_TG_PHASE_MARK(12);
    double xm[kNB] = {0.0, 0.0, 0.0, 0.0};
    if (dir == 0) {
      // middle vertex: left Schur update from Wp/zp (if any vertex was eliminated on the left), right one
      // from the record (if any on the right)
      StepIn in;
      load_step(m, in, false_type{});
      eliminate(m, in, false_type{}, true_type{});
      double* vr = vtx_rec(t, m);
      // x_m = L^-T z
#pragma unroll
      for (int r = kNB - 1; r >= 0; --r) {
        double s = z[r];
#pragma unroll
        for (int k = r + 1; k < kNB; ++k) s -= L[tri(k, r)] * xm[k];
        xm[r] = s * Linv[r];
      }
#pragma unroll
      for (int r = 0; r < kNB; ++r) {
        vr[42 + (kSlot0 + r) * kD + dim] += xm[r];
        vr[26 + r * kD + dim] = xm[r];  // x_m for the direction-1 lanes
      }
    }
    MRS_TG_PHASE_MARK(13);
    red_both = red + __shfl_xor(red, kD, 64);  // the two directions of this dimension; cost and status follow in phase C
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    MRS_TG_PHASE_MARK(14);
    // outward back-substitution: x_v = L_v^-T (z_v - W_v x_next), next = the neighbour towards the middle
    double xn[kNB];
    {
      const double* vr = vtx_rec(t, m);
#pragma unroll
      for (int r = 0; r < kNB; ++r) xn[r] = vr[26 + r * kD + dim];
    }
    for (int s = len - 1; s >= s_first; --s) {
      const int v = dir ? (S - s) : s;
      double* vr = vtx_rec(t, v);
      double tv[kNB], x[kNB];
#pragma unroll
      for (int r = 0; r < kNB; ++r) {
        double sacc = vr[26 + r * kD + dim];
#pragma unroll
        for (int c = 0; c < kNB; ++c) sacc -= vr[10 + r * kNB + c] * xn[c];
        tv[r] = sacc;
      }
#pragma unroll
      for (int r = kNB - 1; r >= 0; --r) {
        double sacc = tv[r];
#pragma unroll
        for (int k = r + 1; k < kNB; ++k) sacc -= vr[tri(k, r)] * x[k];
        x[r] = sacc * vr[tri(r, r)];
      }
#pragma unroll
      for (int r = 0; r < kNB; ++r) {
        vr[42 + (kSlot0 + r) * kD + dim] += x[r];  // d = f + x (x is 0 on constrained slots)
        xn[r] = x[r];
      }
    }
    MRS_TG_PHASE_MARK(15);
  }
  lds_barrier();
  MRS_TG_PHASE_MARK(4);

  // ---- cost and status (wavefront 0, beside the workers' phase C): cost = 0.5 (f^T H f - sum |z|^2) summed over the
  // dimensions; the eight lanes of a path share the loops over its segments and vertices
  if (tid < n_here * 2 * kD) {
    const int t = tid / (2 * kD), dir = (tid / kD) & 1, dim = tid % kD;
    const int S = s_S[t];
    const int half = S / 2, i0 = dir ? half : 0, i1 = dir ? S : half;
    double qa = 0.0, qb = 0.0;
    int i = i0;
    for (; i + 1 < i1; i += 2) {
      qa += seg_rec(t, i)[16 + dim];
      qb += seg_rec(t, i + 1)[16 + dim];
    }
    if (i < i1) qa += seg_rec(t, i)[16 + dim];
    double qf = qa + qb;
    qf += __shfl_xor(qf, kD, 64);
    double cst = 0.5 * (qf - red_both);
    cst += __shfl_xor(cst, 1, 64);
    cst += __shfl_xor(cst, 2, 64);
    bool ok = true;
    for (int v = tid % (2 * kD); v <= S; v += 2 * kD) ok = ok && (vtx_rec(t, v)[63] != 0.0);
    const unsigned long long bal = __ballot(ok);
    const bool pos_ok = ((bal >> (t * 2 * kD)) & 0xFFull) == 0xFFull;
    if (dir == 0 && dim == 0) {
      if (cost) cost[s_p[t]] = cst;
      if (status) status[s_p[t]] = merge_status(pos_ok, status_in, s_p[t]);
    }
  }

  // ---- C (workers): coefficients c = A^-1 [d_i; d_{i+1}] per (segment, dimension)
  if (wtid >= 0) {
    for (int item = wtid; item < n_items; item += kTileWorkers) {
      int t, i, dim;
      if (!item_valid(item, t, i, dim)) continue;
      const double* vs = vtx_rec(t, i);
      const double* ve = vtx_rec(t, i + 1);
      double dv[kN], c[kN];
#pragma unroll
      for (int k = 0; k < kHalf; ++k) {
        dv[k] = vs[42 + k * kD + dim];
        dv[kHalf + k] = ve[42 + k * kD + dim];
      }
      if (FUSED) {
        coefficients_from_time((item == wtid) ? T_first : seg_times[s_s0[t] + i], dv, c);
      } else {
        if (item != wtid) {
          const BlockSource src{Hblk, Ablk, P, q0 + t};
          load_A_blocks32(src, i, ar_first);
        }
#pragma unroll
        for (int k = 0; k < kHalf; ++k) c[k] = ar_first.diag[k] * dv[k];
#pragma unroll
        for (int k = 0; k < kHalf; ++k) {
          double acc = 0.0;
#pragma unroll
          for (int s = 0; s < kN; ++s) acc += ar_first.low[k][s] * dv[s];
          c[kHalf + k] = acc;
        }
      }
      double* out = coeffs + ((size_t)(s_s0[t] + i) * kD + dim) * kN;
#pragma unroll
      for (int k = 0; k < kN; ++k) out[k] = c[k];
    }
  }
  MRS_TG_PHASE_MARK_T(5, 64);  // a worker's clock: wavefront 0 has nothing to do in phase C
}

// ---------------------------------------------------------------------------------------------
// launcher: returns false when the tile kernel does not apply (caller falls back to the per-lane kernel)

static constexpr size_t kTileLdsBudget = 144 * 1024;

bool tile_kernel_applies(const BatchView& b, bool fused) {
  // measured, 10 segments, one batch in flight, us per linear step / nonlinear pipeline (tile vs lane kernels):
  //   blocks: 8192 paths 61 vs 83, 32768 271 vs 385, 65536 516 vs 572;
  //   fused:  4096 316 vs 343, 8192 513 vs 537, 16384 840 vs 886, 32768 1498 vs 1639, 65536 2861 vs 2872
  // -- since the vertex-accumulator layout (two workgroups per CU at eight paths per tile) the tile kernel wins at
  // every size tried on uniform batches.  Ragged batches (3..30 segments: every tile is sized and looped for the
  // longest path of the batch) cross over earlier: blocks 4096 paths 130 vs 163, 8192 225 vs 234, 16384 478 vs 349;
  // fused 2048 779 vs 857, 4096 744 vs 736, 8192 1070 vs 960 (scripts/ragged_tile_sweep.py).
  long long max_paths = (b.uniform_S > 0) ? (1ll << 40) : (fused ? 4096 : 8192);
  if (const char* e = std::getenv("MRS_TG_TILE_MAX_PATHS")) max_paths = std::atoll(e);  // tuning knob (scripts/sweep_tile.sh)
  if (b.n_paths == 0 || b.n_paths > max_paths) return false;
  // 32-bit byte offsets into the block buffers (load_H_blocks32)
  if (!fused && (unsigned long long)b.max_segments * 800ull * (unsigned long long)b.n_paths > 0xFFFFFFFFull) return false;
  return (size_t)tile_path_doubles(b.max_segments) * sizeof(double) <= kTileLdsBudget;
}

hipError_t launch_solve_tile(const BatchView& b, int d, bool fused, const uint8_t* mask, const double* vals,
                             const double* seg_times, const double* H, const double* Ainv, double* coeffs,
                             int32_t* status, double* cost, const int32_t* status_in, hipStream_t stream) {
  const size_t per_path = (size_t)tile_path_doubles(b.max_segments) * sizeof(double);
  int TP = (int)(kTileLdsBudget / per_path);
  if (TP > 8) TP = 8;  // phase B runs eight lanes per path inside wavefront 0
  // more, smaller tiles so that every CU gets work (256 CUs) and several tiles share a CU
  while (TP > 4 && (b.n_paths + TP - 1) / TP < 512) TP >>= 1;
  const size_t lds_bytes = per_path * (size_t)TP;
  const unsigned grid = (unsigned)((b.n_paths + TP - 1) / TP);
  if (fused) {
    hipError_t e = hipFuncSetAttribute((const void*)solve_tile_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)kTileLdsBudget);
    if (e != hipSuccess) return e;
    MRS_TG_LAUNCH_TIMED(solve_tile_kernel<true>, dim3(grid), dim3(kTileThreads), lds_bytes, stream, b, d, TP, b.max_segments, mask,
                       vals, seg_times, H, Ainv, coeffs, status, cost, status_in);
  } else {
    hipError_t e = hipFuncSetAttribute((const void*)solve_tile_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)kTileLdsBudget);
    if (e != hipSuccess) return e;
    MRS_TG_LAUNCH_TIMED(solve_tile_kernel<false>, dim3(grid), dim3(kTileThreads), lds_bytes, stream, b, d, TP, b.max_segments, mask,
                       vals, seg_times, H, Ainv, coeffs, status, cost, status_in);
  }
  return hipGetLastError();
}

}  // namespace mrs_tg
