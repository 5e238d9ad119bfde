// mrs_tg_kernels.hip -- gfx950 kernels of the batched trajectory optimiser and their launchers.
//
// Data layout in HBM (DESIGN.md "layout"):
//   * ABI-facing arrays are CSR/AoS exactly as include/mrs_tg.h states;
//   * materialised per-segment blocks (H, A^-1) are slot-major SoA: element e = r*10+c of segment j of
//     the path at sorted position q sits at ((j*100 + e) * P + q), so consecutive lanes (consecutive q)
//     touch consecutive doubles for every (j, e): both the assembly kernel's stores and the solve
//     kernel's loads are fully coalesced without any LDS transposition;
//   * the back-substitution workspace uses the same idea: ((v*42 + e) * P + q).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "mrs_tg_device.hpp"
#include "mrs_tg_estimate.hpp"
#include "mrs_tg_sampling.hpp"
#include "mrs_tg_solve.hpp"
#include "mrs_tg_launch.h"
#include "mrs_tg_pool.h"

namespace mrs_tg {

// ---------------------------------------------------------------------------------------------
// K1: Hessian / mapping-block assembly.  HBM-write bound: 8 B read + 1600 B written per segment.
// One thread per (path position, matrix row, segment slot): blockIdx.z is the slot and blockIdx.y the
// row, so there is no index arithmetic beyond one multiply-add, the row's constants are wave-uniform
// (scalar loads) and every store instruction writes 64 consecutive doubles.  The H row is stored before
// the A^-1 row (which needs the division) is computed, so the stores start one dependent load +
// ~20 multiplies after the wave starts.  Launch shapes measured in scripts/k1_variants.hip
// (1024 x 10: 4.8 us against 4.6 us for a pure fill of the same 16.5 MB and 5.2 us for the previous
// two-paths-per-lane, 16-byte-store shape).

// row a of H(T) = T^(1-2d) D_T HBAR_d D_T
__device__ __forceinline__ void hessian_row(double T, int d, int a, double (&hrow)[kN]) {
  const double t2 = T * T, t3 = t2 * T, t4 = t2 * t2;
  const double w[kHalf] = {1.0, T, t2, t3, t4};
  const int pa = a % kHalf;
  const double td = (d == 0) ? 1.0 : (d == 1) ? T : (d == 2) ? t2 : (d == 3) ? t3 : t4;
  const double wa = (pa == 0) ? 1.0 : (pa == 1) ? T : (pa == 2) ? t2 : (pa == 3) ? t3 : t4;
  const double sa = (T / (td * td)) * wa;  // T^(1-2d) * T^pa
#pragma unroll
  for (int c = 0; c < kN; ++c) hrow[c] = c_hbar[d][a][c] * sa * w[c % kHalf];
}

// row a of A^-1(T): ABAR_INV[a][c] * T^(c%5) / T^a
__device__ __forceinline__ void mapping_inverse_row(double T, int a, double (&arow)[kN]) {
  const double t2 = T * T, t3 = t2 * T, t4 = t2 * t2;
  const double w[kHalf] = {1.0, T, t2, t3, t4};
  const double ti = 1.0 / T;
  double tia = 1.0;
  for (int k = 0; k < a; ++k) tia *= ti;
#pragma unroll
  for (int c = 0; c < kN; ++c) arow[c] = c_abar_inv[a][c] * w[c % kHalf] * tia;
}

// The blocks leave as STREAMING stores (round 6): 1 600 bytes per segment that the kernel itself never reads again.  Marked
// non-temporal, a 16.5 MB launch takes 5.3 instead of 5.9 us (0.385 instead of 0.35 of the HBM peak), a 1.05 GB launch 154-157
// instead of 161-165 us (0.84-0.85); the step that solves from the blocks right behind it finds fewer of them in the L2 and is
// 0.5 us slower (12.7-13.5 -> 13.1-14.0 us per 1024 x 10 step).  -DMRS_TG_ASSEMBLE_NT=0: ordinary stores, as until round 5.
#ifndef MRS_TG_ASSEMBLE_NT
#define MRS_TG_ASSEMBLE_NT 1
#endif
__device__ __forceinline__ void store_block_rows(double T, int d, int a, size_t base, size_t P, double* __restrict__ Hout,
                                                 double* __restrict__ Aout) {
  double row[kN];
  hessian_row(T, d, a, row);
#if MRS_TG_ASSEMBLE_NT
#pragma unroll
  for (int c = 0; c < kN; ++c) __builtin_nontemporal_store(row[c], &Hout[base + (size_t)c * P]);
  mapping_inverse_row(T, a, row);
#pragma unroll
  for (int c = 0; c < kN; ++c) __builtin_nontemporal_store(row[c], &Aout[base + (size_t)c * P]);
#else
#pragma unroll
  for (int c = 0; c < kN; ++c) Hout[base + (size_t)c * P] = row[c];
  mapping_inverse_row(T, a, row);
#pragma unroll
  for (int c = 0; c < kN; ++c) Aout[base + (size_t)c * P] = row[c];
#endif
}

// general (ragged) batches: slot j holds the first slot_start[j+1] - slot_start[j] positions of the
// longest-first order; blocks beyond that count exit at once
__global__ __launch_bounds__(256) void assemble_blocks_kernel(BatchView b, int d, const double* __restrict__ seg_times,
                                                              double* __restrict__ Hout, double* __restrict__ Aout) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  const int j = blockIdx.z;
  if (q >= b.slot_start[j + 1] - b.slot_start[j]) return;
  const int a = blockIdx.y;
  const int p = b.order[q];
  const double T = seg_times[b.seg_offsets[p] + j];
  const size_t P = (size_t)b.n_paths;
  store_block_rows(T, d, a, ((size_t)j * 100 + (size_t)a * kN) * P + (size_t)q, P, Hout, Aout);
}

// uniform batches (every path has S segments, so position q == path q and segment (q, j) sits at q*S + j):
// no indirection loads
// One-dimensional grid of (path chunk, row, slot) workgroups, the chunk slowest, numbered so that each XCD owns a
// contiguous range (xcd_contiguous_index): the XCD that writes the blocks of paths [128 c, 128 c + 128) is then the XCD
// whose tiles of the solve kernel read them (same partition of the paths over the XCDs), and finds them in its own L2.
constexpr int kAssembleChunk = 128;
#ifndef MRS_TG_ASSEMBLE_ORDER
#define MRS_TG_ASSEMBLE_ORDER 2   // by size
#endif
__global__ __launch_bounds__(kAssembleChunk) void assemble_blocks_uniform_kernel(int n_paths, int S, int d,
                                                                                 const double* __restrict__ seg_times,
                                                                                 double* __restrict__ Hout,
                                                                                 double* __restrict__ Aout) {
  // Which workgroup writes what.  Small launches (the headline's 1024 paths: 8 chunks): workgroups of one chunk of paths are
  // neighbours in the XCD-contiguous numbering, so the XCD that writes the blocks of a chunk is the one whose solve tiles read
  // them.  Large launches: the CHUNK varies fastest -- the workgroups in flight then write neighbouring 1 KB runs of the same few
  // rows instead of runs scattered over the whole gigabyte (twenty arrays 512 KB apart per workgroup): 65536 x 10 on buffers the
  // driver placed badly 0.63-0.66 -> 0.73-0.75 of the HBM peak, on well-placed ones 0.78 -> 0.80 (round 6,
  // profiles/round6_assembly_placement.txt); MRS_TG_ASSEMBLE_ORDER=0 / 1: the first / second order at every size (A / B builds)
  const int n_chunks = (n_paths + kAssembleChunk - 1) / kAssembleChunk;
  const bool chunk_fastest = MRS_TG_ASSEMBLE_ORDER == 1 || (MRS_TG_ASSEMBLE_ORDER == 2 && n_chunks >= 64);
  int chunk, rem;
  if (chunk_fastest) {
    const int logical = (int)blockIdx.x;
    rem = logical / n_chunks;
    chunk = logical - rem * n_chunks;
  } else {
    const int logical = xcd_contiguous_index(blockIdx.x, gridDim.x);
    const int per_chunk = kN * S;
    chunk = logical / per_chunk;
    rem = logical - chunk * per_chunk;
  }
  const int j = rem / kN, a = rem - j * kN;
  const int q = chunk * kAssembleChunk + threadIdx.x;
  if (q >= n_paths) return;
  const double T = seg_times[(size_t)q * S + j];
  const size_t P = (size_t)n_paths;
  store_block_rows(T, d, a, ((size_t)j * 100 + (size_t)a * kN) * P + (size_t)q, P, Hout, Aout);
}

// ---------------------------------------------------------------------------------------------
// K2: linear QP solve (mrs_tg_solve.hpp).  ND = 4: one lane per path; ND = 1: four lanes per path.
// FUSED = false consumes the materialised blocks written by K1, FUSED = true recomputes them from T.

template <int ND, bool FUSED>
__global__ __launch_bounds__(64) void solve_linear_kernel(BatchView b, int d, const uint8_t* __restrict__ mask,
                                                          const double* __restrict__ vals,
                                                          const double* __restrict__ seg_times,
                                                          const double* __restrict__ Hblk, const double* __restrict__ Ablk,
                                                          double* __restrict__ ws, double* __restrict__ coeffs,
                                                          int32_t* __restrict__ status, double* __restrict__ cost,
                                                          const int32_t* __restrict__ status_in) {
  constexpr int LPP = kD / ND;  // lanes per path
  const int tid = blockIdx.x * 64 + threadIdx.x;
  const int lanes_total = b.n_paths * LPP;
  if (tid >= lanes_total) return;
  const int q = tid / LPP;
  const int dim0 = (tid % LPP) * ND;
  const PathRef pr = path_at(b, q);
  BlockSource src{Hblk, Ablk, (size_t)b.n_paths, q};
  bool pos_ok;
  double c = solve_path<ND, FUSED>(mask, vals, pr.v0, pr.S, d, seg_times + pr.s0, dim0, src, ws, (size_t)lanes_total,
                                   (unsigned)tid, coeffs + (size_t)pr.s0 * kD * kN, pos_ok);
  if (LPP == 4) {
    c += __shfl_xor(c, 1, 64);
    c += __shfl_xor(c, 2, 64);
  }
  if (dim0 == 0) {
    if (cost) cost[pr.p] = c;
    if (status) status[pr.p] = merge_status(pos_ok, status_in, pr.p);
  }
}

// ---------------------------------------------------------------------------------------------
// segment-time initialisation: estimateSegmentTimesEuclidean
// (/root/reference/src/eth_trajectory_generation/vertex.cpp:491-565), one thread per segment

__global__ __launch_bounds__(256) void estimate_times_kernel(BatchView b, const double* __restrict__ wp,
                                                             const double* __restrict__ limits,
                                                             double* __restrict__ seg_times) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= b.n_segments) return;
  // path of CSR segment idx: a division for uniform batches, binary search over seg_offsets otherwise
  int p;
  if (b.uniform_S > 0) {
    p = idx / b.uniform_S;
  } else {
    int lo = 0, hi = b.n_paths;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (b.seg_offsets[mid] <= idx) lo = mid;
      else hi = mid;
    }
    p = lo;
  }
  const int v = idx + p;  // vertex index of the segment's start
  seg_times[idx] = estimate_segment_time(wp + (size_t)v * 4, limits + (size_t)p * 9);
}

// ---------------------------------------------------------------------------------------------
// sampler: Trajectory::evaluateRange's accumulate-and-carry walk
// (/root/reference/src/eth_trajectory_generation/trajectory.cpp:93-151), positions + wrapped heading
// (the only fields the nodelet reads, src/mrs_trajectory_generation.cpp:1582-1599).
//
// The reference's sample times are defined by repeated floating-point addition
// (time_in_segment += dt; accumulated += dt, with the remainder carried into the next segment), so the
// sample count and the segment a boundary sample falls into depend on that exact rounding.  One wavefront
// per path walks the trajectory a chunk of up to 64 samples at a time: lane j adds dt to the chunk's start
// values j times -- the same additions in the same order as the reference's running sums, so its
// time_in_segment / accumulated values are bit-identical to those of the reference's j-th next sample -- a ballot
// finds the first lane at which the walk stops (segment end, trajectory end or capacity), the lanes before it
// evaluate the four polynomials of the current segment and store their sample, and the stopping lane's values
// seed the next chunk.  A serial replay on one lane cost ~190 shader cycles per sample (divergent-branch
// control flow around three compares); the chunked walk costs ~20.
// The heading goes through the reference's quaternion round trip, atan2(2wz, 1-2z^2) with
// w = cos(y/2), z = sin(y/2) (eth_mav_msgs/common.h:130-140), which is the wrap of y to (-pi, pi];
// it is computed as y - 2 pi rint(y / 2 pi) (agreement 1e-15; same value at the +-pi seam).
// A trajectory longer than the caller's capacity is reported as capacity + 1 samples ("too long": the
// nodelet rejects such results by its length check, src/mrs_trajectory_generation.cpp:1178-1199); this
// bounds the walk, which matters because a path whose outer loop ended on a rejected trial point
// can come back with segment times of thousands of seconds.

template <int NDER>
__global__ __launch_bounds__(64) void sample_kernel(BatchView b, const double* __restrict__ coeffs,
                                                    const double* __restrict__ seg_times, double dt, int capacity,
                                                    int32_t* __restrict__ n_samples, double* __restrict__ samples,
                                                    const double* __restrict__ acc_table, int acc_n) {
  extern __shared__ double s_T[];  // [max_segments] segment times | [S][4][10] coefficients of this path | sample buffer
  const int lane = threadIdx.x;
  double* s_c = s_T + b.max_segments;
  double* s_t = s_c + (size_t)b.max_segments * kD * kN;                         // [kSampleBuffer] time in segment
  unsigned short* s_seg = reinterpret_cast<unsigned short*>(s_t + kSampleBuffer);  // [kSampleBuffer] segment index
  // (a launch may carry fewer workgroups than paths: a workgroup then walks the paths q, q + gridDim.x, ...)
  for (int q = blockIdx.x; q < b.n_paths; q += gridDim.x) {
    const PathRef pr = path_at(b, q);
    const int S = pr.S;
    MRS_TG_PHASE_MARK(0);
    for (int i = lane; i < S; i += 64) s_T[i] = seg_times[pr.s0 + i];
    if (samples) {  // one coalesced pass instead of a global round trip per segment inside the walk
      const double* __restrict__ cg = coeffs + (size_t)pr.s0 * kD * kN;
      for (int e = lane; e < S * kD * kN; e += 64) s_c[e] = cg[e];
    }
    __syncthreads();
    MRS_TG_PHASE_MARK(1);
    double* out = samples ? samples + (size_t)pr.p * capacity * (NDER + 1) * kD : nullptr;
    const int n = sample_path_walk<NDER>(s_T, s_c, s_t, s_seg, S, dt, capacity, out, acc_table, acc_n);
    MRS_TG_PHASE_MARK(2);
    if (lane == 0 && n_samples) n_samples[pr.p] = n;
    __syncthreads();  // (the next path's staging overwrites what this walk read)
  }
}

// ---------------------------------------------------------------------------------------------
// The sampler of large launches (round 6): a GROUP of G = 8 or 16 lanes per path, 64 / G paths per wavefront, walk and
// evaluation fused.  sample_kernel spends a whole wavefront on a walk that is a scalar, sequential algorithm: per path ~13
// chunks, each with ~150 instructions of bookkeeping on all 64 lanes in front of the next (8 % of its wave cycles had a VALU
// instruction in flight: profiles/round5_pmc_sq_nonlinear_65536.csv).  Here the same bookkeeping serves 64 / G paths at once:
// per step lane j of a group adds dt to the group's time in segment j times (the reference's running sum, bit for bit), one
// ballot finds in every group the first lane at which the walk stops (segment end, trajectory end, capacity), the lanes before
// it evaluate the four polynomials of their path's segment (coefficients in LDS) and store their sample, and the stopping
// lane's value seeds the group's next step; groups whose time has run over their segment carry it into the next one in a
// predicated loop.  Same samples and counts as sample_kernel to the last bit
// (tests/test_gpu_large_batches.py::test_separate_sampler_equals_the_sampler_in_the_solve_kernels_tail, test_gpu_round6.py).
// (The round-5 verdict's proposal -- a lane-per-path walk that emits chunk descriptors + a sample-per-lane evaluation -- was
// built first and measured 2.4 x SLOWER than sample_kernel at 65536 x 10: profiles/round6_sampler_two_kernel_ab.txt.)
// LDS of a group: its path's segment times [Smax] and a RING of three segments' coefficients (the segment being walked, the
// next one, and the slot the one after next is written to when the walk moves on).  With all S segments of every path staged
// (26 KB per wavefront of eight 10-segment paths) a CU held six wavefronts and the SIMDs waited 62 % of their cycles
// (profiles/round6_pmc_sampler.txt); the ring is 1 KB per path whatever its length.
#ifndef MRS_TG_SAMPLE_EXP
#define MRS_TG_SAMPLE_EXP 0   // experiment builds of the group sampler (python -m ...build --variant NAME -DMRS_TG_SAMPLE_EXP=n)
#endif
constexpr int kRingSlots = 3;
__host__ __device__ constexpr int sample_group_ring_stride() { return kRingSlots * kD * kN + 10; }  // 130 doubles: 2 (mod 32), bank spread

template <int G>
__device__ __forceinline__ double group_lane_value(double v, int base, int idx) {  // lane base + idx of this wavefront
  return __shfl(v, base + idx, 64);
}

template <int G, int NDER>
__global__ __launch_bounds__(64, 4) void sample_group_kernel(BatchView b, const double* __restrict__ coeffs,
                                                             const double* __restrict__ seg_times, double dt, int capacity,
                                                             int32_t* __restrict__ n_samples, double* __restrict__ samples,
                                                             const double* __restrict__ acc_table, int acc_n) {
  constexpr int P = 64 / G;                       // paths per wavefront
  constexpr int kPer = (kD * kN + G - 1) / G;     // coefficients a lane moves when its group's ring advances
  extern __shared__ double lds[];                 // [P][Smax] segment times | [P] rings
  const int lane = threadIdx.x, g = lane / G, j = lane % G, base = g * G;
  const int Smax = b.max_segments;
  const int q = blockIdx.x * P + g;
  const bool active = q < b.n_paths;
  const PathRef pr = path_at(b, active ? q : b.n_paths - 1);
  const int S = pr.S;
  double* s_T = lds + (size_t)g * Smax;
  double* s_ring = lds + (size_t)P * Smax + (size_t)g * sample_group_ring_stride();
  const double* __restrict__ cg = coeffs + (size_t)pr.s0 * (kD * kN);
  if (b.uniform_S > 0) {  // the wavefront's paths are one contiguous run of segments: coalesced over all 64 lanes
    const int q0 = blockIdx.x * P;
    const int np = min(P, b.n_paths - q0);
    const double* __restrict__ tg = seg_times + (size_t)q0 * S;
    for (int e = lane; e < np * S; e += 64) lds[e] = tg[e];
  } else if (active) {
    for (int e = j; e < S; e += G) s_T[e] = seg_times[pr.s0 + e];
  }
  __syncthreads();
  // every lane of a group carries the group's walk state (i, Ti, tin, n): t_end and the start segment as the reference
  // computes them (trajectory.cpp:100-120, t_start = 0); N = #{k : A[k] < t_end} from the accumulated-time table
  // (mrs_tg_sampling.hpp: "accumulated < t_end" for sample k is "k < N")
  double t_end = 0.0;
  for (int i = 0; i < S; ++i) t_end += s_T[i];
  const double inv_dt = 1.0 / dt;
  int n_total = 0;
  if (t_end == t_end) {  // (a t_end that is not a number ends the reference's loop at once: no sample)
    int k = (int)fmin(fmax(t_end * inv_dt - 1.0, 0.0), (double)(acc_n - 1));
    while (k > 0 && acc_table[k - 1] >= t_end) --k;
    while (k < acc_n && acc_table[k] < t_end) ++k;   // the first k with A[k] >= t_end
    n_total = (k >= acc_n) ? 0x3fffffff : k;         // (none in the table: more samples than any buffer it was built for holds)
  }
  int i = 0;
  {
    double cum = 0.0;
    for (i = 0; i < S; ++i) {
      cum += s_T[i];
      if (cum > 0.0) break;
    }
  }
  bool done = !active || i >= S;
  const bool eval = samples != nullptr;
  // the ring: segments i and i + 1 into their slots (slot = segment mod 3), segment i + 2 requested into registers
  double pre[kPer];
  auto request = [&](int seg) {  // this lane's share of a segment's coefficients, global -> registers (no wait here)
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int e = j * kPer + u;
      pre[u] = (eval && !done && seg < S && e < kD * kN) ? cg[(size_t)seg * (kD * kN) + e] : 0.0;
    }
  };
  auto deposit = [&](int seg) {  // ... registers -> the segment's slot
    double* slot = s_ring + (seg % kRingSlots) * (kD * kN);
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int e = j * kPer + u;
      if (e < kD * kN) slot[e] = pre[u];
    }
  };
  request(i);
  deposit(i);
  request(i + 1);
  deposit(i + 1);
  request(i + 2);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  double dt_r[G];
#pragma unroll
  for (int r = 1; r < G; ++r) dt_r[r] = (j >= r) ? dt : 0.0;
  double tin = 0.0;  // (t_start = 0: the walk enters the first segment of positive length at its start)
  double Ti = done ? 0.0 : s_T[i];
  int n = 0;
  double* out = (eval && active) ? samples + (size_t)pr.p * capacity * (NDER + 1) * kD : nullptr;
  for (;;) {
    done = done || n >= n_total;  // trajectory.cpp:131
    // carry the remainder into the next segment(s) (:132-139), group by group; a group that moves on deposits the segment
    // after next (requested one segment ago) into the slot the segment it leaves behind no longer needs, and requests another
    while (__ballot(!done && tin > Ti) != 0ull) {
      if (!done && tin > Ti) {
        tin = tin - Ti;
        ++i;
        if (i >= S) {
          done = true;
        } else {
          Ti = s_T[i];
          deposit(i + 1);
          request(i + 2);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");   // (LDS only: over every address space the fence
      __builtin_amdgcn_wave_barrier();                                    //  waits for the previous step's sample stores as well)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
    if (__ballot(!done) == 0ull) break;
    // lane j adds dt j times (the addends dt_r = j >= r ? dt : +0.0 are set up once: x + 0.0 is x, bit for bit, for x >= 0)
    double tj = tin;
#pragma unroll
    for (int r = 1; r < G; ++r) tj += dt_r[r];
    const bool ok = !done && (n + j < n_total) && !(tj > Ti) && (n + j <= capacity);
    const unsigned long long ball = __ballot(ok);
    const unsigned bits = (unsigned)(ball >> base) & ((1u << G) - 1u);
    const int m = (bits == (1u << G) - 1u) ? G : __builtin_ctz(~bits);  // lanes [0, m) of the group emit a sample
    if (out && j < m && n + j < capacity) {
      const double* c = s_ring + (i % kRingSlots) * (kD * kN);
      double* o = out + (size_t)(n + j) * (NDER + 1) * kD;
#pragma unroll
      for (int k = 0; k <= NDER; ++k) {
        double v[kD];
#pragma unroll
        for (int dd = 0; dd < kD; ++dd) {
#if MRS_TG_SAMPLE_EXP == 2   // experiment: no evaluation (no LDS coefficient reads, no Horner): the walk and the stores only
          v[dd] = tj + (double)dd;
          (void)c;
#else
          double accv = falling_factorial(kN - 1, k) * c[dd * kN + kN - 1];
#pragma unroll
          for (int jj = kN - 2; jj >= k; --jj) accv = accv * tj + falling_factorial(jj, k) * c[dd * kN + jj];
          v[dd] = accv;
#endif
        }
        if (k == 0) v[3] = wrap_heading(v[3]);
#if MRS_TG_SAMPLE_EXP == 1   // experiment: the samples are computed and not stored
#pragma unroll
        for (int dd = 0; dd < kD; ++dd) asm volatile("" ::"v"(v[dd]));
        (void)o;
#else
        {  // (streaming stores, as sample_kernel's: mrs_tg_sampling.hpp)
          typedef double sample_pair __attribute__((ext_vector_type(2)));
          sample_pair* o2 = reinterpret_cast<sample_pair*>(o + k * kD);
          sample_pair lo, hi;
          lo.x = v[0], lo.y = v[1], hi.x = v[2], hi.y = v[3];
          __builtin_nontemporal_store(lo, o2);
          __builtin_nontemporal_store(hi, o2 + 1);
        }
#endif
      }
    }
    if (!done) {
#if MRS_TG_SAMPLE_EXP == 3   // experiment: no cross-lane exchange (a wrong walk: every step advances by G samples' worth of time)
      tin = tin + (double)(m > 0 ? m : 1) * dt;
#else
      if (m == G) {  // every lane emitted: the step ran out before the walk stopped
        tin = group_lane_value<G>(tj, base, G - 1) + dt;
      } else {       // lane m is the first that did not emit: its value is the walk's state at the stop
        tin = group_lane_value<G>(tj, base, m);
      }
#endif
      n += m;
      if (n > capacity) done = true;  // overflow: reported as capacity + 1
    }
  }
  if (active && j == 0 && n_samples) n_samples[pr.p] = n;
}

// ---------------------------------------------------------------------------------------------
// launchers

static inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

hipError_t launch_assemble(const BatchView& b, int d, const double* seg_times, double* H, double* Ainv,
                           hipStream_t stream) {
  if (b.n_segments == 0) return hipSuccess;
  if (b.uniform_S > 0) {
    dim3 grid(cdiv(b.n_paths, kAssembleChunk) * kN * b.uniform_S);
    MRS_TG_LAUNCH_TIMED(assemble_blocks_uniform_kernel, grid, dim3(kAssembleChunk), 0, stream, b.n_paths, b.uniform_S, d,
                        seg_times, H, Ainv);
  } else {
    dim3 grid(cdiv(b.n_paths, 256), kN, b.max_segments);
    MRS_TG_LAUNCH_TIMED(assemble_blocks_kernel, grid, dim3(256), 0, stream, b, d, seg_times, H, Ainv);
  }
  return hipGetLastError();
}

// four lanes per path shorten the dependency chain ~4x but repeat the factorisation: worth it until the
// lanes of the 1-lane variant alone fill the machine
static inline bool use_split_dims(int n_paths) { return n_paths <= 32768; }

hipError_t launch_solve_linear(const BatchView& b, int d, bool fused, const uint8_t* mask, const double* vals,
                               const double* seg_times, const double* H, const double* Ainv, double* ws,
                               double* coeffs, int32_t* status, double* cost, const int32_t* status_in,
                               hipStream_t stream, const double* pos_wp) {
  if (b.n_paths == 0) return hipSuccess;
  if (fused && ws != nullptr && quad_kernel_applies(b, b.n_paths, false)) {  // saturated device: four lanes per path, factors in LDS
    RowsTail tail;
    tail.pos_wp = pos_wp;
    return launch_solve_quad(b, d, mask, vals, seg_times, coeffs, status, cost, status_in, ws, stream, tail);
  }
  if (fused && rows_kernel_applies(b))
    return launch_solve_rows(b, d, mask, vals, seg_times, coeffs, status, cost, status_in, stream);
  if (tile_kernel_applies(b, fused))
    return launch_solve_tile(b, d, fused, mask, vals, seg_times, H, Ainv, coeffs, status, cost, status_in, stream);
  // (the one-lane-per-path solve from materialised blocks -- 512 registers and 264 bytes of scratch per lane -- is gone: a
  // blocks solve of a batch the tile kernel does not take runs four lanes per path whatever its size)
  if (use_split_dims(b.n_paths) || !fused) {
    dim3 grid(cdiv((long long)b.n_paths * 4, 64));
    if (fused)
      MRS_TG_LAUNCH_TIMED((solve_linear_kernel<1, true>), grid, dim3(64), 0, stream, b, d, mask, vals, seg_times, H, Ainv,
                         ws, coeffs, status, cost, status_in);
    else
      MRS_TG_LAUNCH_TIMED((solve_linear_kernel<1, false>), grid, dim3(64), 0, stream, b, d, mask, vals, seg_times, H,
                         Ainv, ws, coeffs, status, cost, status_in);
  } else {
    dim3 grid(cdiv(b.n_paths, 64));
    MRS_TG_LAUNCH_TIMED((solve_linear_kernel<4, true>), grid, dim3(64), 0, stream, b, d, mask, vals, seg_times, H, Ainv,
                        ws, coeffs, status, cost, status_in);
  }
  return hipGetLastError();
}

// blockIdx.y = entry of the list; 16-byte words where both ends are 16-byte aligned (every array of the ABI in practice),
// narrower words otherwise; grid-stride, so that a few hundred workgroups keep enough PCIe reads / writes in flight
__global__ __launch_bounds__(256) void copy_many_kernel(CopyList cl) {
  const int e = blockIdx.y;
  const char* src = static_cast<const char*>(cl.src[e]);
  char* dst = static_cast<char*>(cl.dst[e]);
  const unsigned long long bytes = cl.bytes[e];
  const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  const unsigned long long align = (unsigned long long)(uintptr_t)src | (unsigned long long)(uintptr_t)dst;
  unsigned long long done = 0;
  if ((align & 15ull) == 0) {
    const unsigned long long n = bytes >> 4;
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    for (unsigned long long i = tid; i < n; i += stride) d4[i] = s4[i];
    done = n << 4;
  } else if ((align & 7ull) == 0) {
    const unsigned long long n = bytes >> 3;
    const unsigned long long* s8 = reinterpret_cast<const unsigned long long*>(src);
    unsigned long long* d8 = reinterpret_cast<unsigned long long*>(dst);
    for (unsigned long long i = tid; i < n; i += stride) d8[i] = s8[i];
    done = n << 3;
  } else if ((align & 3ull) == 0) {
    const unsigned long long n = bytes >> 2;
    const unsigned* s32 = reinterpret_cast<const unsigned*>(src);
    unsigned* d32 = reinterpret_cast<unsigned*>(dst);
    for (unsigned long long i = tid; i < n; i += stride) d32[i] = s32[i];
    done = n << 2;
  }
  for (unsigned long long i = done + tid; i < bytes; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void copy_samples_kernel(const double* __restrict__ src, double* __restrict__ dst,
                                                           const int32_t* __restrict__ n_samples, int n_paths, int capacity) {
  for (int p = blockIdx.x; p < n_paths; p += gridDim.x) {
    const int n = min(n_samples[p], capacity);
    const size_t base = (size_t)p * (size_t)capacity * kD;
    const double2* s2 = reinterpret_cast<const double2*>(src + base);  // a row of four doubles = two 16-byte words
    double2* d2 = reinterpret_cast<double2*>(dst + base);
    for (int e = threadIdx.x; e < 2 * n; e += blockDim.x) d2[e] = s2[e];
  }
}

hipError_t launch_copy_samples(const double* src, double* dst, const int32_t* n_samples, int n_paths, int capacity,
                               hipStream_t stream) {
  if (n_paths <= 0 || capacity <= 0) return hipSuccess;
  MRS_TG_LAUNCH(copy_samples_kernel, dim3((unsigned)(n_paths < 2048 ? n_paths : 2048)), dim3(256), 0, stream, src, dst,
                     n_samples, n_paths, capacity);
  return hipGetLastError();
}

hipError_t launch_copy_many(const CopyList& cl, hipStream_t stream) {
  if (cl.n == 0) return hipSuccess;
  unsigned long long most = 0;
  for (int i = 0; i < cl.n; ++i) most = cl.bytes[i] > most ? cl.bytes[i] : most;
  // one 16-byte word per thread up to 512 workgroups, grid-stride beyond
  unsigned blocks = (unsigned)((most / 16 + 255) / 256);
  blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
  MRS_TG_LAUNCH(copy_many_kernel, dim3(blocks, (unsigned)cl.n), dim3(256), 0, stream, cl);
  return hipGetLastError();
}

hipError_t launch_estimate_times(const BatchView& b, const double* wp, const double* limits, double* seg_times,
                                 hipStream_t stream) {
  if (b.n_segments == 0) return hipSuccess;
  MRS_TG_LAUNCH(estimate_times_kernel, dim3(cdiv(b.n_segments, 256)), dim3(256), 0, stream, b, wp, limits,
                     seg_times);
  return hipGetLastError();
}

// The accumulated time of the reference's sampling walk, A[k] = k additions of dt to 0 (mrs_tg_sampling.hpp).  One lane makes
// the additions the reference makes, in its order (IEEE double, round to nearest: v_add_f64) -- a dependent chain of n
// additions, ~20 us for 4096 entries, once per (device, dt).
__global__ void sample_acc_table_kernel(double* __restrict__ table, int n, double dt) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double acc = 0.0;
  for (int k = 0; k < n; ++k) {
    table[k] = acc;
    acc += dt;
  }
}

// Cache of those tables, per (device, bit pattern of dt).  Everything a call does here is ordered on ITS stream and legal
// under stream capture once the table exists (the first version ran hipMalloc + a synchronous hipMemcpy inside
// the launch path, never evicted, and leaked outgrown tables):
//   * a table is built by a kernel on the calling stream, from a block of the library's pool; an event recorded behind the
//     build is what a launch on ANOTHER stream waits for (hipStreamWaitEvent) until the build has been seen complete once;
//   * at most kAccCacheMax tables are kept; the least recently used one is retired when a new dt arrives, and a table that
//     is outgrown (a larger capacity) is retired too.  A retired table may still be read by a launch in flight on some
//     stream, so it is parked; when kAccRetiredMax tables are parked the device is synchronised ONCE and all of them go back
//     to the pool -- a host that derives a new dt per request pays one device synchronisation per 64 requests and holds at
//     most 96 small tables;
//   * sample_tables_release() (the last context of the process is destroyed) frees everything.
namespace {
struct AccBlock {  // a table's memory: lives until no cache entry names it AND no launch that was handed it is still to be enqueued
  double* d = nullptr;
  int device = 0;
  int pins = 0;  // (under g_acc_mu) callers between sample_acc_table's return and the enqueue of their kernel
};
struct AccEntry {
  AccBlock* blk = nullptr;
  int n = 0;
  hipEvent_t ready = nullptr;  // recorded behind the build
  hipStream_t built_on = nullptr;
  bool ready_seen = false;
  unsigned long long stamp = 0;
};
constexpr size_t kAccCacheMax = 32, kAccRetiredMax = 64;
std::mutex g_acc_mu;
std::map<std::pair<int, unsigned long long>, AccEntry> g_acc_cache;
std::vector<AccBlock*> g_acc_retired;
unsigned long long g_acc_clock = 0;

void acc_retire(AccEntry& en) {
  if (en.ready) (void)hipEventDestroy(en.ready);
  if (en.blk) g_acc_retired.push_back(en.blk);
  en = AccEntry{};
}

// (g_acc_mu held) After the synchronisations no ENQUEUED launch reads a parked table; a table some caller still holds a pin
// on -- handed out, its kernel not enqueued yet -- stays parked for the next drain.  A device synchronisation is illegal while
// the calling stream records a graph: the parked list then simply grows until a call outside a capture drains it.
void acc_drain_retired(hipStream_t stream) {
  if (g_acc_retired.empty()) return;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (stream != nullptr && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return;
  (void)hipGetLastError();
  int cur = 0;
  (void)hipGetDevice(&cur);
  int last = -1;
  std::vector<AccBlock*> kept;
  for (AccBlock* r : g_acc_retired) {
    if (r->pins > 0) {
      kept.push_back(r);
      continue;
    }
    if (r->device != last) {
      (void)hipSetDevice(r->device);
      (void)hipDeviceSynchronize();
      last = r->device;
    }
    pool_free(r->d);
    delete r;
  }
  g_acc_retired.swap(kept);
  (void)hipSetDevice(cur);
}
}  // namespace

AccPin::~AccPin() {
  if (!block) return;
  std::lock_guard<std::mutex> lock(g_acc_mu);
  --static_cast<AccBlock*>(block)->pins;
}

void sample_tables_release() {
  std::lock_guard<std::mutex> lock(g_acc_mu);
  for (auto& kv : g_acc_cache) acc_retire(kv.second);
  g_acc_cache.clear();
  acc_drain_retired(nullptr);
}

hipError_t sample_acc_table(double dt, int capacity, hipStream_t stream, const double** table_out, int* n_out, AccPin* pin) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  unsigned long long bits;
  static_assert(sizeof(bits) == sizeof(dt), "dt as a key");
  std::memcpy(&bits, &dt, sizeof(bits));
  const int need = (capacity > 0 ? capacity : 0) + 80;
  std::lock_guard<std::mutex> lock(g_acc_mu);
  const auto key = std::make_pair(dev, bits);
  auto it = g_acc_cache.find(key);
  if (it == g_acc_cache.end() || it->second.n < need) {
    if (it == g_acc_cache.end()) {
      if (g_acc_cache.size() >= kAccCacheMax) {  // retire the least recently used table
        auto lru = g_acc_cache.begin();
        for (auto j = g_acc_cache.begin(); j != g_acc_cache.end(); ++j)
          if (j->second.stamp < lru->second.stamp) lru = j;
        acc_retire(lru->second);
        g_acc_cache.erase(lru);
      }
      it = g_acc_cache.emplace(key, AccEntry{}).first;
    } else {
      acc_retire(it->second);  // outgrown
    }
    if (g_acc_retired.size() >= kAccRetiredMax) acc_drain_retired(stream);
    AccEntry& en = it->second;
    int n = 1024 + 80;
    while (n < need) n *= 2;  // (growth by doubling: a caller that raises its capacity step by step rebuilds log2 times)
    en.blk = new (std::nothrow) AccBlock();
    if (!en.blk) {
      g_acc_cache.erase(it);
      return hipErrorOutOfMemory;
    }
    en.blk->device = dev;
    if ((e = pool_alloc(&en.blk->d, sizeof(double) * (size_t)n)) != hipSuccess) {
      delete en.blk;
      g_acc_cache.erase(it);
      return e;
    }
    en.n = n;
    MRS_TG_LAUNCH(sample_acc_table_kernel, dim3(1), dim3(64), 0, stream, en.blk->d, n, dt);
    if ((e = hipGetLastError()) == hipSuccess) e = hipEventCreateWithFlags(&en.ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(en.ready, stream);
    if (e != hipSuccess) {
      acc_retire(en);
      g_acc_cache.erase(it);
      return e;
    }
    en.built_on = stream;
  }
  AccEntry& en = it->second;
  en.stamp = ++g_acc_clock;
  if (!en.ready_seen) {
    if (hipEventQuery(en.ready) == hipSuccess) {
      en.ready_seen = true;
    } else if (stream != en.built_on) {  // (the building stream itself is ordered behind the build)
      (void)hipGetLastError();
      if ((e = hipStreamWaitEvent(stream, en.ready, 0)) != hipSuccess) return e;
    }
  }
  *table_out = en.blk->d;
  *n_out = en.n;
  if (pin && !pin->block) {  // the caller's launch is not enqueued yet: the block may be retired meanwhile, not recycled
    pin->block = en.blk;
    ++en.blk->pins;
  }
  return hipSuccess;
}

static size_t sample_group_lds_bytes(int Smax, int G) {
  return (size_t)(64 / G) * ((size_t)Smax + sample_group_ring_stride()) * sizeof(double);
}

// Which sampler a launch takes: 0 = one wavefront per path (sample_kernel), 8 / 16 = sample_group_kernel with that many lanes
// per path.  The group kernels need 64 / G paths' coefficients in LDS (40 KB at most: four wavefronts per CU and more) and
// enough paths to fill the SIMDs with 64 / G of them per wavefront.  MRS_TG_SAMPLE_GROUP=0 | 8 | 16 forces.
int sample_group_lanes(const BatchView& b) {
  const char* env = std::getenv("MRS_TG_SAMPLE_GROUP");  // (read at every call: the tests run all three kernels)
  const int forced = env ? std::atoi(env) : -1;
  auto fits = [&](int G) { return sample_group_lds_bytes(b.max_segments, G) <= 40 * 1024; };
  if (forced == 0) return 0;
  if (forced == 8 || forced == 16) return fits(forced) ? forced : 0;
  (void)fits;
  return 0;  // measured slower than sample_kernel at every size so far (profiles/round6_sampler_group_ab.txt): on request only
}

template <int NDER>
static hipError_t launch_sample_n(const BatchView& b, const double* coeffs, const double* seg_times, double dt, int capacity,
                                  int32_t* n_samples, double* samples, hipStream_t stream) {
  const double* acc_table = nullptr;
  int acc_n = 0;
  AccPin pin;  // (released when this function returns: behind the enqueue of the kernel that reads the table)
  if (!dry_run()) {  // (a dry run builds no table: an entry whose build kernel was not enqueued must not enter the cache)
    hipError_t et = sample_acc_table(dt, capacity, stream, &acc_table, &acc_n, &pin);
    if (et != hipSuccess) return et;
  }
  const int G = sample_group_lanes(b);
  if (G != 0) {
    const int P = 64 / G;
    const size_t lds_g = sample_group_lds_bytes(b.max_segments, G);
    const unsigned grid = (unsigned)((b.n_paths + P - 1) / P);
    if (G == 8) {
      note_kernel(NDER == 0 ? "sample_group_kernel<8, 0>" : "sample_group_kernel<8, NDER>");
      if (!dry_run()) hipLaunchKernelGGL((sample_group_kernel<8, NDER>), dim3(grid), dim3(64), lds_g, stream, b, coeffs, seg_times, dt, capacity,
                         n_samples, samples, acc_table, acc_n);
    } else {
      note_kernel(NDER == 0 ? "sample_group_kernel<16, 0>" : "sample_group_kernel<16, NDER>");
      if (!dry_run()) hipLaunchKernelGGL((sample_group_kernel<16, NDER>), dim3(grid), dim3(64), lds_g, stream, b, coeffs, seg_times, dt, capacity,
                         n_samples, samples, acc_table, acc_n);
    }
    return hipGetLastError();
  }
  const size_t lds = sizeof(double) * ((size_t)b.max_segments * (1 + kD * kN) + kSampleBuffer) + sizeof(unsigned short) * kSampleBuffer;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)sample_kernel<NDER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  // one workgroup per path.  (Fewer, persistent workgroups that walk several paths each -- the kernel's loop allows it --
  // were measured in round 5: 65536 x 10 pipeline 1328 -> 1347 / 1385 / 1384 us with 15360 / 7680 / 3840 workgroups.)
  MRS_TG_LAUNCH(sample_kernel<NDER>, dim3((unsigned)b.n_paths), dim3(64), lds, stream, b, coeffs, seg_times, dt, capacity,
                     n_samples, samples, acc_table, acc_n);
  return hipGetLastError();
}

hipError_t launch_sample(const BatchView& b, const double* coeffs, const double* seg_times, double dt, int capacity,
                         int32_t* n_samples, double* samples, hipStream_t stream) {
  if (b.n_paths == 0) return hipSuccess;
  return launch_sample_n<0>(b, coeffs, seg_times, dt, capacity, n_samples, samples, stream);
}

hipError_t launch_sample_states(const BatchView& b, const double* coeffs, const double* seg_times, double dt, int capacity,
                                int32_t* n_samples, double* states, hipStream_t stream) {
  if (b.n_paths == 0) return hipSuccess;
  return launch_sample_n<kSampleStateOrders - 1>(b, coeffs, seg_times, dt, capacity, n_samples, states, stream);
}

__global__ void position_mismatch_kernel(int n_vertices, const double* __restrict__ wp, const uint8_t* __restrict__ mask,
                                         const double* __restrict__ vals, unsigned long long* __restrict__ count) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  bool bad = false;
  if (v < n_vertices) {
    bad = mask[(size_t)v * kHalf] == 0;
#pragma unroll
    for (int q = 0; q < kD; ++q)  // bitwise: the solve must not depend on which array it read
      bad = bad || __double_as_longlong(wp[(size_t)v * kD + q]) != __double_as_longlong(vals[(size_t)v * kHalf * kD + q]);
  }
  const unsigned long long n = __popcll(__ballot(bad));
  if ((threadIdx.x & 63) == 0 && n != 0) atomicAdd(count, n);
}

hipError_t count_position_mismatches(const BatchView& b, const double* wp, const uint8_t* mask, const double* vals,
                                     hipStream_t stream, long long* count_out) {
  *count_out = 0;
  const int n_vertices = b.n_segments + b.n_paths;
  if (n_vertices == 0) return hipSuccess;
  unsigned long long* d_count = nullptr;
  hipError_t e = pool_alloc(&d_count, sizeof(unsigned long long));
  if (e != hipSuccess) return e;
  unsigned long long h = 0;
  e = hipMemsetAsync(d_count, 0, sizeof(unsigned long long), stream);
  if (e == hipSuccess) {
    MRS_TG_LAUNCH(position_mismatch_kernel, dim3(cdiv(n_vertices, 256)), dim3(256), 0, stream, n_vertices, wp, mask, vals, d_count);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(&h, d_count, sizeof(h), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  pool_free(d_count);
  *count_out = (long long)h;
  return e;
}

size_t linear_workspace_doubles(const BatchView& b) {
  // worst case: four lanes per path, ws_per_vertex<1>() doubles each
  return (size_t)b.max_segments * ws_per_vertex<1>() * 4 * (size_t)b.n_paths;
}

}  // namespace mrs_tg
