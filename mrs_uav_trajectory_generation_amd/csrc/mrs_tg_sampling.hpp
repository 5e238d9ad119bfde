// mrs_tg_sampling.hpp -- the sampler's walk over one path whose segment times and coefficients sit in LDS (shared by
// sample_kernel and by the tail of solve_rows_kernel).
//
// Trajectory::evaluateRange's accumulate-and-carry walk
// (/root/reference/src/eth_trajectory_generation/trajectory.cpp:93-151), positions + wrapped heading (the only fields the
// nodelet reads, src/mrs_trajectory_generation.cpp:1582-1599); see the comment in mrs_tg_kernels.hip.
#pragma once
#include "mrs_tg_device.hpp"

#ifndef MRS_TG_SAMPLE_EXP
#define MRS_TG_SAMPLE_EXP 0   // experiment builds of the samplers (python -m ...build --variant NAME -DMRS_TG_SAMPLE_EXP=n; HISTORY.md)
#endif

namespace mrs_tg {

__device__ __forceinline__ double wrap_heading(double y) {
  const double two_pi_hi = 6.283185307179586232e+00, two_pi_lo = 2.449293598294706414e-16;
  const double kf = rint(y * 1.591549430918953456e-01);
  return fma(-kf, two_pi_lo, fma(-kf, two_pi_hi, y));
}

// j!/(j-k)!, a compile-time constant wherever j and k are
__host__ __device__ constexpr double falling_factorial(int j, int k) {
  double v = 1.0;
  for (int n = 0; n < k; ++n) v *= (double)(j - n);
  return v;
}

__device__ __forceinline__ double lane_value(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// The walk and the evaluation are separate passes.  The walk produces, chunk by chunk, the time in its segment of every
// sample (lane j of a chunk holds the j-th one) and parks it with its segment index in an LDS buffer; nothing else sits
// on the walk's dependent chain.  When the buffer is full (or the walk has ended) all buffered samples are evaluated at
// once, one lane per sample whatever its segment -- with the evaluation inside the chunk loop (one segment per chunk,
// ~30 of 64 lanes busy, 40 broadcast LDS reads and the Horner chains in front of the next chunk's additions) the kernel
// took 25 us for 1024 paths of ~300 samples; 1024 x 10 nonlinear 147 -> us.
constexpr int kSampleBuffer = 192;  // samples parked per flush.  1024 made the LDS block of a 10-segment path 13.5 KB (11 wavefronts
                                    // per CU); 192: 5.2 KB, 30 per CU -- 8192 x 10: 50 -> 40 us (scripts/tile_phases.hip), 1024 x 10 unchanged

// One wavefront (all 64 lanes, wave-uniform control flow).  s_T [S] segment times, s_c [S][4][10] coefficients, s_t / s_seg
// the sample buffer (kSampleBuffer entries each); out: the path's [capacity][4] samples or nullptr (count only).
// Returns the number of samples (capacity + 1 = "more than fit").
// NDER = 0: out[sample][4] positions + wrapped heading (what the nodelet reads).  NDER = 4: out[sample][5][4], the
// derivative orders 0..4 of every dimension -- sampleTrajectoryInRange's position / velocity / acceleration / jerk / snap
// and yaw / yaw rate / yaw acceleration (trajectory_sampling.cpp:49-104), each evaluated as Polynomial::evaluate(t, k)
// does: Horner over j!/(j-k)! c_j (polynomial.h:150-163).
//
// The ACCUMULATED time of the reference's walk (`accumulated_time += dt` from 0, trajectory.cpp:131-149: the loop runs while
// it is below the trajectory's end) does not depend on the path: sample k is taken at A[k] = (...((0 + dt) + dt)...) + dt, k
// additions, whatever the segments are.  The launcher hands the kernel that sequence (acc_table[0 .. acc_n), computed once per
// dt with the same IEEE additions, sample_acc_table), the walk finds N = #{k : A[k] < t_end} with one look at the table around
// t_end / dt, and "accumulated < t_end" for sample k is "k < N": the second of the two dependent additions per sample that the
// walk used to carry is gone, bit for bit the same samples.
template <int NDER = 0>
__device__ __forceinline__ int sample_path_walk(const double* s_T, const double* s_c, double* s_t, unsigned short* s_seg, int S,
                                                double dt, int capacity, double* out, const double* __restrict__ acc_table,
                                                int acc_n) {
  const int lane = threadIdx.x & 63;
  // every lane carries the same walk state (i, Ti, tin, n): t_end and the start segment as the reference
  // computes them (trajectory.cpp:100-120, t_start = 0)
  double t_end = 0.0;
  for (int i = 0; i < S; ++i) t_end += s_T[i];
  const double inv_dt = 1.0 / dt;
  // N = the first k with A[k] >= t_end (A increases strictly): a window of 64 table entries around t_end / dt, moved if the
  // estimate was off; no entry at or above t_end in the table = more samples than any buffer this table was built for holds
  int n_total = 0;
  if (t_end == t_end) {  // (a t_end that is not a number ends the reference's loop at once: no sample)
    const double est = fmin(fmax(t_end * inv_dt, 0.0), (double)(acc_n - 1));
    int base = max((int)est - 8, 0);
    for (;;) {
      const int k = min(base + lane, acc_n - 1);
      const unsigned long long ge = __ballot(acc_table[k] >= t_end);
      if (ge == 0ull) {
        if (base + 64 >= acc_n) {
          n_total = 0x3fffffff;
          break;
        }
        base += 56;
      } else if ((ge & 1ull) && base > 0) {
        base = max(base - 56, 0);
      } else {
        n_total = base + __builtin_ctzll(ge);
        break;
      }
    }
  }
  int i = 0;
  {
    double acc = 0.0;
    for (i = 0; i < S; ++i) {
      acc += s_T[i];
      if (acc > 0.0) break;
    }
  }
  int n = 0;
  int n_flushed = 0;  // samples [n_flushed, n) are parked in the buffer
  auto flush = [&](int upto) {
    if (!out) return;
    // (the fences order LDS only: over every address space they wait for the samples stored by the previous flush as well)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    for (int e = n_flushed + lane; e < upto; e += 64) {
      if (e >= capacity) break;
      const double tj = s_t[e - n_flushed];
      const double* c = s_c + (size_t)s_seg[e - n_flushed] * (kD * kN);
#pragma unroll
      for (int k = 0; k <= NDER; ++k) {
        double v[kD];
#pragma unroll
        for (int dd = 0; dd < kD; ++dd) {
#if MRS_TG_SAMPLE_EXP == 12   // experiment: no evaluation (no coefficient reads, no Horner)
          v[dd] = tj + (double)dd;
          (void)c;
#else
          double accv = falling_factorial(kN - 1, k) * c[dd * kN + kN - 1];
#pragma unroll
          for (int j = kN - 2; j >= k; --j) accv = accv * tj + falling_factorial(j, k) * c[dd * kN + j];
          v[dd] = accv;
#endif
        }
        if (k == 0) v[3] = wrap_heading(v[3]);
#if MRS_TG_SAMPLE_EXP == 11   // experiment: the samples are computed and not stored
#pragma unroll
        for (int dd = 0; dd < kD; ++dd) asm volatile("" ::"v"(v[dd]));
#elif MRS_TG_SAMPLE_EXP == 14  // experiment: ordinary (cached) stores, as until round 5
#pragma unroll
        for (int dd = 0; dd < kD; ++dd) out[((size_t)e * (NDER + 1) + k) * kD + dd] = v[dd];
#else
        // STREAMING stores (round 6): the samples are written once and read by nobody on the device; as ordinary stores they
        // went through the L2 like data that will be used again -- 65536 x 10: 327 -> 275 us with the stores marked
        // non-temporal, 8192 x 10 46.4 -> 44.4 (profiles/round6_sampler_experiments.txt)
        {
          typedef double sample_pair __attribute__((ext_vector_type(2)));
          sample_pair* o2 = reinterpret_cast<sample_pair*>(out + ((size_t)e * (NDER + 1) + k) * kD);
          sample_pair lo, hi;
          lo.x = v[0], lo.y = v[1], hi.x = v[2], hi.y = v[3];
          __builtin_nontemporal_store(lo, o2);
          __builtin_nontemporal_store(hi, o2 + 1);
        }
#endif
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  };
  if (i < S) {
    double tin = 0.0;  // (t_start = 0: the walk enters the first segment of positive length at its start)
    double Ti = s_T[i];
    while (true) {  // trajectory.cpp:131-150, one chunk per iteration
      if (n >= n_total) break;
      bool past_end = false;
      while (tin > Ti) {  // carry the remainder into the next segment(s)
        tin = tin - Ti;
        ++i;
        if (i >= S) {
          past_end = true;
          break;
        }
        Ti = s_T[i];
      }
      if (past_end) break;
      // lanes 0..last are computed in this chunk; `last` only has to be non-negative: an underestimate splits the
      // segment into several chunks, an overestimate adds idle additions
      // (an estimate is all that is needed, so a multiplication by 1/dt stands in for the division: a dependent chain of
      // ~40 instructions per chunk less)
      const double room = (Ti - tin) * inv_dt;
      const int last = __builtin_amdgcn_readfirstlane((room < 61.0) ? (int)room + 2 : 63);
      // lane j adds dt j times: in iteration r the lanes above r add.  EXEC starts as "lanes 1..63" and is shifted left by
      // one lane per iteration, so an iteration is one addition and one scalar shift; as a lane compare and a select per
      // iteration the walk was three times as long
      double tj = tin;
      {
        unsigned long long saved_exec;
        int counter;
        const double dtv = dt;
        // (iterations beyond `last` only touch lanes above `last`, which are not used: the count is rounded up to the unrolling)
#define MRS_TG_WALK_STEP "v_add_f64 %[tj], %[tj], %[dt]\n\ts_lshl_b64 exec, exec, 1\n\t"
        asm volatile(
            "s_mov_b64 %[save], exec\n\t"
            "s_add_u32 %[cnt], %[n], 7\n\t"
            "s_lshr_b32 %[cnt], %[cnt], 3\n\t"
            "s_cmp_eq_u32 %[cnt], 0\n\t"
            "s_cbranch_scc1 .Lwalk_done_%=\n\t"
            "s_mov_b64 exec, -2\n"
            ".Lwalk_loop_%=:\n\t" MRS_TG_WALK_STEP MRS_TG_WALK_STEP MRS_TG_WALK_STEP MRS_TG_WALK_STEP MRS_TG_WALK_STEP
                MRS_TG_WALK_STEP MRS_TG_WALK_STEP MRS_TG_WALK_STEP
            "s_sub_u32 %[cnt], %[cnt], 1\n\t"
            "s_cmp_lg_u32 %[cnt], 0\n\t"
            "s_cbranch_scc1 .Lwalk_loop_%=\n"
            ".Lwalk_done_%=:\n\t"
            "s_mov_b64 exec, %[save]"
            : [tj] "+v"(tj), [save] "=&s"(saved_exec), [cnt] "=&s"(counter)
            : [dt] "v"(dtv), [n] "s"(last)
            : "scc");
#undef MRS_TG_WALK_STEP
      }
      const bool ok = (lane <= last) && (n + lane < n_total) && !(tj > Ti) && (n + lane <= capacity);
      const unsigned long long okmask = __ballot(ok);
      const int m = (~okmask == 0ull) ? 64 : __builtin_ctzll(~okmask);  // lanes [0, m) emit a sample
      if (out) {
        if (n + m - n_flushed > kSampleBuffer) {  // the chunk does not fit: evaluate what is parked first
          flush(n);
          n_flushed = n;
        }
        if (lane < m) {
          s_t[n - n_flushed + lane] = tj;
          s_seg[n - n_flushed + lane] = (unsigned short)i;
        }
      }
      n += m;
      if (m == last + 1) {  // every computed lane emitted: the chunk ran out before the walk stopped
        tin = lane_value(tj, last) + dt;
        if (n > capacity) break;
        continue;
      }
      // lane m is the first that did not emit: its value is the walk's state at the stop
      tin = lane_value(tj, m);
      if (n > capacity) break;  // overflow: report capacity + 1
    }
  }
  flush(n);
  return n;
}

}  // namespace mrs_tg
