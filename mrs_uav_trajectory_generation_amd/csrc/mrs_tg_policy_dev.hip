// mrs_tg_policy_dev.hip -- the policy layer's per-round work that scales with the batch, on the device (round 6).
//
// MrsTrajectoryGeneration::optimize() (/root/reference/src/mrs_trajectory_generation.cpp:620-851) re-solves a request up to seven
// times; around every solve it builds the vertices (:923-977) and scans the sampled trajectory against the waypoint polyline
// (validateTrajectorySpatial, :1401-1455).  For a batch of requests the host used to build [vertex][5][4] value arrays that
// are 80 % zeros, send them up, bring every path's samples down, and scan them on 16 threads: 19-24 of the 51 ms of 4096
// requests were those transfers, 13 the scans.  Here
//   policy_expand_kernel    one lane per VERTEX: constraint mask and values from (unwrapped waypoint, flags, initial state);
//                           what travels up is 36 bytes per vertex instead of 197;
//   policy_validate_kernel  one lane per PATH: the nodelet's gate on the optimiser's code (:1138-1149), the length check
//                           against the Baca total (:1178-1199), then validateTrajectorySpatial as written -- a sequential
//                           scan with a waypoint cursor -- on the samples where they are; what travels down is a few words
//                           per path, one byte per segment, and the samples of the paths that are FINISHED.
// The arithmetic of the scan is the host's, operation by operation (no contraction into fused multiply-adds: the decisions
// `distance > max_deviation` and the reported maximum must be the bits mrs_tg_policy_host.hpp::validate_spatial produces;
// tests/test_gpu_policy.py compares the two routes).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "mrs_tg_launch.h"
#include "mrs_tg_policy_host.hpp"

namespace mrs_tg {

// vinfo[v] = path index (position in the round's batch) << 4 | flags
__global__ __launch_bounds__(256) void policy_expand_kernel(int n_vertices, int d, const double* __restrict__ wp,
                                                            const int32_t* __restrict__ vinfo, const double* __restrict__ init,
                                                            uint8_t* __restrict__ mask, double* __restrict__ vals) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= n_vertices) return;
  const int info = vinfo[v];
  const int a = info >> 4;
  const bool first = info & kVertexFirst, last = info & kVertexLast, stop = info & kVertexStop, has_init = info & kVertexInit;
  uint8_t m[5] = {1, 0, 0, 0, 0};
  double val[20];
#pragma unroll
  for (int e = 0; e < 20; ++e) val[e] = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) val[k] = wp[(size_t)v * 4 + k];
  if (first || last) {  // makeStartOrEnd(., d): derivatives 1 .. d at rest (:940-976)
    for (int k = 1; k <= d; ++k) m[k] = 1;
    if (first && has_init) {  // the initial state's velocity / acceleration / jerk (:946-957)
      m[1] = m[2] = m[3] = 1;
#pragma unroll
      for (int e = 0; e < 12; ++e) val[4 + e] = init[(size_t)a * 12 + e];
    }
  } else if (stop) {  // a stop_at waypoint (:969-973)
    m[1] = m[2] = m[3] = 1;
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) mask[(size_t)v * 5 + k] = m[k];
  double2* out = reinterpret_cast<double2*>(vals + (size_t)v * 20);
#pragma unroll
  for (int e = 0; e < 10; ++e) out[e] = make_double2(val[2 * e], val[2 * e + 1]);
}

namespace {

// distFromSegment (:1533-1554) with the host's operations in the host's order
__device__ __forceinline__ double dist_from_segment_dev(const double* p, const double* s1, const double* s2) {
#pragma clang fp contract(off)
  const double sv0 = s2[0] - s1[0], sv1 = s2[1] - s1[1], sv2 = s2[2] - s1[2];
  const double len = sqrt(sv0 * sv0 + sv1 * sv1 + sv2 * sv2);
  double n0 = sv0, n1 = sv1, n2 = sv2;
  if (len * len > 0) {
    n0 /= len;
    n1 /= len;
    n2 /= len;
  }
  const double d0 = p[0] - s1[0], d1 = p[1] - s1[1], d2 = p[2] - s1[2];
  const double coord = n0 * d0 + n1 * d1 + n2 * d2;
  if (coord < 0) return sqrt(d0 * d0 + d1 * d1 + d2 * d2);
  if (coord > len) {
    const double e0 = p[0] - s2[0], e1 = p[1] - s2[1], e2 = p[2] - s2[2];
    return sqrt(e0 * e0 + e1 * e1 + e2 * e2);
  }
  const double f0 = p[0] - (s1[0] + n0 * coord), f1 = p[1] - (s1[1] + n1 * coord), f2 = p[2] - (s1[2] + n2 * coord);
  return sqrt(f0 * f0 + f1 * f1 + f2 * f2);
}

}  // namespace

__global__ __launch_bounds__(64) void policy_validate_kernel(PolicyValidateArgs g) {
#pragma clang fp contract(off)
  const int a = blockIdx.x * 64 + threadIdx.x;
  if (a >= g.n_paths) return;
  const int s0 = g.seg_offsets[a], S = g.seg_offsets[a + 1] - s0, n_wp = S + 1;
  const int ns = g.n_samples[a], st = g.status[a];
  bool ok = (st >= 1 && st != 6) || st == -1;  // :1138-1149
  if (ok) {                                     // :1178-1199
    const double len = (double)ns * g.dt;
    if (len > 1.0 && ((g.max_len_factor > 0 && len > g.max_len_factor * g.baca_total[a]) ||
                      (g.min_len_factor > 0 && len < g.min_len_factor * g.baca_total[a])))
      ok = false;
  }
  if (ns > g.capacity) ok = false;
  bool is_safe = true;
  double max_dev = 0.0;
  uint8_t* safe = g.safe_out + s0;
  if (ok && !g.last_round) {  // validateTrajectorySpatial :1401-1455 (the last re-solve is not validated again, :729)
    for (int i = 0; i < S; ++i) safe[i] = 1;
    const double* wps = g.wp + (size_t)(s0 + a) * 4;   // (positions: x, y, z of a waypoint are what the scan reads)
    const double* smp = g.samples + (size_t)a * g.capacity * 4;
    int widx = 0;
    for (int i = 0; i + 1 < ns; ++i) {
      const double* sample = smp + (size_t)i * 4;
      const double* next = sample + 4;
      const double* w0 = wps + (size_t)widx * 4;
      const double* w1 = w0 + 4;
      const double d_seg = dist_from_segment_dev(sample, w0, w1);
      const double d_end = dist_from_segment_dev(w1, sample, next);
      if (widx > 0 || g.first_segment || n_wp <= 2) {
        if (d_seg > max_dev) max_dev = d_seg;
        if (d_seg > g.max_deviation) {
          safe[widx] = 0;
          is_safe = false;
        }
      }
      if (d_end < 0.05 && widx < n_wp - 2) ++widx;
    }
  }
  const bool done = !ok || g.last_round || !(g.check_enabled && !is_safe);
  g.ok_out[a] = ok ? 1 : 0;
  g.ns_out[a] = ns;
  g.status_out[a] = st;
  g.max_dev_out[a] = max_dev;
  g.is_safe_out[a] = is_safe ? 1 : 0;
  g.ns_copy[a] = (ok && done) ? (ns < g.capacity ? ns : g.capacity) : 0;  // rows that travel: the finished paths' samples
}

hipError_t launch_policy_expand(int n_vertices, int d, const double* wp, const int32_t* vinfo, const double* init, uint8_t* mask,
                                double* vals, hipStream_t stream) {
  if (n_vertices <= 0) return hipSuccess;
  MRS_TG_LAUNCH(policy_expand_kernel, dim3((unsigned)((n_vertices + 255) / 256)), dim3(256), 0, stream, n_vertices, d, wp, vinfo, init,
                mask, vals);
  return hipGetLastError();
}

hipError_t launch_policy_validate(const PolicyValidateArgs& args, hipStream_t stream) {
  if (args.n_paths <= 0) return hipSuccess;
  MRS_TG_LAUNCH(policy_validate_kernel, dim3((unsigned)((args.n_paths + 63) / 64)), dim3(64), 0, stream, args);
  return hipGetLastError();
}

}  // namespace mrs_tg
