// mrs_tg_estimate.hpp -- the segment-time initialisation of one segment: estimateSegmentTimesEuclidean
// (/root/reference/src/eth_trajectory_generation/vertex.cpp:491-565).  Shared by estimate_times_kernel (mrs_tg_kernels.hip) and by
// the outer-loop kernel that starts from the estimate without a launch of its own in front (optimize_wave_kernel,
// mrs_tg_wave.hip); contraction is off inside so that both give the same bits.
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>

namespace mrs_tg {

__device__ __forceinline__ double wrap_pi(double a) {
  const double two_pi = 2.0 * M_PI;
  double r = fmod(a + M_PI, two_pi);
  if (r < 0) r += two_pi;
  return r - M_PI;
}

__device__ __forceinline__ double angle_dist(double a, double bb) {
  const double two_pi = 2.0 * M_PI;
  double dlt = wrap_pi(a) - wrap_pi(bb);
  if (dlt < -M_PI) dlt += two_pi;
  else if (dlt >= M_PI) dlt -= two_pi;
  return fabs(dlt);
}

// s: the segment's start waypoint (x, y, z, heading), the end waypoint behind it; lim: the path's nine limits
__device__ __forceinline__ double estimate_segment_time(const double* __restrict__ s, const double* __restrict__ lim) {
#pragma clang fp contract(off)
  const double* e = s + 4;
  const double v_h = lim[0], v_v = lim[1], w_max = lim[2], a_max = lim[5];
  const double dx = e[0] - s[0], dy = e[1] - s[1], dz = e[2] - s[2];
  const double inclinator = atan2(dz, sqrt(dx * dx + dy * dy));
  const double thr = atan2(v_v, v_h);
  const double vmax = (inclinator > thr || inclinator < -thr) ? fabs(v_v / sin(inclinator)) : fabs(v_h / cos(inclinator));
  double t = sqrt(dx * dx + dy * dy + dz * dz) / vmax;
  if (t < 0.01) t = 0.01;
  const double ang = angle_dist(s[3], e[3]);
  double t_vel = 0.0, t_acc = 0.0;
  if (w_max < (double)FLT_MAX && a_max < (double)FLT_MAX) {
    const double reduced = (ang - (w_max * w_max) / a_max) / w_max;
    t_vel = (reduced < 0) ? ang / w_max : reduced;
    if (ang > M_PI / 4) t_acc = 2 * (w_max / a_max);
  }
  const double hf = 1.5 * (t_vel + t_acc);
  if (hf > t) t = hf;
  return t;
}

}  // namespace mrs_tg
