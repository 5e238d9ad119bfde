// mrs_tg_policy_host.hpp -- the PURE-HOST half of the path-policy layer: what MrsTrajectoryGeneration::optimize() does around
// findTrajectory() (/root/reference/src/mrs_trajectory_generation.cpp:620-851), for a batch of independent paths, with the
// solver behind a callback:
//   preprocessPath (:431-500) -> vertices as findTrajectory builds them (:923-977) -> host.solve(all active paths of the
//   round) -> nlopt-code gate (:1138-1149) -> length sanity check against the Baca estimate (:1048-1056, :1178-1199) ->
//   validateTrajectorySpatial (:1401-1455) -> mid-points into unsafe segments (:739-753) -> next round; optional
//   findTrajectoryFallback (:1215-1395) and override_heading_atan2 (:1582-1597).
// No HIP type or call appears here: mrs_tg_policy.hip instantiates optimize_paths() with the batched GPU solve and
// mrs_tg_abi.hip's mrs_tg_find_trajectory uses the vertex builder, the Baca total and the two gates; the same header compiles
// with g++ and is driven by tests/host/policy_host_harness.cpp with the CPU oracle as the solver on 16 threads under
// ASan / UBSan / TSan (tests/test_host_sanitizers.py).  ROS-only branches (tf, stamps, "path from the future", MPC
// prediction splicing) have no counterpart.
#pragma once

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <stdexcept>
#include <system_error>
#include <thread>
#include <type_traits>
#include <vector>

#include <sched.h>

#include "../../include/mrs_tg.h"

struct mrs_tg_ctx;

namespace mrs_tg {

// ---- one round of the policy loop on the device (mrs_tg_abi.hip::policy_round_device, kernels in mrs_tg_policy_dev.hip) ----
// The host block of a round: inputs | results, every field 256-byte aligned.  Plain arithmetic, shared by the host code that
// fills / reads the block and the device code that mirrors it in its arena.
struct PolicyRoundLayout {
  size_t wp, init, lim, baca, vinfo, so, in_bytes;                      // inputs (offsets in the block)
  size_t ok, ns, status, max_dev, is_safe, safe, samples, total_bytes;   // results
};
inline PolicyRoundLayout policy_round_layout(size_t A, size_t nS, int32_t capacity) {
  auto up = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
  const size_t nV = nS + A;
  PolicyRoundLayout L{};
  size_t off = 0;
  L.wp = off, off += up(nV * 4 * sizeof(double));          // unwrapped waypoints [vertex][4]
  L.init = off, off += up(A * 12 * sizeof(double));        // initial velocity / acceleration / jerk [path][3][4]
  L.lim = off, off += up(A * 9 * sizeof(double));          // limits after relax_heading [path][9]
  L.baca = off, off += up(A * sizeof(double));             // initial_total_time_baca [path]
  L.vinfo = off, off += up(nV * sizeof(int32_t));          // position of the vertex's path << 4 | kVertex* flags
  L.so = off, off += up((A + 1) * sizeof(int32_t));        // segment offsets
  L.in_bytes = off;
  L.ok = off, off += up(A * sizeof(int32_t));              // both gates passed (and the samples fit)
  L.ns = off, off += up(A * sizeof(int32_t));
  L.status = off, off += up(A * sizeof(int32_t));
  L.max_dev = off, off += up(A * sizeof(double));
  L.is_safe = off, off += up(A);
  L.safe = off, off += up(nS ? nS : 1);                    // validateTrajectorySpatial's flag of every segment
  L.samples = off, off += up(A * (size_t)capacity * 4 * sizeof(double));  // rows of the FINISHED paths only
  L.total_bytes = off;
  return L;
}
constexpr int kVertexFirst = 1, kVertexLast = 2, kVertexStop = 4, kVertexInit = 8;
struct PolicyRoundIn {
  int32_t n_paths;
  const int32_t* seg_offsets;  // [n_paths + 1] (host; also at block + layout.so)
  size_t n_segments, n_vertices;
  char* block;                 // policy_round_layout(n_paths, n_segments, sample_capacity).total_bytes
  mrs_tg_options opt;
  double max_len_factor, min_len_factor, max_deviation;
  int32_t first_segment, check_enabled, last_round, sample_capacity;
};
int policy_round_device(::mrs_tg_ctx* ctx, const PolicyRoundIn& in);  // (defined in mrs_tg_abi.hip: the GPU build only)

namespace policy {

inline double wrap_range(double a, double lo, double range) {
  double r = std::fmod(a - lo, range);
  if (r < 0) r += range;
  return r + lo;
}
// mrs_lib radians::diff / radians::interp (angles in [0, 2 pi)), sradians::unwrap
inline double radians_diff(double minuend, double subtrahend) {
  const double two_pi = 2.0 * M_PI;
  double d = wrap_range(minuend, 0.0, two_pi) - wrap_range(subtrahend, 0.0, two_pi);
  if (d < -M_PI) d += two_pi;
  else if (d >= M_PI) d -= two_pi;
  return d;
}
inline double radians_interp(double from, double to, double coeff) {
  return wrap_range(from + coeff * radians_diff(to, from), 0.0, 2.0 * M_PI);
}
inline double sradians_unwrap(double what, double from) {
  const double two_pi = 2.0 * M_PI;
  double d = wrap_range(what, -M_PI, two_pi) - wrap_range(from, -M_PI, two_pi);
  if (d < -M_PI) d += two_pi;
  else if (d >= M_PI) d -= two_pi;
  return from + d;
}
// the yaw a sample carries after EigenTrajectoryPoint::setFromYaw / getYaw (the nodelet reads it back at :1599):
// quaternionFromYaw = (cos(yaw / 2), 0, 0, sin(yaw / 2)), yawFromQuaternion = atan2(2 (w z + x y), 1 - 2 (y^2 + z^2))
// (include/eth_mav_msgs/common.h:130-140) -- the round trip's own arithmetic, not atan2(sin, cos), which differs in the last bit
inline double wrap_yaw(double y) {
  const double w = std::cos(y * 0.5), z = std::sin(y * 0.5);
  return std::atan2(2.0 * (w * z), 1.0 - 2.0 * (z * z));
}

inline double dist_from_segment(const double* p, const double* s1, const double* s2) {  // :1533-1554
  const double sv[3] = {s2[0] - s1[0], s2[1] - s1[1], s2[2] - s1[2]};
  const double len = std::sqrt(sv[0] * sv[0] + sv[1] * sv[1] + sv[2] * sv[2]);
  double n[3] = {sv[0], sv[1], sv[2]};
  if (len * len > 0) {
    n[0] /= len;
    n[1] /= len;
    n[2] /= len;
  }
  const double d1[3] = {p[0] - s1[0], p[1] - s1[1], p[2] - s1[2]};
  const double coord = n[0] * d1[0] + n[1] * d1[1] + n[2] * d1[2];
  if (coord < 0) return std::sqrt(d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2]);
  if (coord > len) {
    const double d2[3] = {p[0] - s2[0], p[1] - s2[1], p[2] - s2[2]};
    return std::sqrt(d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2]);
  }
  const double e[3] = {p[0] - (s1[0] + n[0] * coord), p[1] - (s1[1] + n[1] * coord), p[2] - (s1[2] + n[2] * coord)};
  return std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
}

inline void interpolate_point(const double* a, const double* b, double coeff, double* out) {  // :1612-1625
  for (int k = 0; k < 3; ++k) out[k] = a[k] + coeff * (b[k] - a[k]);
  out[3] = radians_interp(a[3], b[3], coeff);
}

inline double limit_for_inclination(double inclinator, double lim_v, double lim_h) {  // vertex.cpp:337-353
  if (inclinator > std::atan2(lim_v, lim_h) || inclinator < -std::atan2(lim_v, lim_h)) return std::fabs(lim_v / std::sin(inclinator));
  return std::fabs(lim_h / std::cos(inclinator));
}

inline void unit3(const double* a, const double* b, double* u) {
  double v[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
  const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (n * n > 0) {
    v[0] /= n;
    v[1] /= n;
    v[2] /= n;
  }
  u[0] = v[0];
  u[1] = v[1];
  u[2] = v[2];
}

// estimateSegmentTimesBaca, /root/reference/src/eth_trajectory_generation/vertex.cpp:301-485; wp [V][4] unwrapped
inline void estimate_times_baca(int S, const double* wp, const double* lim, std::vector<double>& out) {
  const double v_h = lim[0], v_v = lim[1], w_hdg = lim[2], a_h = lim[3], a_v = lim[4], a_hdg = lim[5], j_h = lim[6], j_v = lim[7];
  const int V = S + 1;
  out.assign(S, 0.0);
  for (int i = 0; i < S; ++i) {
    const double* s = wp + (size_t)i * 4;
    const double* e = s + 4;
    const double dx = e[0] - s[0], dy = e[1] - s[1], dz = e[2] - s[2];
    const double distance = std::sqrt(dx * dx + dy * dy + dz * dz);
    const double inclinator = std::atan2(dz, std::sqrt(dx * dx + dy * dy));
    const double v_max = limit_for_inclination(inclinator, v_v, v_h);
    const double a_max = limit_for_inclination(inclinator, a_v, a_h);
    const double j_max = limit_for_inclination(inclinator, j_v, j_h);
    double t1 = 0, t2 = 0;
    const double full = (v_max / a_max) + (a_max / j_max);
    if (i >= 1) {
      double u1[3], u2[3];
      unit3(wp + (size_t)(i - 1) * 4, s, u1);
      unit3(s, e, u2);
      const double dot = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
      t1 = (1 - (dot < 0 ? 0.0 : dot)) * full;
    }
    if (i == 0) t1 = full;
    if (i == V - 2) t2 = full;
    if (i < V - 2) {
      double u1[3], u2[3];
      unit3(s, e, u1);
      unit3(e, wp + (size_t)(i + 2) * 4, u2);
      const double dot = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
      t2 = (1 - (dot < 0 ? 0.0 : dot)) * full;
    }
    const double cap = std::sqrt(2 * distance / a_max);
    t1 = std::min(t1, cap);
    t2 = std::min(t2, cap);
    double t = distance / v_max + t1 + t2;
    if (t < 0.01) t = 0.01;
    // heading rotation time (:457-480)
    double dh = radians_diff(s[3], e[3]);
    const double ang = std::fabs(dh);
    double tv = 0, ta = 0;
    if (w_hdg < (double)FLT_MAX && a_hdg < (double)FLT_MAX) {
      const double reduced = (ang - 2 * (w_hdg * w_hdg) / a_hdg) / w_hdg;
      tv = (reduced < 0) ? ang / w_hdg : reduced;
      if (ang > M_PI / 4) ta = 2 * (w_hdg / a_hdg);
    }
    const double hf = 1.5 * (tv + ta);
    if (hf > t) t = hf;
    out[i] = t;
  }
}

// The policy's per-path host work (vertex building, Baca estimates, spatial validation, mid-point insertion) is independent
// from path to path: batches of requests run it on a few threads.  Ranges of [0, n) in order, one per thread; small batches
// (a nodelet's single request) stay on the calling thread.  MRS_TG_POLICY_THREADS=1 switches the threads off.
inline int policy_threads() {
  static const int n = [] {
    if (const char* e = std::getenv("MRS_TG_POLICY_THREADS")) return std::max(1, std::atoi(e));
    int cpus = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = std::min(cpus > 0 ? cpus : 1 << 20, CPU_COUNT(&set));
    return std::max(1, std::min(cpus, 16));
  }();
  return n;
}

// body(begin, end) over ranges of [0, n).  An exception thrown by any range (std::bad_alloc from a worker's vectors) is
// caught on the thread that raised it, every thread is joined, and the FIRST exception is rethrown on the calling thread --
// nothing escapes a std::thread (which would be std::terminate) and no joinable thread is ever destroyed.
template <class F>
void parallel_ranges(size_t n, size_t min_per_thread, F&& body) {
  // MRS_TG_POLICY_GRAIN=k (test knob, read once): k items are enough for a thread, whatever the call site asks for -- lets a
  // test put a few dozen requests on 16 threads (tests/host/policy_host_harness.cpp)
  static const size_t grain_override = [] {
    const char* e = std::getenv("MRS_TG_POLICY_GRAIN");
    return e ? (size_t)std::max(1, std::atoi(e)) : (size_t)0;
  }();
  if (grain_override) min_per_thread = grain_override;
  const size_t threads = std::min<size_t>((size_t)policy_threads(), n / std::max<size_t>(min_per_thread, 1));
  if (threads <= 1) {
    body((size_t)0, n);
    return;
  }
  const size_t chunk = (n + threads - 1) / threads;
  std::vector<std::exception_ptr> raised(threads);   // slot t: what range t threw (written by one thread each)
  auto guarded = [&body, &raised](size_t t, size_t b, size_t e) noexcept {
    try {
      body(b, e);
    } catch (...) {
      raised[t] = std::current_exception();
    }
  };
  struct Joiner {   // joins on every way out of this scope
    std::vector<std::thread> pool;
    ~Joiner() {
      for (std::thread& th : pool)
        if (th.joinable()) th.join();
    }
  } joiner;
  size_t first_inline = threads;  // ranges [first_inline, threads) run on this thread: a thread that could not be started
  try {
    joiner.pool.reserve(threads - 1);
    for (size_t t = 1; t < threads; ++t) {
      const size_t b = std::min(n, t * chunk), e = std::min(n, b + chunk);
      if (b >= e) continue;
      try {
        joiner.pool.emplace_back(guarded, t, b, e);
      } catch (const std::system_error&) {
        first_inline = t;
        break;
      }
    }
  } catch (const std::bad_alloc&) {  // (the pool's own vector): everything not started runs here
    first_inline = joiner.pool.size() + 1;
  }
  guarded(0, (size_t)0, std::min(n, chunk));
  for (size_t t = first_inline; t < threads; ++t) {
    const size_t b = std::min(n, t * chunk), e = std::min(n, b + chunk);
    if (b < e) guarded(t, b, e);
  }
  for (std::thread& th : joiner.pool) th.join();
  for (const std::exception_ptr& ex : raised)
    if (ex) std::rethrow_exception(ex);
}

struct PathState {
  std::vector<double> wps;     // [n][4] current waypoints (raw headings)
  std::vector<uint8_t> stop;
  int n_wp = 0;
  bool done = false, ok = false;
  int n_samples = 0, iterations = 0;
  double max_dev = 0.0;
  double baca_total = 0.0;
};

inline void preprocess(const mrs_tg_waypoint* in, int n_in, const mrs_tg_policy_options& o, PathState& st) {  // :431-500
  st.wps.clear();
  st.stop.clear();
  int last_added = 0;
  for (int i = 0; i < n_in; ++i) {
    const double* w = in[i].coords;
    if (o.path_straightener_enabled && n_in >= 3 && i > 0 && i < n_in - 1) {
      const double* first = in[last_added].coords;
      const double* last = in[i + 1].coords;
      bool segment_is_ok = true;
      for (int j = last_added + 1; j < i + 1; ++j) {
        const double* mid = in[j].coords;
        // quirk B3 of the reference: fabs() wraps the comparison, so the heading test is signed
        if (dist_from_segment(mid, first, last) > o.path_straightener_max_deviation ||
            (radians_diff(first[3], mid[3]) > o.path_straightener_max_hdg_deviation) ||
            (radians_diff(last[3], mid[3]) > o.path_straightener_max_hdg_deviation)) {
          segment_is_ok = false;
          break;
        }
      }
      if (segment_is_ok) continue;
    }
    if (i > 0 && i < n_in - 1) {
      const double* first = in[last_added].coords;
      const double dx = first[0] - w[0], dy = first[1] - w[1], dz = first[2] - w[2];
      if (std::sqrt(dx * dx + dy * dy + dz * dz) < o.min_waypoint_distance) continue;
    }
    st.wps.insert(st.wps.end(), w, w + 4);
    st.stop.push_back(in[i].stop_at);
    last_added = i;
  }
  st.n_wp = (int)st.stop.size();
}

// validateTrajectorySpatial :1401-1455
inline bool validate_spatial(const double* samples, int n_samples, const PathState& st, const mrs_tg_policy_options& o,
                      std::vector<uint8_t>& safe, double& max_dev) {
  const int n_wp = st.n_wp;
  safe.assign(std::max(n_wp - 1, 0), 1);
  int widx = 0;
  bool is_safe = true;
  max_dev = 0;
  for (int i = 0; i + 1 < n_samples; ++i) {
    const double* sample = samples + (size_t)i * 4;
    const double* next = sample + 4;
    const double* s0 = st.wps.data() + (size_t)widx * 4;
    const double* s1 = s0 + 4;
    const double d_seg = dist_from_segment(sample, s0, s1);
    const double d_end = dist_from_segment(s1, sample, next);
    if (widx > 0 || o.max_deviation_first_segment || n_wp <= 2) {
      if (d_seg > max_dev) max_dev = d_seg;
      if (d_seg > o.max_deviation) {
        safe[widx] = 0;
        is_safe = false;
      }
    }
    if (d_end < 0.05 && widx < n_wp - 2) ++widx;
  }
  return is_safe;
}

inline void insert_midpoints(PathState& st, const std::vector<uint8_t>& safe, const mrs_tg_policy_options& o) {  // :739-753
  int w = 0, sidx = 0;
  while (w < st.n_wp - 1) {
    if (!safe[sidx] && (w > 0 || o.max_deviation_first_segment || st.n_wp <= 2)) {
      double mid[4];
      interpolate_point(st.wps.data() + (size_t)w * 4, st.wps.data() + (size_t)(w + 1) * 4, 0.5, mid);
      st.wps.insert(st.wps.begin() + (size_t)(w + 1) * 4, mid, mid + 4);
      st.stop.insert(st.stop.begin() + (w + 1), (uint8_t)0);
      ++st.n_wp;
      ++w;
    }
    ++sidx;
    ++w;
  }
}

// findTrajectoryFallback :1215-1395
inline int fallback_sampling(const PathState& st, const double* limits9, bool relax_heading, const mrs_tg_policy_options& o, double dt,
                      double* out, int capacity) {
  const int n_wp = st.n_wp;
  std::vector<double> wps(st.wps);
  double last = wps[3];
  for (int i = 0; i < n_wp; ++i) {
    wps[(size_t)i * 4 + 3] = sradians_unwrap(st.wps[(size_t)i * 4 + 3], last);
    last = wps[(size_t)i * 4 + 3];
  }
  double lim[9];
  std::memcpy(lim, limits9, sizeof(lim));
  lim[0] *= o.fallback_speed_factor;
  lim[1] *= o.fallback_speed_factor;
  lim[3] *= o.fallback_accel_factor;
  lim[4] *= o.fallback_accel_factor;
  if (relax_heading) lim[2] = lim[5] = lim[8] = (double)FLT_MAX;
  std::vector<double> t_baca;
  estimate_times_baca(n_wp - 1, wps.data(), lim, t_baca);
  int count = 0;
  for (int i = 0; i < n_wp - 1; ++i) {
    int n_samples = 0;
    double step = 0;
    if (t_baca[i] > 1e-1) {
      n_samples = (int)std::ceil(t_baca[i] / dt);
      step = (n_samples > 0) ? 1.0 / (double)n_samples : 0.5;
    }
    if (n_samples > 0 && i == n_wp - 2) ++n_samples;
    for (int j = 0; j < n_samples; ++j) {
      double p[4];
      interpolate_point(st.wps.data() + (size_t)i * 4, st.wps.data() + (size_t)(i + 1) * 4, j * step, p);
      p[3] = wrap_yaw(p[3]);
      int repeat = 1;
      if (j == 0 && i > 0 && st.stop[i]) repeat += (int)std::round(o.fallback_stopping_time / dt);
      for (int r = 0; r < repeat; ++r) {
        if (count < capacity) std::memcpy(out + (size_t)count * 4, p, sizeof(p));
        ++count;
      }
    }
  }
  return count;
}

// ---- the single-path seam: the pieces of findTrajectory() around the solver -------------------------------------------

// the limits findTrajectory hands to the estimators and the optimiser: relax_heading lifts the heading limits (:1030-1038)
inline void effective_limits(const double* limits9, bool relax_heading, double* lim_out) {
  for (int k = 0; k < 9; ++k) lim_out[k] = limits9[k];
  if (relax_heading) lim_out[2] = lim_out[5] = lim_out[8] = (double)FLT_MAX;
}

// The vertices findTrajectory builds for one path (:923-977): headings unwrapped along the path from the initial state's
// heading (:925-936), every vertex constrains its position (:944, :963, :967), the ends are makeStartOrEnd(0, d) with the
// initial state's velocity / acceleration / jerk at the first one (:946-957), stop_at vertices pin derivatives 1..3 to
// zero (:969-973).  wps_raw [n_wp][4] raw headings; wp_out [n_wp][4], mask_out [n_wp][5], vals_out [n_wp][5][4].
inline void build_vertices(const double* wps_raw, const uint8_t* stop_at, int n_wp, const mrs_tg_initial_state* init, int d,
                           double* wp_out, uint8_t* mask_out, double* vals_out) {
  std::memset(vals_out, 0, sizeof(double) * 20 * (size_t)n_wp);
  std::memset(mask_out, 0, 5 * (size_t)n_wp);
  double last_heading = init ? init->heading : wps_raw[3];
  for (int i = 0; i < n_wp; ++i) {
    double* w = wp_out + (size_t)i * 4;
    for (int k = 0; k < 3; ++k) w[k] = wps_raw[(size_t)i * 4 + k];
    w[3] = sradians_unwrap(wps_raw[(size_t)i * 4 + 3], last_heading);
    last_heading = w[3];
    uint8_t* m = mask_out + (size_t)i * 5;
    double* vv = vals_out + (size_t)i * 20;
    m[0] = 1;
    for (int k = 0; k < 4; ++k) vv[k] = w[k];
    if (i == 0 || i == n_wp - 1) {
      for (int k = 1; k <= d; ++k) m[k] = 1;
      if (i == 0 && init) {
        m[1] = m[2] = m[3] = 1;
        for (int k = 0; k < 4; ++k) {
          vv[4 + k] = init->velocity[k];
          vv[8 + k] = init->acceleration[k];
          vv[12 + k] = init->jerk[k];
        }
      }
    } else if (stop_at && stop_at[i]) {
      m[1] = m[2] = m[3] = 1;
    }
  }
}

// initial_total_time_baca (:1048-1056): the sum of estimateSegmentTimesBaca over the path's vertices
inline double baca_total_time(int n_seg, const double* wp_unwrapped, const double* lim, std::vector<double>& scratch) {
  estimate_times_baca(n_seg, wp_unwrapped, lim, scratch);
  double tot = 0;
  for (double t : scratch) tot += t;
  return tot;
}

// the nodelet's gate on the optimiser's code (:1138-1149): >= 1 except 6 (MAXTIME), and -1
inline bool code_accepted(int status) { return (status >= 1 && status != 6) || status == -1; }

// the length sanity check (:1178-1199): 0 = passes, +1 = "too long", -1 = "too short".  Only trajectories longer than one
// second are checked; a factor <= 0 switches its side of the check off (the reference has no such switch: its parameters
// are always loaded, config/public/trajectory_generation.yaml:35-36)
inline int length_check(int n_samples, double dt, double baca_total, double max_factor, double min_factor) {
  const double len = (double)n_samples * dt;
  if (!(len > 1.0)) return 0;
  if (max_factor > 0 && len > max_factor * baca_total) return 1;
  if (min_factor > 0 && len < min_factor * baca_total) return -1;
  return 0;
}


// a Host with `int round(const PolicyRoundIn&)` and `bool device_round_enabled(size_t active_paths)` runs the rounds' batch-sized work on the device
template <class H, class = void>
struct HasDeviceRound : std::false_type {};
template <class H>
struct HasDeviceRound<H, std::void_t<decltype(&H::round)>> : std::true_type {};

template <class Host, class... Args>
int failf(Host& host, int code, const char* fmt, Args... args) {
  char buf[256];
  std::snprintf(buf, sizeof(buf), fmt, args...);
  return host.fail(code, buf);
}

// optimize() (:620-851) for n_paths independent requests.  Host supplies
//   void* scratch(size_t bytes)            a block the solver reads / writes cheaply (pinned), or nullptr: ordinary memory
//   int solve(A, seg_offsets, waypoints, mask, values, limits, &options, times, status, n_samples, samples)
//                                          mrs_tg_solve_batch's contract for the round's A active paths (samples only)
//   int fail(code, message)                records the message, returns the code
// May throw std::bad_alloc (the callers map it to MRS_TG_ERR_NOMEM).
template <class Host>
int optimize_paths(Host& host, int32_t n_paths, const int32_t* wp_offsets, const mrs_tg_waypoint* waypoints,
                   const mrs_tg_initial_state* initial_states, const uint8_t* has_initial_state, const double* limits,
                   const uint8_t* relax_heading, const mrs_tg_policy_options* opt, int32_t sample_capacity,
                   int32_t* success_out, int32_t* n_samples_out, double* samples_out, double* max_deviation_out,
                   int32_t* n_waypoints_out, int32_t* iterations_out) {
  if (!wp_offsets || !waypoints || !limits || !opt || !success_out || !n_samples_out || !samples_out)
    return host.fail(MRS_TG_ERR_INVALID_ARG, "wp_offsets, waypoints, limits, options, success_out, n_samples_out and samples_out are required");
  if (n_paths < 0 || sample_capacity <= 0)
    return failf(host, MRS_TG_ERR_INVALID_ARG, "n_paths %d / sample_capacity %d: need >= 0 / > 0", n_paths, sample_capacity);
  const mrs_tg_policy_options& o = *opt;
  const int d = o.solver.derivative_to_optimize;
  if (d < 2 || d > 4)
    return failf(host, MRS_TG_ERR_INVALID_ARG, "derivative_to_optimize must be 2, 3 or 4 (got %d)", d);
  const double dt = o.solver.sampling_dt;
  if (!(dt > 0)) return failf(host, MRS_TG_ERR_INVALID_ARG, "the policy layer needs sampling_dt > 0 (got %g)", dt);
  const auto t_begin = std::chrono::steady_clock::now();
  auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
  std::vector<PathState> st((size_t)n_paths);
  parallel_ranges((size_t)n_paths, 256, [&](size_t p0, size_t p1) {
    for (size_t p = p0; p < p1; ++p) {
      preprocess(waypoints + wp_offsets[p], wp_offsets[p + 1] - wp_offsets[p], o, st[p]);
      if (st[p].n_wp <= 1) {  // "the path is empty (after postprocessing)" :676-681
        st[p].done = true;
        st[p].ok = false;
      }
    }
  });
  std::vector<int> active;
  // MRS_TG_POLICY_TRACE=1: where the call's time went (host phases and the batched GPU call), on stderr
  static const bool trace = [] {
    const char* e = std::getenv("MRS_TG_POLICY_TRACE");
    return e != nullptr && std::atoi(e) != 0;
  }();
  double t_build = 0, t_solve = 0, t_post = 0, t_validate = 0;
  auto now = [&]() { return trace ? elapsed() : 0.0; };
  for (int round = 0; round <= o.max_deviation_iterations; ++round) {
    active.clear();
    for (int p = 0; p < n_paths; ++p)
      if (!st[p].done) active.push_back(p);
    if (active.empty()) break;
    // optimize() picks the solver of a round in this order (:702-716, :754-768): fallback sampling when it was asked for,
    // fallback sampling when overtime() says the request is running late ("executing fallback sampling, we are running
    // over time" -- the request still succeeds), else findTrajectory.  Only the checks BEHIND the solve (:1085, :1156,
    // :1171, :1516-1522) give a request up.
    double budget_left = 0.0;  // timeLeft() :1749-1761
    auto overtime = [&]() {    // overtime() :1730-1743 (OVERTIME_SAFETY_FACTOR 0.95, OVERTIME_SAFETY_OFFSET 0.01 s)
      return o.max_execution_time_s > 0 && elapsed() > 0.95 * o.max_execution_time_s - 0.01;
    };
    bool use_fallback = o.fallback_sampling != 0;
    if (!use_fallback && o.max_execution_time_s > 0) {
      const double spent = elapsed();
      budget_left = spent >= o.max_execution_time_s ? 0.0 : o.max_execution_time_s - spent;
      use_fallback = overtime();
    }
    // the requests are independent: one that cannot be solved (its deviation loop has subdivided it beyond the longest
    // path a plan takes; the reference has no such limit) fails on its own and leaves the others alone
    if (!use_fallback) {
      size_t kept = 0;
      for (int p : active) {
        if (st[p].n_wp - 1 > MRS_TG_MAX_SEGMENTS) {
          st[p].done = true;
          st[p].ok = false;
          st[p].n_samples = 0;
        } else {
          active[kept++] = p;
        }
      }
      active.resize(kept);
      if (active.empty()) break;
    }
    // ---- a host that runs the round on the device (the GPU build): vertices expanded, gates and validateTrajectorySpatial
    // applied where the samples are; the waypoints travel up, a few words per path and the FINISHED paths' samples come down
    if constexpr (HasDeviceRound<Host>::value) {
      if (!use_fallback && host.device_round_enabled(active.size())) {
        const double t0 = now();
        const size_t A = active.size();
        std::vector<int32_t> so(A + 1, 0);
        for (size_t a = 0; a < A; ++a) so[a + 1] = so[a] + st[active[a]].n_wp - 1;
        const size_t nS = (size_t)so.back(), nV = nS + A;
        const PolicyRoundLayout L = policy_round_layout(A, nS, sample_capacity);
        static const bool pinned_allowed = [] {  // MRS_TG_POLICY_PINNED=0: ordinary memory (test knob, read once per process)
          const char* e = std::getenv("MRS_TG_POLICY_PINNED");
          return e == nullptr || std::atoi(e) != 0;
        }();
        char* block = pinned_allowed ? static_cast<char*>(host.scratch(L.total_bytes)) : nullptr;
        std::vector<char> pageable;
        if (!block) {
          pageable.resize(L.total_bytes);
          block = pageable.data();
        }
        double* wp = reinterpret_cast<double*>(block + L.wp);
        double* init = reinterpret_cast<double*>(block + L.init);
        double* lim = reinterpret_cast<double*>(block + L.lim);
        double* baca = reinterpret_cast<double*>(block + L.baca);
        int32_t* vinfo = reinterpret_cast<int32_t*>(block + L.vinfo);
        std::memcpy(block + L.so, so.data(), sizeof(int32_t) * (A + 1));
        bool any_stop = false;
        for (size_t a = 0; a < A && !any_stop; ++a)
          for (int i = 1; i + 1 < st[active[a]].n_wp && !any_stop; ++i) any_stop = st[active[a]].stop[i] != 0;
        parallel_ranges(A, 128, [&](size_t a0, size_t a1) {
          std::vector<double> tb;
          for (size_t a = a0; a < a1; ++a) {
            const int p = active[a];
            const PathState& s = st[p];
            const bool has_init = has_initial_state && has_initial_state[p] && initial_states;
            const size_t v0 = (size_t)so[a] + a;
            // headings unwrapped along the path from the initial state's heading (:925-936)
            double last_heading = has_init ? initial_states[p].heading : s.wps[3];
            for (int i = 0; i < s.n_wp; ++i) {
              double* w = wp + (v0 + i) * 4;
              for (int k = 0; k < 3; ++k) w[k] = s.wps[(size_t)i * 4 + k];
              w[3] = sradians_unwrap(s.wps[(size_t)i * 4 + 3], last_heading);
              last_heading = w[3];
              vinfo[v0 + i] = (int32_t)(a << 4) | (i == 0 ? kVertexFirst : 0) | (i == s.n_wp - 1 ? kVertexLast : 0) |
                              ((i > 0 && i < s.n_wp - 1 && s.stop[i]) ? kVertexStop : 0) | ((i == 0 && has_init) ? kVertexInit : 0);
            }
            for (int k = 0; k < 4; ++k) {
              init[a * 12 + k] = has_init ? initial_states[p].velocity[k] : 0.0;
              init[a * 12 + 4 + k] = has_init ? initial_states[p].acceleration[k] : 0.0;
              init[a * 12 + 8 + k] = has_init ? initial_states[p].jerk[k] : 0.0;
            }
            effective_limits(limits + (size_t)p * 9, relax_heading && relax_heading[p], lim + a * 9);
            st[p].baca_total = baca[a] = baca_total_time(s.n_wp - 1, wp + v0 * 4, lim + a * 9, tb);
          }
        });
        PolicyRoundIn in{};
        in.n_paths = (int32_t)A;
        in.seg_offsets = so.data();
        in.n_segments = nS;
        in.n_vertices = nV;
        in.block = block;
        in.opt = o.solver;
        if (any_stop && d == 4) in.opt.flags |= MRS_TG_FLAG_CONSTRAINED_SLOTS;
        if (o.max_execution_time_s > 0) in.opt.max_time_s = 2.0 * 0.95 * budget_left;  // :899
        in.max_len_factor = o.max_trajectory_len_factor;
        in.min_len_factor = o.min_trajectory_len_factor;
        in.max_deviation = o.max_deviation;
        in.first_segment = o.max_deviation_first_segment;
        in.check_enabled = o.check_deviation_enabled;
        in.last_round = round == o.max_deviation_iterations;
        in.sample_capacity = sample_capacity;
        const double t1 = now();
        t_build += t1 - t0;
        const int rc = host.round(in);
        if (rc != MRS_TG_OK) return rc;
        const double t2 = now();
        t_solve += t2 - t1;
        const bool late = overtime();  // findTrajectory's own checks behind optimize() and the sampler: "return {}" (:1085, :1156, :1171)
        const int32_t* r_ok = reinterpret_cast<const int32_t*>(block + L.ok);
        const int32_t* r_ns = reinterpret_cast<const int32_t*>(block + L.ns);
        const double* r_dev = reinterpret_cast<const double*>(block + L.max_dev);
        const uint8_t* r_safe_path = reinterpret_cast<const uint8_t*>(block + L.is_safe);
        const uint8_t* r_safe = reinterpret_cast<const uint8_t*>(block + L.safe);
        const double* r_smp = reinterpret_cast<const double*>(block + L.samples);
        const bool last = in.last_round != 0;
        parallel_ranges(A, 128, [&](size_t a0, size_t a1) {
          std::vector<uint8_t> safe;
          for (size_t a = a0; a < a1; ++a) {
            const int p = active[a];
            const bool ok = !late && r_ok[a] != 0;
            st[p].ok = ok;
            st[p].n_samples = ok ? r_ns[a] : 0;
            if (!ok) {
              st[p].done = true;  // "failed to find trajectory" :720-727, :771-778
              continue;
            }
            bool finished = last;
            if (!last) {
              st[p].max_dev = r_dev[a];
              if (o.check_deviation_enabled && !r_safe_path[a]) {
                safe.assign(r_safe + so[a], r_safe + so[a + 1]);
                insert_midpoints(st[p], safe, o);
                st[p].iterations = round + 1;
              } else {
                st[p].done = true;
                finished = true;
              }
            }
            if (finished)
              std::memcpy(samples_out + (size_t)p * sample_capacity * 4, r_smp + a * (size_t)sample_capacity * 4,
                          sizeof(double) * 4 * (size_t)r_ns[a]);
          }
        });
        t_post += now() - t2;
        if (last) break;
        continue;
      }
    }
    // ---- solve every active path (one batched GPU call, or the fallback sampler on the host)
    if (use_fallback) {
      for (int p : active) {
        double* out = samples_out + (size_t)p * sample_capacity * 4;
        const int ns = fallback_sampling(st[p], limits + (size_t)p * 9, relax_heading && relax_heading[p], o, dt, out, sample_capacity);
        st[p].n_samples = ns;
        st[p].ok = ns <= sample_capacity;
        if (!st[p].ok) st[p].done = true;
      }
    } else {
      // vertices exactly as findTrajectory builds them (:923-977).  The arrays of the round live in ONE block of pinned
      // host memory kept by the context (no allocation, no page faults and no clearing of a 64 KB sample buffer per
      // request and round; the GPU reads and writes pinned arrays in place, and only the sample rows a path has produced
      // travel); the coefficients, which the policy never reads, stay on the device
      const double t0 = now();
      const size_t A = active.size();
      std::vector<int32_t> so(A + 1, 0);
      for (size_t a = 0; a < A; ++a) so[a + 1] = so[a] + st[active[a]].n_wp - 1;
      const size_t nS = (size_t)so.back(), nV = nS + A;
      auto up = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
      const size_t b_wp = up(nV * 4 * sizeof(double)), b_vals = up(nV * 20 * sizeof(double)), b_lim = up(A * 9 * sizeof(double)),
                   b_times = up(nS * sizeof(double)), b_smp = up(A * (size_t)sample_capacity * 4 * sizeof(double)),
                   b_status = up(A * sizeof(int32_t)), b_ns = up(A * sizeof(int32_t)), b_mask = up(nV * 5);
      const size_t need = b_wp + b_vals + b_lim + b_times + b_smp + b_status + b_ns + b_mask;
      static const bool pinned_allowed = [] {  // MRS_TG_POLICY_PINNED=0: ordinary memory (test knob, read once per process)
        const char* e = std::getenv("MRS_TG_POLICY_PINNED");
        return e == nullptr || std::atoi(e) != 0;
      }();
      char* block = pinned_allowed ? static_cast<char*>(host.scratch(need)) : nullptr;
      std::vector<char> pageable;  // (the runtime refused that much pinned memory: ordinary memory, copied by the runtime)
      if (!block) {
        pageable.resize(need);
        block = pageable.data();
      }
      double* wp = reinterpret_cast<double*>(block);
      double* vals = reinterpret_cast<double*>(block + b_wp);
      double* lim = reinterpret_cast<double*>(block + b_wp + b_vals);
      double* times = reinterpret_cast<double*>(block + b_wp + b_vals + b_lim);
      double* smp = reinterpret_cast<double*>(block + b_wp + b_vals + b_lim + b_times);
      int32_t* status = reinterpret_cast<int32_t*>(block + b_wp + b_vals + b_lim + b_times + b_smp);
      int32_t* ns = reinterpret_cast<int32_t*>(block + b_wp + b_vals + b_lim + b_times + b_smp + b_status);
      uint8_t* mask = reinterpret_cast<uint8_t*>(block + b_wp + b_vals + b_lim + b_times + b_smp + b_status + b_ns);
      parallel_ranges(A, 128, [&](size_t a0, size_t a1) {
        std::vector<double> tb;
        for (size_t a = a0; a < a1; ++a) {
          const int p = active[a];
          const PathState& s = st[p];
          const bool has_init = has_initial_state && has_initial_state[p] && initial_states;
          const size_t v0 = (size_t)so[a] + a;
          std::memset(times + so[a], 0, sizeof(double) * (size_t)(s.n_wp - 1));
          build_vertices(s.wps.data(), s.stop.data(), s.n_wp, has_init ? &initial_states[p] : nullptr, d, wp + v0 * 4, mask + v0 * 5,
                         vals + v0 * 20);
          effective_limits(limits + (size_t)p * 9, relax_heading && relax_heading[p], lim + a * 9);
          st[p].baca_total = baca_total_time(s.n_wp - 1, wp + v0 * 4, lim + a * 9, tb);
        }
      });
      mrs_tg_options so_opt = o.solver;
      so_opt.estimate_times = 1;
      so_opt.sample_capacity = sample_capacity;
      so_opt.flags |= MRS_TG_FLAG_REFERENCE_STATUS;  // the length check below is the reference's answer to a runaway (:1178-1199)
      if (o.max_execution_time_s > 0) so_opt.max_time_s = 2.0 * 0.95 * budget_left;  // :899
      const double t1 = now();
      t_build += t1 - t0;
      const int rc = host.solve((int32_t)A, so.data(), wp, mask, vals, lim, &so_opt, times, status, ns, smp);
      if (rc != MRS_TG_OK) return rc;
      const double t2 = now();
      t_solve += t2 - t1;
      const bool late = overtime();  // findTrajectory's own checks behind optimize() and the sampler: "return {}" (:1085, :1156, :1171)
      parallel_ranges(A, 128, [&](size_t a0, size_t a1) {
        for (size_t a = a0; a < a1; ++a) {
          const int p = active[a];
          bool ok = !late && code_accepted(status[a]);  // :1138-1149
          if (ok && length_check(ns[a], dt, st[p].baca_total, o.max_trajectory_len_factor, o.min_trajectory_len_factor) != 0) ok = false;  // :1178-1199
          if (ns[a] > sample_capacity) ok = false;
          st[p].ok = ok;
          st[p].n_samples = ok ? ns[a] : 0;
          if (!ok) {
            st[p].done = true;  // "failed to find trajectory" :720-727, :771-778
          } else {
            std::memcpy(samples_out + (size_t)p * sample_capacity * 4, smp + a * (size_t)sample_capacity * 4,
                        sizeof(double) * 4 * (size_t)ns[a]);
          }
        }
      });
      t_post += now() - t2;
    }
    if (round == o.max_deviation_iterations) break;  // the last re-solve is not validated again (:729)
    // ---- validate, subdivide the unsafe ones
    const double t3 = now();
    parallel_ranges(active.size(), 128, [&](size_t a0, size_t a1) {
      std::vector<uint8_t> safe;
      for (size_t a = a0; a < a1; ++a) {
        const int p = active[a];
        if (st[p].done) continue;
        double md = 0;
        const bool is_safe = validate_spatial(samples_out + (size_t)p * sample_capacity * 4, st[p].n_samples, st[p], o, safe, md);
        st[p].max_dev = md;
        if (o.check_deviation_enabled && !is_safe) {
          insert_midpoints(st[p], safe, o);
          st[p].iterations = round + 1;
        } else {
          st[p].done = true;
        }
      }
    });
    t_validate += now() - t3;
  }
  if (trace)
    std::fprintf(stderr, "mrs_tg_optimize_paths: %d requests, %.3f ms: vertices + estimates %.3f, mrs_tg_solve_batch %.3f, results %.3f, "
                 "validation + mid-points %.3f\n", n_paths, elapsed() * 1e3, t_build * 1e3, t_solve * 1e3, t_post * 1e3, t_validate * 1e3);
  for (int p = 0; p < n_paths; ++p) {
    const PathState& s = st[p];
    success_out[p] = s.ok ? 1 : 0;
    n_samples_out[p] = s.ok ? s.n_samples : 0;
    if (max_deviation_out) max_deviation_out[p] = s.max_dev;
    if (n_waypoints_out) n_waypoints_out[p] = s.n_wp;
    if (iterations_out) iterations_out[p] = s.iterations;
    if (s.ok && o.override_heading_atan2) {  // getTrajectoryReference :1582-1597
      double* smp = samples_out + (size_t)p * sample_capacity * 4;
      for (int it = 0; it + 1 < s.n_samples; ++it) {
        double* a = smp + (size_t)it * 4;
        const double* b = a + 4;
        const double dist = std::hypot(b[1] - a[1], b[0] - a[0]);
        if (dist < 0.05 && it > 0) a[3] = smp[(size_t)(it - 1) * 4 + 3];
        else a[3] = std::atan2(b[1] - a[1], b[0] - a[0]);
      }
    }
  }
  return MRS_TG_OK;
}

// getWaypointInTrajectoryIdxs :1461-1499 for one path: returns the number of indices written
inline int32_t waypoint_trajectory_idxs(const double* samples, int32_t n_samples, const mrs_tg_waypoint* waypoints,
                                        int32_t n_waypoints, int32_t* idxs_out) {
  if (!samples || !waypoints || !idxs_out) return 0;
  int widx = 0, n = 0;
  for (int i = 0; i + 1 < n_samples; ++i) {
    if (dist_from_segment(waypoints[widx].coords, samples + (size_t)i * 4, samples + (size_t)(i + 1) * 4) < 0.1) {
      idxs_out[n++] = i;
      ++widx;
    }
    if (widx == n_waypoints) break;
  }
  return n;
}

// mrs_tg_default_options: the reference's parameters where it has them
inline void default_solver_options(mrs_tg_options* opt) {
  std::memset(opt, 0, sizeof(*opt));
  opt->derivative_to_optimize = 4;
  opt->time_alloc_method = MRS_TG_TIME_ALLOC_NONE;
  opt->estimate_times = 0;
  opt->max_iterations = 10;  // config/private/trajectory_generation.yaml:10
  opt->f_rel = 0.05;         // src/mrs_trajectory_generation.cpp:884
  opt->f_abs = -1.0;
  opt->x_rel = 0.1;          // src/mrs_trajectory_generation.cpp:885
  opt->x_abs = -1.0;
  opt->sampling_dt = 0.0;
  opt->sample_capacity = 0;
  opt->flags = 0;
  opt->time_penalty = 100.0;           // config/private/trajectory_generation.yaml:4
  opt->soft_constraint_weight = 1.5;   // :6
  opt->use_soft_constraints = 1;       // :5
  opt->initial_stepsize_rel = 0.1;     // src/mrs_trajectory_generation.cpp:893
  opt->max_time_s = 0.0;               // no deadline (the nodelet sets 2 * 0.95 * timeLeft(), :899)
  opt->max_trajectory_len_factor = 3.0;   // config/public/trajectory_generation.yaml:35
  opt->min_trajectory_len_factor = 0.33;  // :36
}

// config/public/trajectory_generation.yaml + config/private/trajectory_generation.yaml; `solver` is filled by the caller
inline void default_policy_fields(mrs_tg_policy_options* o) {
  o->solver.time_alloc_method = MRS_TG_TIME_ALLOC_MELLINGER;  // config/private/trajectory_generation.yaml:7
  o->solver.derivative_to_optimize = 2;                       // :11 (0 -> acceleration)
  o->solver.sampling_dt = 0.2;                                // config/public/trajectory_generation.yaml
  o->check_deviation_enabled = 1;
  o->max_deviation = 0.05;
  o->max_deviation_iterations = 6;
  o->max_deviation_first_segment = 1;
  o->min_waypoint_distance = 0.05;
  o->path_straightener_enabled = 0;
  o->path_straightener_max_deviation = 0.05;
  o->path_straightener_max_hdg_deviation = 0.1;
  o->max_trajectory_len_factor = 3.0;
  o->min_trajectory_len_factor = 0.33;
  o->fallback_sampling = 0;
  o->fallback_speed_factor = 1.0;
  o->fallback_accel_factor = 1.0;
  o->fallback_stopping_time = 2.0;
  o->override_heading_atan2 = 0;
  o->reserved_ = 0;
  o->max_execution_time_s = 0.0;
}

}  // namespace policy
}  // namespace mrs_tg
