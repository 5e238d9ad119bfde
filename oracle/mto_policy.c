/*
 * mto_policy.c -- CPU ORACLE (test infrastructure): the path-policy layer around the solver, i.e. the
 * rows SURVEY.md section 8(f) ranks "next".  See mrs_tg_oracle.h for the rules that apply to oracle/.
 *
 * Follows (relative to /root/reference/src/mrs_trajectory_generation.cpp):
 *   :431-500    preprocessPath            (waypoint thinning, optional straightener incl. quirk B3)
 *   :620-851    optimize                  (solve, validate, insert mid-points, re-solve; <= 6 rounds)
 *   :1178-1199  trajectory length sanity check against the Baca estimate
 *   :1215-1395  findTrajectoryFallback    (constant-velocity interpolation with Baca times)
 *   :1401-1455  validateTrajectorySpatial
 *   :1461-1499  getWaypointInTrajectoryIdxs
 *   :1533-1554  distFromSegment
 *   :1560-1606  getTrajectoryReference    (override_heading_atan2)
 *   :1612-1625  interpolatePoint
 * ROS plumbing (tf, time stamps, "path from the future", MPC prediction splicing, wall-clock overtime)
 * has no counterpart here.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mrs_tg_oracle.h"

#define N MTO_N
#define DIM MTO_D

static double wrap_range(double a, double lo, double range) {
  double r = fmod(a - lo, range);
  if (r < 0) r += range;
  return r + lo;
}

/* mrs_lib radians::diff: signed shortest difference in [-pi, pi) of angles living in [0, 2 pi) */
static double radians_diff(double minuend, double subtrahend) {
  const double two_pi = 2.0 * M_PI;
  double d = wrap_range(minuend, 0.0, two_pi) - wrap_range(subtrahend, 0.0, two_pi);
  if (d < -M_PI) d += two_pi;
  else if (d >= M_PI) d -= two_pi;
  return d;
}

/* mrs_lib radians::interp: from + coeff * diff(to, from), wrapped to [0, 2 pi) */
static double radians_interp(double from, double to, double coeff) {
  return wrap_range(from + coeff * radians_diff(to, from), 0.0, 2.0 * M_PI);
}

double mto_dist_from_segment(const double* point, const double* seg1, const double* seg2) {
  /* :1533-1554 */
  double sv[3] = {seg2[0] - seg1[0], seg2[1] - seg1[1], seg2[2] - seg1[2]};
  const double len = sqrt(sv[0] * sv[0] + sv[1] * sv[1] + sv[2] * sv[2]);
  double n[3] = {sv[0], sv[1], sv[2]};
  if (len * len > 0) { /* Eigen normalize() leaves a zero vector untouched */
    n[0] /= len;
    n[1] /= len;
    n[2] /= len;
  }
  const double d1[3] = {point[0] - seg1[0], point[1] - seg1[1], point[2] - seg1[2]};
  const double coord = n[0] * d1[0] + n[1] * d1[1] + n[2] * d1[2];
  if (coord < 0) return sqrt(d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2]);
  if (coord > len) {
    const double d2[3] = {point[0] - seg2[0], point[1] - seg2[1], point[2] - seg2[2]};
    return sqrt(d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2]);
  }
  /* projection = seg1 + n n^T (point - seg1) */
  const double pr[3] = {seg1[0] + n[0] * coord, seg1[1] + n[1] * coord, seg1[2] + n[2] * coord};
  const double e[3] = {point[0] - pr[0], point[1] - pr[1], point[2] - pr[2]};
  return sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
}

void mto_interpolate_point(const double* a, const double* b, double coeff, double* out) {
  /* :1612-1625 (stop_at of the result is false) */
  for (int k = 0; k < 3; ++k) out[k] = a[k] + coeff * (b[k] - a[k]);
  out[3] = radians_interp(a[3], b[3], coeff);
}

int mto_preprocess_path(const double* wp_in, const uint8_t* stop_in, int n_in, const mto_policy_params* prm,
                        double* wp_out, uint8_t* stop_out) {
  /* :431-500 */
  int n_out = 0, last_added = 0;
  for (int i = 0; i < n_in; ++i) {
    const double* w = wp_in + (size_t)i * 4;
    if (prm->path_straightener_enabled && n_in >= 3 && i > 0 && i < n_in - 1) {
      const double* first = wp_in + (size_t)last_added * 4;
      const double* last = wp_in + (size_t)(i + 1) * 4;
      int segment_is_ok = 1;
      for (int j = last_added + 1; j < i + 1; ++j) {
        const double* mid = wp_in + (size_t)j * 4;
        const double dist = mto_dist_from_segment(mid, first, last);
        /* quirk B3: fabs() is applied to the boolean, so the heading test is a signed comparison */
        if (dist > prm->path_straightener_max_deviation ||
            fabs((double)(radians_diff(first[3], mid[3]) > prm->path_straightener_max_hdg_deviation)) ||
            fabs((double)(radians_diff(last[3], mid[3]) > prm->path_straightener_max_hdg_deviation))) {
          segment_is_ok = 0;
          break;
        }
      }
      if (segment_is_ok) continue;
    }
    if (i > 0 && i < n_in - 1) {
      const double* first = wp_in + (size_t)last_added * 4;
      const double dx = first[0] - w[0], dy = first[1] - w[1], dz = first[2] - w[2];
      if (sqrt(dx * dx + dy * dy + dz * dz) < prm->min_waypoint_distance) continue;
    }
    memcpy(wp_out + (size_t)n_out * 4, w, sizeof(double) * 4);
    stop_out[n_out] = stop_in ? stop_in[i] : 0;
    ++n_out;
    last_added = i;
  }
  return n_out;
}

int mto_validate_trajectory_spatial(const double* samples, int n_samples, const double* wps, int n_wp,
                                    const mto_policy_params* prm, uint8_t* segment_safe, double* max_deviation_out) {
  /* :1401-1455 */
  for (int i = 0; i < n_wp - 1; ++i) segment_safe[i] = 1;
  int waypoint_idx = 0, is_safe = 1;
  double max_deviation = 0;
  for (int i = 0; i + 1 < n_samples; ++i) {
    const double* sample = samples + (size_t)i * 4;
    const double* next = samples + (size_t)(i + 1) * 4;
    const double* s0 = wps + (size_t)waypoint_idx * 4;
    const double* s1 = wps + (size_t)(waypoint_idx + 1) * 4;
    const double d_seg = mto_dist_from_segment(sample, s0, s1);
    const double d_end = mto_dist_from_segment(s1, sample, next);
    if (waypoint_idx > 0 || prm->max_deviation_first_segment || n_wp <= 2) {
      if (d_seg > max_deviation) max_deviation = d_seg;
      if (d_seg > prm->max_deviation) {
        segment_safe[waypoint_idx] = 0;
        is_safe = 0;
      }
    }
    if (d_end < 0.05 && waypoint_idx < n_wp - 2) ++waypoint_idx;
  }
  if (max_deviation_out) *max_deviation_out = max_deviation;
  return is_safe;
}

int mto_waypoint_trajectory_idxs(const double* samples, int n_samples, const double* wps, int n_wp, int32_t* idxs) {
  /* :1461-1499 */
  int waypoint_idx = 0, n = 0;
  for (int i = 0; i + 1 < n_samples; ++i) {
    const double d = mto_dist_from_segment(wps + (size_t)waypoint_idx * 4, samples + (size_t)i * 4, samples + (size_t)(i + 1) * 4);
    if (d < 0.1) {
      idxs[n++] = i;
      ++waypoint_idx;
    }
    if (waypoint_idx == n_wp) break;
  }
  return n;
}

int mto_fallback_sampling(const double* wps_in, const uint8_t* stop_at, int n_wp, const double* limits9, int relax_heading,
                          const mto_policy_params* prm, double dt, double* out, int capacity) {
  /* :1215-1395: headings unwrapped sequentially, Baca times with scaled speed / acceleration limits,
   * linear interpolation, dwell at stop_at waypoints */
  double* wps = (double*)malloc(sizeof(double) * 4 * (size_t)n_wp);
  double* t_baca = (double*)malloc(sizeof(double) * (size_t)(n_wp - 1));
  double last = wps_in[3];
  for (int i = 0; i < n_wp; ++i) {
    memcpy(wps + (size_t)i * 4, wps_in + (size_t)i * 4, sizeof(double) * 4);
    wps[(size_t)i * 4 + 3] = mto_unwrap_heading(wps_in[(size_t)i * 4 + 3], last);
    last = wps[(size_t)i * 4 + 3];
  }
  double lim[9];
  memcpy(lim, limits9, sizeof(lim));
  lim[0] *= prm->fallback_speed_factor;
  lim[1] *= prm->fallback_speed_factor;
  lim[3] *= prm->fallback_accel_factor;
  lim[4] *= prm->fallback_accel_factor;
  if (relax_heading) lim[2] = lim[5] = lim[8] = (double)FLT_MAX; /* :1285-1288 */
  mto_estimate_segment_times_baca(n_wp - 1, wps, lim, t_baca);
  int count = 0;
  for (int i = 0; i < n_wp - 1; ++i) {
    const double segment_time = t_baca[i];
    int n_samples = 0;
    double step = 0;
    if (segment_time > 1e-1) {
      n_samples = (int)ceil(segment_time / dt);
      step = (n_samples > 0) ? 1.0 / (double)n_samples : 0.5;
    }
    if (n_samples > 0 && i == n_wp - 2) ++n_samples;
    for (int j = 0; j < n_samples; ++j) {
      double p[4];
      /* interpolation runs on the ORIGINAL (not unwrapped) waypoints, :1362 */
      mto_interpolate_point(wps_in + (size_t)i * 4, wps_in + (size_t)(i + 1) * 4, j * step, p);
      p[3] = mto_wrap_yaw(p[3]);
      int repeat = 1;
      if (j == 0 && i > 0 && stop_at && stop_at[i]) repeat += (int)round(prm->fallback_stopping_time / dt);
      for (int r = 0; r < repeat; ++r) {
        if (count < capacity && out) memcpy(out + (size_t)count * 4, p, sizeof(p));
        ++count;
      }
    }
  }
  free(wps);
  free(t_baca);
  return count;
}

void mto_default_policy_params(mto_policy_params* p) {
  /* config/public/trajectory_generation.yaml */
  p->check_deviation_enabled = 1;
  p->max_deviation = 0.05;
  p->max_deviation_iterations = 6;
  p->max_deviation_first_segment = 1;
  p->min_waypoint_distance = 0.05;
  p->path_straightener_enabled = 0;
  p->path_straightener_max_deviation = 0.05;
  p->path_straightener_max_hdg_deviation = 0.1;
  p->max_trajectory_len_factor = 3.0;
  p->min_trajectory_len_factor = 0.33;
  p->fallback_sampling = 0;
  p->fallback_speed_factor = 1.0;
  p->fallback_accel_factor = 1.0;
  p->fallback_stopping_time = 2.0;
  p->override_heading_atan2 = 0;
}

/* findTrajectory for one waypoint list (src/mrs_trajectory_generation.cpp:857-1209): vertices (:923-977), limits (:985-1038),
 * Euclidean + Baca estimates (:1046-1056), optimise (:1083), the gate on the optimiser's code (:1138-1149), sampling (:1169),
 * the length check (:1178-1199).  Returns 1 where the reference returns the states, 0 where it returns {}.
 * seg_times_out [n_wp - 1], status_out, baca_total_out, raw_n_samples_out (the sample count before the gates) may be NULL;
 * *rejection_out: 0 accepted, 1 code, 2 too long, 3 too short, 4 more samples than `capacity`. */
int mto_find_trajectory(const double* wps_raw, const uint8_t* stop_at, int n_wp, const double* init /*13 or NULL*/,
                        const double* limits9, int relax_heading, const mto_options* sopt, const mto_policy_params* prm,
                        double* samples, int capacity, int* n_samples_out, double* seg_times_out, int32_t* status_out,
                        double* baca_total_out, int* raw_n_samples_out, int* rejection_out) {
  const int S = n_wp - 1, d = sopt->derivative_to_optimize;
  double* wp = (double*)malloc(sizeof(double) * 4 * (size_t)n_wp);
  uint8_t* mask = (uint8_t*)calloc((size_t)n_wp * 5, 1);
  double* vals = (double*)calloc((size_t)n_wp * 20, sizeof(double));
  double* times = (double*)calloc((size_t)S, sizeof(double));
  double* coeffs = (double*)calloc((size_t)S * 40, sizeof(double));
  double* t_baca = (double*)malloc(sizeof(double) * (size_t)S);
  double last_heading = init ? init[0] : wps_raw[3];
  for (int i = 0; i < n_wp; ++i) {
    memcpy(wp + (size_t)i * 4, wps_raw + (size_t)i * 4, sizeof(double) * 4);
    wp[(size_t)i * 4 + 3] = mto_unwrap_heading(wps_raw[(size_t)i * 4 + 3], last_heading);
    last_heading = wp[(size_t)i * 4 + 3];
    mask[i * 5] = 1;
    memcpy(vals + (size_t)(i * 5) * 4, wp + (size_t)i * 4, sizeof(double) * 4);
    if (i == 0 || i == n_wp - 1) {
      for (int k = 1; k <= d; ++k) mask[i * 5 + k] = 1;
      if (i == 0 && init) {
        for (int k = 1; k <= 3; ++k) {
          mask[k] = 1;
          memcpy(vals + (size_t)k * 4, init + 1 + (k - 1) * 4, sizeof(double) * 4);
        }
      }
    } else if (stop_at && stop_at[i]) {
      mask[i * 5 + 1] = mask[i * 5 + 2] = mask[i * 5 + 3] = 1;
    }
  }
  double lim[9];
  memcpy(lim, limits9, sizeof(lim));
  if (relax_heading) lim[2] = lim[5] = lim[8] = (double)FLT_MAX;
  mto_estimate_segment_times_baca(S, wp, lim, t_baca);
  double total_baca = 0;
  for (int i = 0; i < S; ++i) total_baca += t_baca[i];
  const int32_t so[2] = {0, S};
  int32_t status = 0, ns = 0;
  double cost = 0;
  mto_options o = *sopt;
  o.estimate_times = 1;
  mto_solve_batch(1, so, wp, mask, vals, lim, &o, times, coeffs, &status, &cost, &ns, samples, capacity, 1);
  int rejection = 0;
  int ok = (status >= 1 && status != 6) || status == -1; /* :1138-1149 */
  if (!ok) rejection = 1;
  const double len = (double)ns * sopt->sampling_dt;     /* :1178-1199 */
  if (ok && len > 1.0 && len > prm->max_trajectory_len_factor * total_baca) {
    ok = 0;
    rejection = 2;
  } else if (ok && len > 1.0 && len < prm->min_trajectory_len_factor * total_baca) {
    ok = 0;
    rejection = 3;
  }
  if (ok && ns > capacity) {
    ok = 0;
    rejection = 4;
  }
  *n_samples_out = ok ? ns : 0;
  if (seg_times_out) memcpy(seg_times_out, times, sizeof(double) * (size_t)S);
  if (status_out) *status_out = status;
  if (baca_total_out) *baca_total_out = total_baca;
  if (raw_n_samples_out) *raw_n_samples_out = ns;
  if (rejection_out) *rejection_out = rejection;
  free(wp);
  free(mask);
  free(vals);
  free(times);
  free(coeffs);
  free(t_baca);
  return ok;
}

static int find_trajectory(const double* wps_raw, const uint8_t* stop_at, int n_wp, const double* init /*13 or NULL*/,
                           const double* limits9, int relax_heading, const mto_options* sopt, const mto_policy_params* prm,
                           double* samples, int capacity, int* n_samples_out) {
  return mto_find_trajectory(wps_raw, stop_at, n_wp, init, limits9, relax_heading, sopt, prm, samples, capacity, n_samples_out,
                             NULL, NULL, NULL, NULL, NULL);
}

int mto_optimize_path(const double* wps_in, const uint8_t* stop_in, int n_in, const double* initial_state,
                      const double* limits9, int relax_heading, const mto_options* sopt, const mto_policy_params* prm,
                      double* samples_out, int capacity, int* n_samples_out, double* max_deviation_out,
                      int* n_waypoints_out, int* iterations_out) {
  /* optimize() :620-851 without the ROS-only branches */
  const int cap_wp = n_in << (prm->max_deviation_iterations > 0 ? prm->max_deviation_iterations : 0);
  double* wps = (double*)malloc(sizeof(double) * 4 * (size_t)(cap_wp + 2));
  uint8_t* stop = (uint8_t*)malloc((size_t)(cap_wp + 2));
  uint8_t* safe = (uint8_t*)malloc((size_t)(cap_wp + 2));
  *n_samples_out = 0;
  if (max_deviation_out) *max_deviation_out = 0;
  if (iterations_out) *iterations_out = 0;
  int n_wp = mto_preprocess_path(wps_in, stop_in, n_in, prm, wps, stop);
  if (n_waypoints_out) *n_waypoints_out = n_wp;
  int ok = 0, ns = 0;
  double max_dev = 0;
  if (n_wp <= 1) goto done; /* "the path is empty (after postprocessing)" */
  if (prm->fallback_sampling) {
    ns = mto_fallback_sampling(wps, stop, n_wp, limits9, relax_heading, prm, sopt->sampling_dt, samples_out, capacity);
    ok = ns <= capacity;
  } else {
    ok = find_trajectory(wps, stop, n_wp, initial_state, limits9, relax_heading, sopt, prm, samples_out, capacity, &ns);
  }
  if (!ok) goto done;
  for (int k = 0; k < prm->max_deviation_iterations; ++k) {
    const int is_safe = mto_validate_trajectory_spatial(samples_out, ns, wps, n_wp, prm, safe, &max_dev);
    if (prm->check_deviation_enabled && !is_safe) {
      /* insert a mid-point into every unsafe segment (:739-753) */
      int w = 0, sidx = 0;
      while (w < n_wp - 1) {
        if (!safe[sidx] && (w > 0 || prm->max_deviation_first_segment || n_wp <= 2)) {
          double mid[4];
          mto_interpolate_point(wps + (size_t)w * 4, wps + (size_t)(w + 1) * 4, 0.5, mid);
          memmove(wps + (size_t)(w + 2) * 4, wps + (size_t)(w + 1) * 4, sizeof(double) * 4 * (size_t)(n_wp - w - 1));
          memmove(stop + w + 2, stop + w + 1, (size_t)(n_wp - w - 1));
          memcpy(wps + (size_t)(w + 1) * 4, mid, sizeof(mid));
          stop[w + 1] = 0;
          ++n_wp;
          ++w; /* the iterator now points at the inserted mid-point */
        }
        ++sidx;
        ++w;
      }
      if (iterations_out) *iterations_out = k + 1;
      if (prm->fallback_sampling) {
        ns = mto_fallback_sampling(wps, stop, n_wp, limits9, relax_heading, prm, sopt->sampling_dt, samples_out, capacity);
        ok = ns <= capacity;
      } else {
        ok = find_trajectory(wps, stop, n_wp, initial_state, limits9, relax_heading, sopt, prm, samples_out, capacity, &ns);
      }
      if (!ok) goto done;
    } else {
      break;
    }
  }
  if (prm->override_heading_atan2) { /* getTrajectoryReference :1582-1597 */
    for (int it = 0; it < ns; ++it) {
      double* p = samples_out + (size_t)it * 4;
      if (it < ns - 1) {
        const double* q = samples_out + (size_t)(it + 1) * 4;
        const double dist = hypot(q[1] - p[1], q[0] - p[0]);
        if (dist < 0.05 && it > 0) p[3] = samples_out[(size_t)(it - 1) * 4 + 3];
        else p[3] = atan2(q[1] - p[1], q[0] - p[0]);
      }
    }
  }
done:
  /* bookkeeping of the state the loop ended in, for a failed request as well (the product reports the same) */
  if (max_deviation_out) *max_deviation_out = max_dev;
  if (n_waypoints_out) *n_waypoints_out = n_wp;
  *n_samples_out = ok ? ns : 0;
  free(wps);
  free(stop);
  free(safe);
  return ok;
}
