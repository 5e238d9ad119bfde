/*
 * mto_nonlinear.c -- CPU ORACLE (test infrastructure): segment-time outer loop, feasibility
 * scaling, sampling, segment-time estimators and the batch driver.  See mrs_tg_oracle.h for the
 * rules that apply to everything under oracle/.
 *
 * Follows (relative to /root/reference/):
 *   include/eth_trajectory_generation/impl/polynomial_optimization_nonlinear_impl.h:160-234,257-333,336-408,617-649
 *   src/eth_trajectory_generation/trajectory.cpp:93-151,211-243,422-565,598-692
 *   src/eth_trajectory_generation/segment.cpp:113-212
 *   src/eth_trajectory_generation/polynomial.cpp:36-85,218-224
 *   src/eth_trajectory_generation/trajectory_sampling.cpp:49-124
 *   src/eth_trajectory_generation/vertex.cpp:301-565
 *   include/eth_mav_msgs/common.h:130-140
 *   src/mrs_trajectory_generation.cpp:931-977,1046-1083,1163-1169
 *
 * NLopt (>= 2.4.2, un-vendored; LD_LBFGS selected at src/mrs_trajectory_generation.cpp:891) is not
 * available, and its Luksan PLIS internals are not restated.  The outer loop below is this
 * project's own deterministic projected L-BFGS with NLopt's documented stopping rules
 * (maxeval, ftol_rel/abs, xtol_rel/abs) and return codes; the HIP path implements the same
 * specification ("outer loop").
 */
#include "mrs_tg_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define N MTO_N
#define DIM MTO_D
#define HALF MTO_HALF

static const double kTimeLowerBound = 0.01; /* kOptimizationTimeLowerBound, nonlinear.h:304 */

/* ------------------------------------------------------------------------------------------- */
/* Mellinger cost + forward-difference gradient                                                 */

double mto_cost_and_gradient_mellinger(const mto_path* path, const double* seg_times, double* grad) {
  const int S = path->n_seg, d = path->derivative_to_optimize;
  const mto_scratch_state mark = mto_scratch_mark();
  double* coeffs = (double*)mto_scratch_alloc(sizeof(double) * (size_t)S * DIM * N, 0);
  double bigger[MTO_MAX_SEG];
  /* objectiveFunctionTimeMellingerOuterLoop: updateSegmentTimes + solveLinear (nonlinear_impl.h:626-627) */
  mto_solve_linear(path, seg_times, coeffs);
  const double J_d = mto_compute_cost(S, d, seg_times, coeffs);
  if (S == 1) { /* nonlinear_impl.h:264-271 */
    if (grad) grad[0] = 0.0;
    mto_scratch_release(mark);
    return J_d;
  }
  if (grad) {
    const double increment_time = 0.1; /* nonlinear_impl.h:281 */
    for (int n = 0; n < S; ++n) {
      const double corr = increment_time / ((double)S - 1.0);
      for (int i = 0; i < S; ++i) {
        bigger[i] = seg_times[i];
        if (i == n) bigger[i] += increment_time;
        else bigger[i] -= corr;
      }
      for (int i = 0; i < S; ++i) bigger[i] = (bigger[i] > kTimeLowerBound) ? bigger[i] : kTimeLowerBound;
      mto_solve_linear(path, bigger, coeffs);
      const double J_bigger = mto_compute_cost(S, d, bigger, coeffs);
      grad[n] = (J_bigger - J_d) / increment_time;
    }
    /* the reference restores and re-solves here (nonlinear_impl.h:327-328); the result equals the
     * first solve, nothing observable depends on it, so the oracle does not repeat it */
  }
  mto_scratch_release(mark);
  return J_d;
}

/* ------------------------------------------------------------------------------------------- */
/* outer loop: projected L-BFGS, memory 5, Armijo backtracking                                   */

static int relstop(double vold, double vnew, double reltol, double abstol) {
  /* NLopt's stopping rule for a scalar: |dv| < abstol, or |dv| < reltol*(|vnew|+|vold|)/2,
   * or (reltol > 0 and vnew == vold) */
  if (isinf(vold)) return 0;
  const double dv = fabs(vnew - vold);
  return dv < abstol || dv < reltol * (fabs(vnew) + fabs(vold)) * 0.5 || (reltol > 0 && vnew == vold);
}

#define LBFGS_M 5

/* Optional per-thread trace of the search's decisions (scripts/divergence_histogram.py): one record of MTO_TRACE_REC doubles
 * per objective evaluation after the first --
 *   [0] evaluation number  [1] f of the current iterate  [2] fn of the trial  [3] slope g^T (xn - x)  [4] alpha
 *   [5] Armijo margin (f + 1e-4 slope - fn) / |f|: >= 0 accepted
 *   [6] ftol margin (f_rel (|fn| + |f|) / 2 - |fn - f|) / |f|: > 0 stops (accepted trials only, else 0)
 *   [7] xtol margin min_i (x_rel (|xn_i| + |x_i|) / 2 - |xn_i - x_i|) / |x_i|: > 0 stops (accepted trials only, else 0) */
static __thread double* t_trace = NULL;
static __thread int t_trace_cap = 0, t_trace_n = 0;
void mto_set_optimizer_trace(double* buf, int cap_records) {
  t_trace = buf;
  t_trace_cap = cap_records;
  t_trace_n = 0;
}
int mto_optimizer_trace_count(void) { return t_trace_n; }

int mto_optimize_times_mellinger(const mto_path* path, const mto_nlopt_params* prm, double* x, int* n_eval_out,
                                 double* final_cost_out) {
  const int S = path->n_seg;
  double g[MTO_MAX_SEG], xn[MTO_MAX_SEG], gn[MTO_MAX_SEG], dir[MTO_MAX_SEG];
  double sm[LBFGS_M][MTO_MAX_SEG], ym[LBFGS_M][MTO_MAX_SEG], rho[LBFGS_M], al[LBFGS_M];
  int npairs = 0, neval = 0, ret = MTO_FAILURE;
  for (int i = 0; i < S; ++i)
    if (x[i] < kTimeLowerBound) { /* NLopt rejects a start outside the bounds */
      if (n_eval_out) *n_eval_out = 0;
      return MTO_INVALID_ARGS;
    }
  double f = mto_cost_and_gradient_mellinger(path, x, g);
  neval = 1;
  double f_last = f;
  if (prm->max_iterations > 0 && neval >= prm->max_iterations) {
    ret = MTO_MAXEVAL_REACHED;
    goto done;
  }
  for (int iter = 0;; ++iter) {
    /* two-loop recursion (newest pair last) */
    for (int i = 0; i < S; ++i) dir[i] = -g[i];
    if (npairs > 0) {
      for (int k = npairs - 1; k >= 0; --k) {
        double sd = 0.0;
        for (int i = 0; i < S; ++i) sd += sm[k][i] * dir[i];
        al[k] = rho[k] * sd;
        for (int i = 0; i < S; ++i) dir[i] -= al[k] * ym[k][i];
      }
      double sy = 0.0, yy = 0.0;
      for (int i = 0; i < S; ++i) {
        sy += sm[npairs - 1][i] * ym[npairs - 1][i];
        yy += ym[npairs - 1][i] * ym[npairs - 1][i];
      }
      const double gamma = sy / yy;
      for (int i = 0; i < S; ++i) dir[i] *= gamma;
      for (int k = 0; k < npairs; ++k) {
        double yd = 0.0;
        for (int i = 0; i < S; ++i) yd += ym[k][i] * dir[i];
        const double beta = rho[k] * yd;
        for (int i = 0; i < S; ++i) dir[i] += (al[k] - beta) * sm[k][i];
      }
    }
    /* project: a variable sitting on its lower bound may not move further down */
    double gd = 0.0;
    for (int i = 0; i < S; ++i) {
      if (x[i] <= kTimeLowerBound && dir[i] < 0.0) dir[i] = 0.0;
      gd += g[i] * dir[i];
    }
    if (!(gd < 0.0)) { /* not a descent direction: fall back to projected steepest descent */
      gd = 0.0;
      for (int i = 0; i < S; ++i) {
        dir[i] = -g[i];
        if (x[i] <= kTimeLowerBound && dir[i] < 0.0) dir[i] = 0.0;
        gd += g[i] * dir[i];
      }
      npairs = 0;
      if (!(gd < 0.0)) { /* projected gradient vanishes */
        ret = MTO_SUCCESS;
        goto done;
      }
    }
    double alpha = 1.0;
    if (iter == 0 || npairs == 0) { /* first trial step moves x by at most 10 % in norm */
      double nx = 0.0, nd = 0.0;
      for (int i = 0; i < S; ++i) {
        nx += x[i] * x[i];
        nd += dir[i] * dir[i];
      }
      const double cap = 0.1 * sqrt(nx) / sqrt(nd);
      if (cap < alpha) alpha = cap;
    }
    double fn = f;
    for (;;) { /* Armijo backtracking; every trial is one objective evaluation */
      double slope = 0.0;
      for (int i = 0; i < S; ++i) {
        const double t = x[i] + alpha * dir[i];
        xn[i] = (t > kTimeLowerBound) ? t : kTimeLowerBound;
        slope += g[i] * (xn[i] - x[i]);
      }
      fn = mto_cost_and_gradient_mellinger(path, xn, gn);
      ++neval;
      f_last = fn;
      if (t_trace && t_trace_n < t_trace_cap) {
        double* r = t_trace + (size_t)t_trace_n * MTO_TRACE_REC;
        const double fa = fabs(f) > 1e-300 ? fabs(f) : 1e-300;
        r[0] = neval;
        r[1] = f;
        r[2] = fn;
        r[3] = slope;
        r[4] = alpha;
        r[5] = (f + 1e-4 * slope - fn) / fa;
        r[6] = 0.0;
        r[7] = 0.0;
        if (fn <= f + 1e-4 * slope) {
          r[6] = (prm->f_rel * (fabs(fn) + fabs(f)) * 0.5 - fabs(fn - f)) / fa;
          double mn = DBL_MAX;
          for (int i = 0; i < S; ++i) {
            const double m = (prm->x_rel * (fabs(xn[i]) + fabs(x[i])) * 0.5 - fabs(xn[i] - x[i])) / fabs(x[i]);
            if (m < mn) mn = m;
          }
          r[7] = mn;
        }
        ++t_trace_n;
      }
      if (fn <= f + 1e-4 * slope) break;
      if (prm->max_iterations > 0 && neval >= prm->max_iterations) { /* budget ends on a rejected trial */
        memcpy(x, xn, sizeof(double) * (size_t)S);
        ret = MTO_MAXEVAL_REACHED;
        goto done;
      }
      alpha *= 0.5;
      if (alpha < 1e-12) {
        memcpy(x, xn, sizeof(double) * (size_t)S);
        ret = MTO_XTOL_REACHED;
        goto done;
      }
    }
    /* accepted step */
    int stop = 0;
    if (relstop(f, fn, prm->f_rel, prm->f_abs)) stop = MTO_FTOL_REACHED;
    else {
      int allx = 1;
      for (int i = 0; i < S; ++i)
        if (!relstop(x[i], xn[i], prm->x_rel, prm->x_abs)) {
          allx = 0;
          break;
        }
      if (allx) stop = MTO_XTOL_REACHED;
    }
    double sy = 0.0, ss = 0.0, yy = 0.0;
    double snew[MTO_MAX_SEG], ynew[MTO_MAX_SEG];
    for (int i = 0; i < S; ++i) {
      snew[i] = xn[i] - x[i];
      ynew[i] = gn[i] - g[i];
      sy += snew[i] * ynew[i];
      ss += snew[i] * snew[i];
      yy += ynew[i] * ynew[i];
    }
    memcpy(x, xn, sizeof(double) * (size_t)S);
    memcpy(g, gn, sizeof(double) * (size_t)S);
    f = fn;
    if (stop) {
      ret = stop;
      goto done;
    }
    if (prm->max_iterations > 0 && neval >= prm->max_iterations) {
      ret = MTO_MAXEVAL_REACHED;
      goto done;
    }
    if (sy > 0.0 && sy * sy > 1e-20 * (ss * yy)) { /* curvature condition s^T y > 1e-10 |s| |y| (in squares): remember the pair */
      if (npairs == LBFGS_M) {
        for (int k = 1; k < LBFGS_M; ++k) {
          memcpy(sm[k - 1], sm[k], sizeof(double) * (size_t)S);
          memcpy(ym[k - 1], ym[k], sizeof(double) * (size_t)S);
          rho[k - 1] = rho[k];
        }
        npairs = LBFGS_M - 1;
      }
      memcpy(sm[npairs], snew, sizeof(double) * (size_t)S);
      memcpy(ym[npairs], ynew, sizeof(double) * (size_t)S);
      rho[npairs] = 1.0 / sy;
      ++npairs;
    }
  }
done:
  if (n_eval_out) *n_eval_out = neval;
  if (final_cost_out) *final_cost_out = f_last;
  return ret;
}

/* ------------------------------------------------------------------------------------------- */
/* feasibility                                                                                  */

double mto_segment_max_magnitude(const double* seg_coeffs, double T, int derivative, const int* dims, int n_dims) {
  double cand[2 * N + 4];
  int n_cand = 0;
  double re[2 * N], im[2 * N];
  int n_roots = 0;
  if (n_dims > 1) {
    /* roots of sum_dim p^(k) * p^(k+1)   segment.cpp:122-147 */
    const int n_d = N - derivative, n_dd = n_d - 1, len = n_d + n_dd - 1;
    double conv[2 * N], acc[2 * N], dcoef[N], ddcoef[N];
    for (int i = 0; i < len; ++i) acc[i] = 0.0;
    for (int q = 0; q < n_dims; ++q) {
      mto_poly_derivative(seg_coeffs + dims[q] * N, N, derivative, dcoef);
      mto_poly_derivative(seg_coeffs + dims[q] * N, N, derivative + 1, ddcoef);
      mto_convolve(dcoef, n_d, ddcoef, n_dd, conv);
      for (int i = 0; i < len; ++i) acc[i] += conv[i];
    }
    /* Polynomial(convolved).computeMinMaxCandidates(t0, t1, -1): roots of getCoefficients(0) */
    n_roots = mto_find_roots_jenkins_traub(acc, len, re, im);
  } else {
    /* roots of p^(k+1), zero-padded to N coefficients   polynomial.cpp:69-85 */
    double dd[N];
    mto_poly_derivative(seg_coeffs + dims[0] * N, N, derivative + 1, dd);
    n_roots = mto_find_roots_jenkins_traub(dd, N, re, im);
  }
  /* selectMinMaxCandidatesFromRoots polynomial.cpp:36-63 */
  cand[n_cand++] = 0.0;
  cand[n_cand++] = T;
  for (int i = 0; i < n_roots; ++i) {
    if (fabs(im[i]) > DBL_EPSILON) continue;
    if (re[i] < 0.0 || re[i] > T) continue;
    cand[n_cand++] = re[i];
  }
  /* magnitudes segment.cpp:170-179, maximum segment.cpp:199-209 */
  double best = -DBL_MAX;
  for (int i = 0; i < n_cand; ++i) {
    double m = 0.0;
    for (int q = 0; q < n_dims; ++q) m += pow(mto_poly_eval(seg_coeffs + dims[q] * N, N, cand[i], derivative), 2);
    m = sqrt(m);
    if (m > best) best = m;
  }
  return best;
}

static void scale_polynomial_in_time(double* c, double scaling_factor) {
  /* polynomial.cpp:218-224 */
  double scale = 1.0;
  for (int n = 0; n < N; ++n) {
    c[n] *= scale;
    scale *= scaling_factor;
  }
}

/* violations[3] = {velocity, acceleration, jerk}: max over groups of maximum/limit for one segment */
static void segment_violations(const double* seg_coeffs, double T, const double* limits, double* viol) {
  static const int dims_h[2] = {0, 1}, dims_v[1] = {2}, dims_hdg[1] = {3};
  for (int k = 1; k <= 3; ++k) {
    const double h = mto_segment_max_magnitude(seg_coeffs, T, k, dims_h, 2) / limits[(k - 1) * 3 + 0];
    const double v = mto_segment_max_magnitude(seg_coeffs, T, k, dims_v, 1) / limits[(k - 1) * 3 + 1];
    const double y = mto_segment_max_magnitude(seg_coeffs, T, k, dims_hdg, 1) / limits[(k - 1) * 3 + 2];
    double m = (h > v) ? h : v;
    m = (m > y) ? m : y;
    viol[k - 1] = m;
  }
}

int mto_scale_segment_times_to_meet_constraints(int n_seg, double* coeffs, double* seg_times, const double* limits,
                                                int* n_sweeps_out) {
  const int kMaxCounter = 20;
  const double kTolerance = 1e-3;
  int within_range = 0, sweeps = 0;
  for (int it = 0; it < kMaxCounter; ++it) {
    ++sweeps;
    for (int s = 0; s < n_seg; ++s) {
      double viol[3];
      double* sc = coeffs + (size_t)s * DIM * N;
      segment_violations(sc, seg_times[s], limits, viol);
      double scaling = viol[0];
      const double sa = sqrt(viol[1]), sj = cbrt(viol[2]);
      if (sa > scaling) scaling = sa;
      if (sj > scaling) scaling = sj;
      if (scaling < 1.0) scaling = 1.0;
      const double inv = 1.0 / scaling;
      for (int k = 0; k < DIM; ++k) scale_polynomial_in_time(sc + k * N, inv);
      seg_times[s] = seg_times[s] * scaling;
    }
    /* whole-trajectory check trajectory.cpp:660-689: maximum over segments per group */
    double vmax[3] = {0.0, 0.0, 0.0};
    {
      static const int dims_h[2] = {0, 1}, dims_v[1] = {2}, dims_hdg[1] = {3};
      for (int k = 1; k <= 3; ++k) {
        double mh = -DBL_MAX, mv = -DBL_MAX, my = -DBL_MAX;
        for (int s = 0; s < n_seg; ++s) {
          const double* sc = coeffs + (size_t)s * DIM * N;
          const double a = mto_segment_max_magnitude(sc, seg_times[s], k, dims_h, 2);
          const double b = mto_segment_max_magnitude(sc, seg_times[s], k, dims_v, 1);
          const double c = mto_segment_max_magnitude(sc, seg_times[s], k, dims_hdg, 1);
          if (a > mh) mh = a;
          if (b > mv) mv = b;
          if (c > my) my = c;
        }
        const double rh = mh / limits[(k - 1) * 3 + 0], rv = mv / limits[(k - 1) * 3 + 1], ry = my / limits[(k - 1) * 3 + 2];
        double m = (rh > rv) ? rh : rv;
        vmax[k - 1] = (m > ry) ? m : ry;
      }
    }
    within_range = vmax[0] <= 1.0 + kTolerance && vmax[1] <= 1.0 + kTolerance && vmax[2] <= 1.0 + kTolerance;
    if (within_range) break;
  }
  if (n_sweeps_out) *n_sweeps_out = sweeps;
  return within_range;
}

/* ------------------------------------------------------------------------------------------- */
/* sampling                                                                                     */

double mto_wrap_yaw(double yaw) {
  /* quaternionFromYaw: AngleAxis(yaw, z) -> (w, 0, 0, z) = (cos(yaw/2), 0, 0, sin(yaw/2));
   * yawFromQuaternion: atan2(2 (w z + x y), 1 - 2 (y^2 + z^2))      common.h:130-140 */
  const double w = cos(yaw * 0.5), z = sin(yaw * 0.5);
  return atan2(2.0 * (w * z), 1.0 - 2.0 * (z * z));
}

int mto_sample_trajectory(int n_seg, const double* coeffs, const double* seg_times, double dt, int derivative,
                          double* out, int capacity) {
  /* sampleWholeTrajectory: t_start = 0, t_end = sum of segment times (trajectory.h max_time_) */
  double t_end = 0.0;
  for (int i = 0; i < n_seg; ++i) t_end += seg_times[i];
  const double t_start = 0.0;
  double accumulated = 0.0;
  int i = 0;
  for (i = 0; i < n_seg; ++i) { /* trajectory.cpp:108-120 */
    accumulated += seg_times[i];
    if (accumulated > t_start) break;
  }
  if (t_start > accumulated) return 0;
  if (i >= n_seg) return 0;
  accumulated -= seg_times[i];
  double time_in_segment = t_start - accumulated;
  int count = 0;
  while (accumulated < t_end) { /* trajectory.cpp:131-150 */
    if (time_in_segment > seg_times[i]) {
      time_in_segment = time_in_segment - seg_times[i];
      ++i;
      if (i >= n_seg) break;
      continue;
    }
    if (count < capacity && out) {
      for (int k = 0; k < DIM; ++k)
        out[(size_t)count * DIM + k] = mto_poly_eval(coeffs + ((size_t)i * DIM + k) * N, N, time_in_segment, derivative);
    }
    ++count;
    time_in_segment += dt;
    accumulated += dt;
    /* A trajectory whose outer loop ended on a rejected trial point can have segment times of 1e12 s after the
     * feasibility scaling (the nodelet discards such results by its length check, src/...cpp:1178-1199); the
     * reference would push 5e12 samples here.  Callers only distinguish "more than capacity", so stop there. */
    if (capacity > 0 && count > capacity) break;
  }
  return count;
}

/* ------------------------------------------------------------------------------------------- */
/* input side                                                                                   */

static double wrap_pi(double a) {
  /* mrs_lib cyclic<double, sradians>::wrap to [-pi, pi) (mrs_lib/geometry/cyclic.h, not vendored:
   * restated from its documented meaning) */
  const double two_pi = 2.0 * M_PI;
  double r = fmod(a + M_PI, two_pi);
  if (r < 0) r += two_pi;
  return r - M_PI;
}

static double angle_diff(double minuend, double subtrahend) {
  /* cyclic::diff: signed shortest difference in [-pi, pi) */
  const double two_pi = 2.0 * M_PI;
  double d = wrap_pi(minuend) - wrap_pi(subtrahend);
  if (d < -M_PI) d += two_pi;
  else if (d >= M_PI) d -= two_pi;
  return d;
}

double mto_unwrap_heading(double what, double from) { return from + angle_diff(what, from); }

static double vmax_for_inclination(double inclinator, double lim_vertical, double lim_horizontal) {
  /* vertex.cpp:516-520 (and :337-353 for a_max, j_max) */
  if (inclinator > atan2(lim_vertical, lim_horizontal) || inclinator < -atan2(lim_vertical, lim_horizontal))
    return fabs(lim_vertical / sin(inclinator));
  return fabs(lim_horizontal / cos(inclinator));
}

static double heading_fix_time(double start_hdg, double end_hdg, double w_max, double a_max, double factor) {
  /* vertex.cpp:536-555 (Euclidean, factor 1) and :457-476 (Baca, factor 2) */
  const double angular_distance = fabs(angle_diff(start_hdg, end_hdg));
  double t_vel = 0.0, t_acc = 0.0;
  if (w_max < FLT_MAX && a_max < FLT_MAX) {
    const double reduced = (angular_distance - factor * (w_max * w_max) / a_max) / w_max;
    t_vel = (reduced < 0) ? angular_distance / w_max : reduced;
    if (angular_distance > M_PI / 4) t_acc = 2 * (w_max / a_max);
  }
  return 1.5 * (t_vel + t_acc);
}

void mto_estimate_segment_times_euclidean(int n_seg, const double* wp, const double* lim, double* out) {
  const double v_h = lim[0], v_v = lim[1], w_hdg = lim[2], a_hdg = lim[5];
  for (int i = 0; i < n_seg; ++i) {
    const double* s = wp + (size_t)i * 4;
    const double* e = wp + (size_t)(i + 1) * 4;
    const double dx = e[0] - s[0], dy = e[1] - s[1], dz = e[2] - s[2];
    const double inclinator = atan2(dz, sqrt(pow(dx, 2) + pow(dy, 2)));
    const double v_max = vmax_for_inclination(inclinator, v_v, v_h);
    const double distance = sqrt(dx * dx + dy * dy + dz * dz);
    double t = distance / v_max;
    if (t < 0.01) t = 0.01;
    const double hf = heading_fix_time(s[3], e[3], w_hdg, a_hdg, 1.0);
    if (hf > t) t = hf;
    out[i] = t;
  }
}

static void unit3(const double* a, const double* b, double* u) {
  double v[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
  const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  /* Eigen normalize(): divides only when the squared norm is > 0 */
  if (n * n > 0) {
    v[0] /= n;
    v[1] /= n;
    v[2] /= n;
  }
  u[0] = v[0];
  u[1] = v[1];
  u[2] = v[2];
}

void mto_estimate_segment_times_baca(int n_seg, const double* wp, const double* lim, double* out) {
  const double v_h = lim[0], v_v = lim[1], w_hdg = lim[2], a_h = lim[3], a_v = lim[4], a_hdg = lim[5], j_h = lim[6], j_v = lim[7];
  const int V = n_seg + 1;
  for (int i = 0; i < n_seg; ++i) {
    const double* s = wp + (size_t)i * 4;
    const double* e = wp + (size_t)(i + 1) * 4;
    const double dx = e[0] - s[0], dy = e[1] - s[1], dz = e[2] - s[2];
    const double distance = sqrt(dx * dx + dy * dy + dz * dz);
    const double inclinator = atan2(dz, sqrt(pow(dx, 2) + pow(dy, 2)));
    const double v_max = vmax_for_inclination(inclinator, v_v, v_h);
    const double a_max = vmax_for_inclination(inclinator, a_v, a_h);
    const double j_max = vmax_for_inclination(inclinator, j_v, j_h);
    double acc_t1 = 0, acc_t2 = 0, jerk_t1 = 0, jerk_t2 = 0;
    if (i >= 1) { /* vertex.cpp:355-376 */
      double u1[3], u2[3];
      unit3(wp + (size_t)(i - 1) * 4, s, u1);
      unit3(s, e, u2);
      const double dot = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
      const double c1 = 1 - (dot < 0 ? 0.0 : dot);
      acc_t1 = c1 * ((v_max / a_max) + (a_max / j_max));
      jerk_t1 = c1 * (2 * (a_max / j_max));
    }
    if (i == 0) { /* :379-383 */
      acc_t1 = (v_max / a_max) + (a_max / j_max);
      jerk_t1 = 2 * (a_max / j_max);
    }
    if (i == V - 2) { /* :386-390 */
      acc_t2 = (v_max / a_max) + (a_max / j_max);
      jerk_t2 = 2 * (a_max / j_max);
    }
    if (i < V - 2) { /* :393-414 */
      double u1[3], u2[3];
      unit3(s, e, u1);
      unit3(e, wp + (size_t)(i + 2) * 4, u2);
      const double dot = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
      const double c2 = 1 - (dot < 0 ? 0.0 : dot);
      acc_t2 = c2 * ((v_max / a_max) + (a_max / j_max));
      jerk_t2 = c2 * (2 * (a_max / j_max));
    }
    const double acc_cap = sqrt(2 * distance / a_max);
    if (acc_t1 > acc_cap) acc_t1 = acc_cap;
    if (acc_t2 > acc_cap) acc_t2 = acc_cap;
    (void)jerk_t1; /* the jerk times are computed but unused by the reference (:444-445) */
    (void)jerk_t2;
    const double max_velocity_time = distance / v_max; /* :442 overrides the branch above it */
    double t = max_velocity_time + acc_t1 + acc_t2;
    if (t < 0.01) t = 0.01;
    const double hf = heading_fix_time(s[3], e[3], w_hdg, a_hdg, 2.0);
    if (hf > t) t = hf;
    out[i] = t;
  }
}

/* ------------------------------------------------------------------------------------------- */
/* one path, and the batch driver                                                               */

/* 0 (default): the reference's behaviour -- the outer loop's own code whatever the feasibility scaling did to the times;
 * 1: the product's runaway rule (see solve_one). */
static int g_runaway_rule = 0;
void mto_set_runaway_rule(int on) { g_runaway_rule = on ? 1 : 0; }

static int solve_one(int S, const double* wp, const uint8_t* mask, const double* vals, const double* lim,
                     const mto_options* opt, double* times, double* coeffs, double* cost_out, int32_t* n_samples,
                     double* samples, int capacity) {
  mto_path path = {S, opt->derivative_to_optimize, mask, vals};
  int status = MTO_SUCCESS;
  if (opt->estimate_times) mto_estimate_segment_times_euclidean(S, wp, lim, times); /* :1046 */
  if (opt->time_alloc_method == 2) {
    double sum_t0 = 0.0, sum_t1 = 0.0;
    for (int i = 0; i < S; ++i) sum_t0 += times[i];
    int rc = mto_optimize_times_mellinger(&path, &opt->nlopt, times, NULL, NULL); /* :1083 -> nonlinear_impl.h:160 */
    status = rc;
    if (rc == MTO_INVALID_ARGS) {
      /* NLopt throws before any evaluation; the reference then returns FAILURE with an unsolved
       * optimiser (nonlinear_impl.h:193-197).  Documented deviation: solve at the given times. */
      status = MTO_FAILURE;
      mto_solve_linear(&path, times, coeffs);
    } else {
      /* scaleSegmentTimesWithViolation nonlinear_impl.h:336-408: trajectory of the last evaluated
       * point, per-segment scaling, then updateSegmentTimes + solveLinear with the new times */
      mto_solve_linear(&path, times, coeffs);
      mto_scale_segment_times_to_meet_constraints(S, coeffs, times, lim, NULL);
      mto_solve_linear(&path, times, coeffs);
      /* Not in the reference (it hands such a path back with the outer loop's code and leaves it to the nodelet's length
       * check, src/mrs_trajectory_generation.cpp:1178-1199) and therefore OFF by default here: with
       * mto_set_runaway_rule(1) a scaling that multiplied the total time by more than MTO_RUNAWAY_TIME_FACTOR is reported
       * as nlopt's ROUNDOFF_LIMITED, which the nodelet's gate rejects (:1103, :1146) -- the product's documented deviation
       * (include/mrs_tg.h), switched on by the tests that compare status words with the product's. */
      for (int i = 0; i < S; ++i) sum_t1 += times[i];
      if (g_runaway_rule && status > 0 && sum_t1 > MTO_RUNAWAY_TIME_FACTOR * sum_t0) status = MTO_ROUNDOFF_LIMITED;
    }
  } else if (opt->time_alloc_method == 0 || opt->time_alloc_method == 1) {
    /* optimizeTime (nonlinear_impl.h:121-157): no feasibility scaling afterwards; the trajectory is the one
     * of the last objective evaluation */
    mto_dfo_params dp = {opt->time_alloc_method, opt->nlopt, opt->time_penalty, opt->use_soft_constraints,
                         opt->soft_constraint_weight, opt->initial_stepsize_rel};
    int rc = mto_optimize_time_dfo(&path, lim, &dp, times, NULL, NULL);
    status = (rc == MTO_INVALID_ARGS) ? MTO_FAILURE : rc;
    mto_solve_linear(&path, times, coeffs);
  } else if (opt->time_alloc_method == 3 || opt->time_alloc_method == 4) {
    /* optimizeTimeAndFreeConstraints (nonlinear_impl.h:429-536): the trajectory is the one set by the last
     * objective evaluation (setFreeConstraints), no solve and no feasibility scaling afterwards */
    mto_dfo_params dp = {opt->time_alloc_method, opt->nlopt, opt->time_penalty, opt->use_soft_constraints,
                         opt->soft_constraint_weight, opt->initial_stepsize_rel};
    status = mto_optimize_time_and_constraints_dfo(&path, lim, &dp, times, coeffs, NULL, NULL);
  } else {
    if (mto_solve_linear(&path, times, coeffs) != 0) status = MTO_FAILURE;
  }
  if (cost_out) *cost_out = mto_compute_cost(S, opt->derivative_to_optimize, times, coeffs);
  if (opt->sampling_dt > 0 && n_samples) {
    const int cnt = mto_sample_trajectory(S, coeffs, times, opt->sampling_dt, 0, samples, capacity);
    *n_samples = cnt;
    if (samples) {
      const int lim_n = cnt < capacity ? cnt : capacity;
      for (int i = 0; i < lim_n; ++i) samples[(size_t)i * DIM + 3] = mto_wrap_yaw(samples[(size_t)i * DIM + 3]);
    }
  }
  return status;
}

typedef struct {
  int p0, p1;
  const int32_t* seg_offsets;
  const double* waypoints;
  const uint8_t* fixed_mask;
  const double* fixed_values;
  const double* limits;
  const mto_options* opt;
  double* seg_times;
  double* coeffs;
  int32_t* status;
  double* cost;
  int32_t* n_samples;
  double* samples;
  int capacity;
} batch_job;

static void run_paths(const batch_job* j, int p0, int p1) {
  for (int p = p0; p < p1; ++p) {
    const int s0 = j->seg_offsets[p], S = j->seg_offsets[p + 1] - s0, v0 = s0 + p;
    const int st = solve_one(S, j->waypoints + (size_t)v0 * 4, j->fixed_mask + (size_t)v0 * 5,
                             j->fixed_values + (size_t)v0 * 5 * 4, j->limits + (size_t)p * 9, j->opt, j->seg_times + s0,
                             j->coeffs + (size_t)s0 * DIM * N, j->cost ? j->cost + p : NULL,
                             j->n_samples ? j->n_samples + p : NULL,
                             j->samples ? j->samples + (size_t)p * j->capacity * DIM : NULL, j->capacity);
    if (j->status) j->status[p] = st;
  }
}

/* Persistent worker pool (bench.py's all-core CPU baseline): the workers are created once and parked on a condition
 * variable; a batch is handed out in chunks of paths from a shared counter, so the nonlinear path (whose cost per path
 * varies with the number of evaluations) balances itself and a call costs no thread creation. */
static struct {
  pthread_mutex_t mu;
  pthread_cond_t wake, idle;
  pthread_t* th;
  int n_threads;          /* workers alive */
  long long generation;   /* bumped once per batch */
  const batch_job* job;
  int next, chunk, busy, want; /* `want` workers (the lowest ids) take part in the current batch */
} g_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, NULL, 0, 0, NULL, 0, 1, 0, 0};

static int pool_take(int* p0, int* p1) {
  pthread_mutex_lock(&g_pool.mu);
  const int total = g_pool.job->p1;
  int ok = 0;
  if (g_pool.next < total) {
    *p0 = g_pool.next;
    *p1 = g_pool.next + g_pool.chunk < total ? g_pool.next + g_pool.chunk : total;
    g_pool.next = *p1;
    ok = 1;
  }
  pthread_mutex_unlock(&g_pool.mu);
  return ok;
}

static void* pool_worker(void* arg) {
  const int id = (int)(intptr_t)arg;
  long long seen = 0;
  for (;;) {
    pthread_mutex_lock(&g_pool.mu);
    while (g_pool.generation == seen) pthread_cond_wait(&g_pool.wake, &g_pool.mu);
    seen = g_pool.generation;
    const int take_part = id < g_pool.want;
    pthread_mutex_unlock(&g_pool.mu);
    if (!take_part) continue;
    int p0, p1;
    while (pool_take(&p0, &p1)) run_paths(g_pool.job, p0, p1);
    pthread_mutex_lock(&g_pool.mu);
    if (--g_pool.busy == 0) pthread_cond_signal(&g_pool.idle);
    pthread_mutex_unlock(&g_pool.mu);
  }
  return NULL;
}

int mto_solve_batch(int n_paths, const int32_t* seg_offsets, const double* waypoints, const uint8_t* fixed_mask,
                    const double* fixed_values, const double* limits, const mto_options* opt, double* seg_times_inout,
                    double* coeffs_out, int32_t* status_out, double* cost_out, int32_t* n_samples_out, double* samples_out,
                    int sample_capacity, int n_threads) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > n_paths) n_threads = n_paths > 0 ? n_paths : 1;
  const batch_job job = {0, n_paths, seg_offsets, waypoints, fixed_mask, fixed_values, limits, opt, seg_times_inout,
                         coeffs_out, status_out, cost_out, n_samples_out, samples_out, sample_capacity};
  if (n_threads == 1) {
    run_paths(&job, 0, n_paths);
    return 0;
  }
  static pthread_mutex_t call_mu = PTHREAD_MUTEX_INITIALIZER; /* one batch at a time through the pool */
  pthread_mutex_lock(&call_mu);
  pthread_mutex_lock(&g_pool.mu);
  if (g_pool.n_threads < n_threads) { /* grow the pool (never shrinks) */
    g_pool.th = (pthread_t*)realloc(g_pool.th, sizeof(pthread_t) * (size_t)n_threads);
    for (int t = g_pool.n_threads; t < n_threads; ++t) {
      if (pthread_create(&g_pool.th[t], NULL, pool_worker, (void*)(intptr_t)t) != 0) {
        n_threads = t; /* the system refuses more threads: go on with what exists */
        break;
      }
      pthread_detach(g_pool.th[t]);
    }
    if (n_threads > g_pool.n_threads) g_pool.n_threads = n_threads;
  }
  if (n_threads < 1) { /* not a single worker could be created */
    pthread_mutex_unlock(&g_pool.mu);
    pthread_mutex_unlock(&call_mu);
    run_paths(&job, 0, n_paths);
    return 0;
  }
  g_pool.job = &job;
  g_pool.next = 0;
  /* ~8 chunks per worker: small enough to balance, large enough that the shared counter is not contended */
  g_pool.chunk = n_paths / (8 * n_threads) > 0 ? n_paths / (8 * n_threads) : 1;
  g_pool.want = n_threads;
  g_pool.busy = n_threads;
  ++g_pool.generation;
  pthread_cond_broadcast(&g_pool.wake);
  while (g_pool.busy > 0) pthread_cond_wait(&g_pool.idle, &g_pool.mu);
  g_pool.job = NULL;
  pthread_mutex_unlock(&g_pool.mu);
  pthread_mutex_unlock(&call_mu);
  return 0;
}
