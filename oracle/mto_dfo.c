/*
 * mto_dfo.c -- CPU ORACLE (test infrastructure): the gradient-free time-allocation modes 0 / 1
 * (kSquaredTime, kRichterTime) and 3 / 4 (kSquaredTimeAndConstraints, kRichterTimeAndConstraints).
 * See mrs_tg_oracle.h for the rules that apply to oracle/.
 *
 * Follows (relative to /root/reference/include/eth_trajectory_generation/impl/):
 *   polynomial_optimization_nonlinear_impl.h:121-157   optimizeTime (NLopt driver set-up: bounds, initial step)
 *   polynomial_optimization_nonlinear_impl.h:568-614   objectiveFunctionTime
 *   polynomial_optimization_nonlinear_impl.h:429-536   optimizeTimeAndFreeConstraints (variables, bounds, steps)
 *   polynomial_optimization_nonlinear_impl.h:651-722   objectiveFunctionTimeAndConstraints
 *   polynomial_optimization_nonlinear_impl.h:765-804   setFreeEndpointDerivativeHardConstraints
 *   polynomial_optimization_nonlinear_impl.h:725-762   evaluateMaximumMagnitudeConstraint / ...AsSoftConstraint
 *   polynomial_optimization_linear_impl.h:478-508      computeMaximumOfMagnitude
 *   src/mrs_trajectory_generation.cpp:1067-1081        the 12 registered magnitude constraints
 *
 * The reference minimises this objective with NLopt's LN_BOBYQA (un-vendored).  BOBYQA is not restated; the
 * driver below is this project's own deterministic derivative-free search ("MRS-DFO"),
 * which the HIP path implements identically: for the time-only modes 0 / 1 a greedy coordinate search with step doubling /
 * halving (dfo_step), for modes 3 / 4 Powell's decision-free initial interpolation sweep + compass search (dfo_step_sweep,
 * and why); either spends its last evaluation on the best point it has found.  As with mode 2 the point that is kept is the LAST EVALUATED one
 * (the optimiser object's state) -- which is why the search ends there -- and there is no feasibility scaling in these
 * modes.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mrs_tg_oracle.h"

#define N MTO_N
#define DIM MTO_D

double mto_max_of_magnitude(int n_seg, const double* coeffs, const double* seg_times, int derivative) {
  /* computeMaximumOfMagnitude: per segment the candidates {0} + {t_start, t_end, real roots in range} of the
   * 4-D norm, and finally the end of the last segment (linear_impl.h:478-508) */
  static const int dims[4] = {0, 1, 2, 3};
  double best = -DBL_MAX;
  for (int s = 0; s < n_seg; ++s) {
    /* mto_segment_max_magnitude covers {0, T} and the roots; the extra leading 0.0 candidate is a duplicate */
    const double m = mto_segment_max_magnitude(coeffs + (size_t)s * DIM * N, seg_times[s], derivative, dims, 4);
    if (m > best) best = m;
  }
  return best;
}

double mto_soft_constraint_cost(const double maxima[3], const double* limits9, double weight) {
  /* 12 constraints: dimensions 0,1 -> horizontal limits, 2 -> vertical, 3 -> heading; every one of them is
   * evaluated with the 4-D maximum of its derivative (quirk B6) */
  double cost = 0.0;
  for (int dim = 0; dim < 4; ++dim) {
    const int grp = (dim <= 1) ? 0 : (dim == 2 ? 1 : 2);
    for (int k = 0; k < 3; ++k) {
      const double value = limits9[k * 3 + grp];
      const double rel = (maxima[k] - value) / value;
      const double c = exp(rel * weight);
      cost += (c < 1.0e12) ? c : 1.0e12;
    }
  }
  return cost;
}

double mto_objective_time(const mto_path* path, const double* seg_times, const double* limits9, const mto_dfo_params* prm,
                          double* parts_out /* [3] trajectory, time, soft; may be NULL */) {
  const int S = path->n_seg;
  double* coeffs = (double*)malloc(sizeof(double) * (size_t)S * DIM * N);
  mto_solve_linear(path, seg_times, coeffs);
  const double cost_traj = mto_compute_cost(S, path->derivative_to_optimize, seg_times, coeffs);
  double total = 0.0;
  for (int i = 0; i < S; ++i) total += seg_times[i];
  const double cost_time = (prm->time_alloc_method == 1 || prm->time_alloc_method == 4) ? total * prm->time_penalty : total * total * prm->time_penalty;
  double cost_soft = 0.0;
  if (prm->use_soft_constraints) {
    double mx[3];
    for (int k = 1; k <= 3; ++k) mx[k - 1] = mto_max_of_magnitude(S, coeffs, seg_times, k);
    cost_soft = mto_soft_constraint_cost(mx, limits9, prm->soft_constraint_weight);
  }
  free(coeffs);
  if (parts_out) {
    parts_out[0] = cost_traj;
    parts_out[1] = cost_time;
    parts_out[2] = cost_soft;
  }
  return cost_traj + cost_time + cost_soft;
}

static int relstop(double vold, double vnew, double reltol, double abstol) {
  if (isinf(vold)) return 0;
  const double dv = fabs(vnew - vold);
  return dv < abstol || dv < reltol * (fabs(vnew) + fabs(vold)) * 0.5 || (reltol > 0 && vnew == vold);
}

/* ---- MRS-DFO as an explicit state machine: one objective evaluation per step -------------------------- */
/* The search: a greedy coordinate search whose last evaluation is its best point.
 *   - evaluate x0; then for every coordinate i in turn try best + h_i e_i; a trial that improves on the best value is
 *     (by more than 1e-6 of it) is accepted at once (the search moves), h_i doubles (and stays doubled) and the same direction is tried again; when
 *     the first trial in the + direction fails, - h_i is tried the same way; then the next coordinate;
 *   - a sweep over all coordinates that improved: NLopt's relstop rule on f (-> FTOL_REACHED); one that did not: every h
 *     halves, and the search stops with XTOL_REACHED when every h_i < x_abs or < x_rel |best_i|;
 *   - the reference keeps the LAST EVALUATED point (its optimiser object's state), so the search spends its last
 *     evaluation -- the budget's, or one more after a tolerance stop -- on the best point it has found.
 * (Round 4: replaces Powell's initial sweep x0 +- h_i e_i + compass search, whose kept point -- x0 + h e_9 with the
 * shipping budget of 10 evaluations -- was on average no better than the start; measured against scipy's Powell and COBYLA
 * in tests/test_dfo_quality.py.) */

enum { DFO_FIRST = -1, DFO_SEARCH = 0, DFO_REVISIT = 1, DFO_INIT_PLUS = 2, DFO_INIT_MINUS = 3, DFO_COMPASS = 4 };
/* a trial is accepted when it lowers the best value by more than this share of it: decisions then do not hang on the last
 * digits of f (two correct evaluations of f differ by ~1e-9 relative through their linear solves; a trial that moves a free
 * derivative of value 0 by 1e-13 changes f by less than that) */
static const double kDfoMinDecrease = 1.0e-6;

typedef struct {
  int phase, i, sg, neval, improved, ret, done, acc_any, pending, last_is_best;
  double fbest, f_sweep;
} dfo_state;

static double clampd(double t, double lo, double hi) { return t < lo ? lo : (t > hi ? hi : t); }

/* consume the objective value f of the trial held in x (the last evaluated point); write the next trial into x.
 * best, h, lb, ub are the per-path vectors of the search (n variables). */
static void dfo_step(dfo_state* st, int n, double f, double* x, double* best, double* h, const double* lb,
                     const double* ub, const mto_dfo_params* prm) {
  const int maxeval = prm->nlopt.max_iterations;
  int accepted = 0;
  st->neval++;
  if (st->phase == DFO_REVISIT) { /* back on the best point: that was the search's last evaluation */
    st->ret = st->pending;
    st->done = 1;
    return;
  }
  if (st->phase == DFO_FIRST) {
    st->fbest = f;
    st->last_is_best = 1;
  } else if (f < st->fbest - kDfoMinDecrease * fabs(st->fbest)) {
    st->fbest = f;
    memcpy(best, x, sizeof(double) * (size_t)n);
    st->improved = 1;
    accepted = 1;
    st->last_is_best = 1;
    h[st->i] *= 2.0;
  } else {
    st->last_is_best = 0;
  }
  if (maxeval > 0 && st->neval >= maxeval) {
    st->ret = MTO_MAXEVAL_REACHED;
    st->done = 1;
    return;
  }
  for (;;) {
    int stop = 0;
    if (st->phase == DFO_FIRST) {
      st->phase = DFO_SEARCH;
      st->i = 0;
      st->sg = 0;
      st->acc_any = 0;
      st->improved = 0;
      st->f_sweep = st->fbest;
    } else if (accepted) {
      st->acc_any = 1; /* same coordinate, same direction, doubled step */
    } else if (!st->acc_any && st->sg == 0) {
      st->sg = 1; /* the first + trial failed: the other direction */
    } else {
      st->sg = 0;
      st->acc_any = 0;
      if (++st->i >= n) { /* end of a sweep */
        if (st->improved) {
          if (relstop(st->f_sweep, st->fbest, prm->nlopt.f_rel, prm->nlopt.f_abs)) stop = MTO_FTOL_REACHED;
        } else {
          int all_small = 1;
          for (int k = 0; k < n; ++k) {
            h[k] *= 0.5;
            if (!(h[k] < prm->nlopt.x_abs || h[k] < prm->nlopt.x_rel * fabs(best[k]))) all_small = 0;
          }
          if (all_small) stop = MTO_XTOL_REACHED;
        }
        st->f_sweep = st->fbest;
        st->improved = 0;
        st->i = 0;
      }
    }
    accepted = 0;
    if (stop) {
      if (st->last_is_best) {
        st->ret = stop;
        st->done = 1;
        return;
      }
      st->phase = DFO_REVISIT;
      st->pending = stop;
      memcpy(x, best, sizeof(double) * (size_t)n);
      return;
    }
    if (maxeval > 0 && st->neval >= maxeval - 1) { /* the budget's last evaluation belongs to the best point */
      st->phase = DFO_REVISIT;
      st->pending = MTO_MAXEVAL_REACHED;
      memcpy(x, best, sizeof(double) * (size_t)n);
      return;
    }
    const int i = st->i;
    const double t = clampd(best[i] + (st->sg == 0 ? h[i] : -h[i]), lb[i], ub[i]);
    if (t == best[i]) continue; /* nothing to try in this direction: as a failed trial */
    memcpy(x, best, sizeof(double) * (size_t)n);
    x[i] = t;
    return;
  }
}

/* Modes 3 / 4 (variables = segment times AND free end-point derivatives).  Their objective moves by ~1e-3 when the
 * held derivatives move by 1e-9 (a snap cost at times the derivatives were not solved for), so two correct implementations
 * disagree on f in the third digit and any search whose trial points depend on comparisons of f takes different roads in
 * them.  These modes therefore keep the DECISION-FREE start: Powell's initial interpolation sweep x0 + h_i e_i, then
 * x0 - h_i e_i (the points BOBYQA itself starts with; 2 n + 1 = 309 evaluations for a 10-segment min-snap path, i.e. every
 * practical budget), then a compass search with step halving -- and, like the greedy search, the last evaluation goes back
 * to the best point found (one comparison per evaluated point, not a road). */
static void dfo_step_sweep(dfo_state* st, int n, double f, double* x, double* x0, double* best, double* h, const double* lb,
                           const double* ub, const mto_dfo_params* prm) {
  const int maxeval = prm->nlopt.max_iterations;
  int accepted = 0;
  st->neval++;
  if (st->phase == DFO_REVISIT) {
    st->ret = st->pending;
    st->done = 1;
    return;
  }
  if (st->phase == DFO_FIRST) {
    st->fbest = f;
    st->last_is_best = 1;
  } else if (f < st->fbest) {
    st->fbest = f;
    memcpy(best, x, sizeof(double) * (size_t)n);
    st->improved = 1;
    accepted = 1;
    st->last_is_best = 1;
  } else {
    st->last_is_best = 0;
  }
  if (maxeval > 0 && st->neval >= maxeval) {
    st->ret = MTO_MAXEVAL_REACHED;
    st->done = 1;
    return;
  }
  for (;;) {
    int stop = 0;
    if (st->phase == DFO_FIRST) {
      st->phase = DFO_INIT_PLUS;
      st->i = 0;
    } else if (st->phase == DFO_INIT_PLUS) {
      if (++st->i >= n) {
        st->phase = DFO_INIT_MINUS;
        st->i = 0;
      }
    } else if (st->phase == DFO_INIT_MINUS) {
      if (++st->i >= n) {
        st->phase = DFO_COMPASS;
        for (int k = 0; k < n; ++k) h[k] *= 0.5;
        st->i = 0;
        st->sg = 0;
        st->f_sweep = st->fbest;
        st->improved = 0;
        accepted = 0;
      }
    } else { /* compass: advance (coordinate, sign) */
      if (st->sg == 0 && !accepted) {
        st->sg = 1;
      } else {
        st->sg = 0;
        ++st->i;
      }
      accepted = 0;
      if (st->i >= n) { /* end of a sweep */
        if (st->improved) {
          if (relstop(st->f_sweep, st->fbest, prm->nlopt.f_rel, prm->nlopt.f_abs)) stop = MTO_FTOL_REACHED;
        } else {
          int all_small = 1;
          for (int k = 0; k < n; ++k) {
            h[k] *= 0.5;
            if (!(h[k] < prm->nlopt.x_abs || h[k] < prm->nlopt.x_rel * fabs(best[k]))) all_small = 0;
          }
          if (all_small) stop = MTO_XTOL_REACHED;
        }
        st->f_sweep = st->fbest;
        st->improved = 0;
        st->i = 0;
        st->sg = 0;
      }
    }
    if (stop) {
      if (st->last_is_best) {
        st->ret = stop;
        st->done = 1;
        return;
      }
      st->phase = DFO_REVISIT;
      st->pending = stop;
      memcpy(x, best, sizeof(double) * (size_t)n);
      return;
    }
    if (maxeval > 0 && st->neval >= maxeval - 1) { /* the budget's last evaluation belongs to the best point */
      st->phase = DFO_REVISIT;
      st->pending = MTO_MAXEVAL_REACHED;
      memcpy(x, best, sizeof(double) * (size_t)n);
      return;
    }
    /* build the trial of the current (phase, i, sg) */
    const int i = st->i;
    if (st->phase == DFO_INIT_PLUS) {
      memcpy(x, x0, sizeof(double) * (size_t)n);
      x[i] = clampd((x0[i] + h[i] <= ub[i]) ? x0[i] + h[i] : x0[i] - h[i], lb[i], ub[i]);
      return;
    }
    if (st->phase == DFO_INIT_MINUS) {
      memcpy(x, x0, sizeof(double) * (size_t)n);
      x[i] = clampd((x0[i] - h[i] >= lb[i]) ? x0[i] - h[i] : x0[i] + 2.0 * h[i], lb[i], ub[i]);
      return;
    }
    const double t = clampd(best[i] + (st->sg == 0 ? h[i] : -h[i]), lb[i], ub[i]);
    if (t == best[i]) continue; /* nothing to try in this direction */
    memcpy(x, best, sizeof(double) * (size_t)n);
    x[i] = t;
    return;
  }
}

int mto_optimize_time_dfo(const mto_path* path, const double* limits9, const mto_dfo_params* prm, double* x,
                          int* n_eval_out, double* f_last_out) {
  const int n = path->n_seg;
  double h[MTO_MAX_SEG], best[MTO_MAX_SEG], lb[MTO_MAX_SEG], ub[MTO_MAX_SEG];
  for (int i = 0; i < n; ++i) {
    if (x[i] < 0.01) { /* NLopt rejects a start outside the bounds */
      if (n_eval_out) *n_eval_out = 0;
      return MTO_INVALID_ARGS;
    }
    best[i] = x[i];
    h[i] = prm->initial_stepsize_rel * x[i]; /* nonlinear_impl.h:127-130 */
    lb[i] = 0.01;                            /* kOptimizationTimeLowerBound, :134-135 */
    ub[i] = DBL_MAX;
  }
  dfo_state st = {DFO_FIRST, 0, 0, 0, 0, MTO_FAILURE, 0, 0, 0, 0, 0.0, 0.0};
  double f = 0.0;
  while (!st.done) {
    f = mto_objective_time(path, x, limits9, prm, NULL);
    dfo_step(&st, n, f, x, best, h, lb, ub, prm);
  }
  if (n_eval_out) *n_eval_out = st.neval;
  if (f_last_out) *f_last_out = f;
  return st.ret;
}

/* ---- modes 3 / 4: segment times and free end-point derivatives ---------------------------------------- */

double mto_objective_time_and_constraints(const mto_path* path, const double* x, const double* limits9,
                                          const mto_dfo_params* prm, double* coeffs_out, double* parts_out) {
  /* objectiveFunctionTimeAndConstraints nonlinear_impl.h:651-722: x = [T_0..T_{S-1}, free constraints of
   * dimension 0, 1, 2, 3]; updateSegmentTimes + setFreeConstraints, no solve */
  const int S = path->n_seg;
  double* coeffs = coeffs_out ? coeffs_out : (double*)malloc(sizeof(double) * (size_t)S * DIM * N);
  mto_coeffs_from_free_constraints(path, x, x + S, coeffs);
  const double cost_traj = mto_compute_cost(S, path->derivative_to_optimize, x, coeffs);
  double total = 0.0;
  for (int i = 0; i < S; ++i) total += x[i];
  const double cost_time = (prm->time_alloc_method == 4) ? total * prm->time_penalty : total * total * prm->time_penalty;
  double cost_soft = 0.0;
  if (prm->use_soft_constraints) {
    double mx[3];
    for (int k = 1; k <= 3; ++k) mx[k - 1] = mto_max_of_magnitude(S, coeffs, x, k);
    cost_soft = mto_soft_constraint_cost(mx, limits9, prm->soft_constraint_weight);
  }
  if (!coeffs_out) free(coeffs);
  if (parts_out) {
    parts_out[0] = cost_traj;
    parts_out[1] = cost_time;
    parts_out[2] = cost_soft;
  }
  return cost_traj + cost_time + cost_soft;
}

void mto_free_derivative_bounds(const mto_path* path, const double* limits9, double* lower, double* upper) {
  /* setFreeEndpointDerivativeHardConstraints nonlinear_impl.h:765-804 over the 12 constraints registered at
   * src/mrs_trajectory_generation.cpp:1067-1081.  The counter walks derivatives 0..derivative_to_optimize only,
   * while the free constraints cover derivatives 0..4: for derivative_to_optimize < 4 the bound lands on a
   * different free constraint than the one it was meant for.  Replicated as written. */
  const int V = path->n_seg + 1, n_free = mto_count_free_constraints(path), d = path->derivative_to_optimize;
  for (int i = 0; i < DIM * n_free; ++i) {
    lower[i] = -DBL_MAX; /* std::numeric_limits<double>::lowest() */
    upper[i] = DBL_MAX;
  }
  for (int dim = 0; dim < DIM; ++dim) {
    const int grp = (dim <= 1) ? 0 : (dim == 2 ? 1 : 2);
    for (int k = 1; k <= 3; ++k) {
      const double value = limits9[(k - 1) * 3 + grp];
      int counter = 0;
      for (int v = 0; v < V; ++v)
        for (int deriv = 0; deriv <= d; ++deriv)
          if (!path->fixed_mask[v * MTO_HALF + deriv]) {
            if (deriv == k) {
              lower[dim * n_free + counter] = -fabs(value);
              upper[dim * n_free + counter] = fabs(value);
            }
            counter++;
          }
    }
  }
}

int mto_optimize_time_and_constraints_dfo(const mto_path* path, const double* limits9, const mto_dfo_params* prm,
                                          double* seg_times, double* coeffs_out, int* n_eval_out, double* f_last_out) {
  /* optimizeTimeAndFreeConstraints nonlinear_impl.h:429-536 */
  const int S = path->n_seg, n_free = mto_count_free_constraints(path), n = S + DIM * n_free;
  double* buf = (double*)malloc(sizeof(double) * (size_t)n * 6);
  double *x = buf, *x0 = buf + n, *best = buf + 2 * n, *h = buf + 3 * n, *lb = buf + 4 * n, *ub = buf + 5 * n;
  memcpy(x, seg_times, sizeof(double) * (size_t)S);
  mto_solve_linear_free(path, seg_times, coeffs_out, x + S); /* initial solution :436-438 */
  for (int i = 0; i < S; ++i) {
    lb[i] = 0.01;
    ub[i] = DBL_MAX;
  }
  mto_free_derivative_bounds(path, limits9, lb + S, ub + S);
  for (int i = 0; i < n; ++i) {
    const double ax = fabs(x[i]);
    h[i] = (ax <= DBL_EPSILON) ? 1e-13 : prm->initial_stepsize_rel * ax; /* :486-494 */
    if (x[i] < lb[i]) lb[i] = x[i];                                       /* :496-501 */
    else if (x[i] > ub[i]) ub[i] = x[i];
    x0[i] = best[i] = x[i];
  }
  dfo_state st = {DFO_FIRST, 0, 0, 0, 0, MTO_FAILURE, 0, 0, 0, 0, 0.0, 0.0};
  double f = 0.0;
  while (!st.done) {
    f = mto_objective_time_and_constraints(path, x, limits9, prm, coeffs_out, NULL);
    memcpy(seg_times, x, sizeof(double) * (size_t)S); /* the optimiser object holds the last evaluated point */
    dfo_step_sweep(&st, n, f, x, x0, best, h, lb, ub, prm);
  }
  if (n_eval_out) *n_eval_out = st.neval;
  if (f_last_out) *f_last_out = f;
  free(buf);
  return st.ret;
}
