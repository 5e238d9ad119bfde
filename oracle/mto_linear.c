/*
 * mto_linear.c -- CPU ORACLE (test infrastructure): the linear QP of the reference, in the
 * reference's own arithmetic route.  See mrs_tg_oracle.h for the rules that apply to oracle/.
 *
 * Follows (relative to /root/reference/include/eth_trajectory_generation/):
 *   polynomial.h:208-237                       baseCoeffsWithTime
 *   impl/polynomial_optimization_linear_impl.h:113-121  setupMappingMatrix
 *   impl/...linear_impl.h:148-177              invertMappingMatrix (Schur complement)
 *   impl/...linear_impl.h:606-618              computeQuadraticCostJacobian
 *   impl/...linear_impl.h:184-257              setupConstraintReorderingMatrix
 *   impl/...linear_impl.h:311-373              constructR, solveLinear
 *   impl/...linear_impl.h:264-282              updateSegmentsFromCompactConstraints
 *   impl/...linear_impl.h:128-141              computeCost
 *
 * Third-party arithmetic the reference delegates to Eigen3 (absent here, version unpinned by the
 * reference's CMakeLists.txt:41) is restated with textbook algorithms: the fixed-size 5x5
 * .inverse() as LU with partial pivoting, SparseQR<COLAMD> as dense Householder QR.
 */
#include "mrs_tg_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define N MTO_N
#define HALF MTO_HALF
#define DIM MTO_D

/* row `derivative` of A at time t: B[derivative][j] * t^(j-derivative).  polynomial.h:208-237 */
static void base_coeffs_with_time(int derivative, double t, double* row) {
  for (int j = 0; j < N; ++j) row[j] = 0.0;
  row[derivative] = mto_base_coeff(derivative, derivative);
  if (fabs(t) < DBL_EPSILON) return;
  double t_power = t;
  for (int j = derivative + 1; j < N; ++j) {
    row[j] = mto_base_coeff(derivative, j) * t_power;
    t_power = t_power * t;
  }
}

void mto_mapping_matrix(double T, double* A) {
  for (int i = 0; i < HALF; ++i) {
    base_coeffs_with_time(i, 0.0, A + i * N);
    base_coeffs_with_time(i, T, A + (i + HALF) * N);
  }
}

/* inverse of a 5x5 by LU with partial pivoting (stands in for Eigen's fixed-size inverse) */
static void inverse5(const double* M, double* Minv) {
  double lu[HALF][HALF];
  int perm[HALF];
  for (int i = 0; i < HALF; ++i) {
    perm[i] = i;
    for (int j = 0; j < HALF; ++j) lu[i][j] = M[i * HALF + j];
  }
  for (int c = 0; c < HALF; ++c) {
    int piv = c;
    for (int r = c + 1; r < HALF; ++r)
      if (fabs(lu[r][c]) > fabs(lu[piv][c])) piv = r;
    if (piv != c) {
      for (int j = 0; j < HALF; ++j) {
        const double t = lu[c][j];
        lu[c][j] = lu[piv][j];
        lu[piv][j] = t;
      }
      const int t = perm[c];
      perm[c] = perm[piv];
      perm[piv] = t;
    }
    for (int r = c + 1; r < HALF; ++r) {
      lu[r][c] /= lu[c][c];
      for (int j = c + 1; j < HALF; ++j) lu[r][j] -= lu[r][c] * lu[c][j];
    }
  }
  for (int col = 0; col < HALF; ++col) {
    double y[HALF];
    for (int i = 0; i < HALF; ++i) {
      double s = (perm[i] == col) ? 1.0 : 0.0;
      for (int j = 0; j < i; ++j) s -= lu[i][j] * y[j];
      y[i] = s;
    }
    for (int i = HALF - 1; i >= 0; --i) {
      double s = y[i];
      for (int j = i + 1; j < HALF; ++j) s -= lu[i][j] * Minv[j * HALF + col];
      Minv[i * HALF + col] = s / lu[i][i];
    }
  }
}

void mto_invert_mapping_matrix(const double* A, double* Ainv) {
  /* [A_diag 0; C D]^-1 = [A_diag^-1 0; -D^-1 C A_diag^-1  D^-1]   linear_impl.h:159-176 */
  double a_inv[HALF], C[HALF * HALF], Dm[HALF * HALF], Dinv[HALF * HALF];
  for (int i = 0; i < HALF; ++i) a_inv[i] = 1.0 / A[i * N + i];
  for (int i = 0; i < HALF; ++i)
    for (int j = 0; j < HALF; ++j) {
      C[i * HALF + j] = A[(i + HALF) * N + j];
      Dm[i * HALF + j] = A[(i + HALF) * N + j + HALF];
    }
  inverse5(Dm, Dinv);
  memset(Ainv, 0, sizeof(double) * N * N);
  for (int i = 0; i < HALF; ++i) Ainv[i * N + i] = a_inv[i];
  /* -(D^-1 * C) * A_inv, evaluated left to right as the Eigen expression does */
  for (int i = 0; i < HALF; ++i)
    for (int j = 0; j < HALF; ++j) {
      double s = 0.0;
      for (int k = 0; k < HALF; ++k) s += (-Dinv[i * HALF + k]) * C[k * HALF + j];
      Ainv[(i + HALF) * N + j] = s * a_inv[j];
      Ainv[(i + HALF) * N + j + HALF] = Dinv[i * HALF + j];
    }
}

void mto_cost_matrix(int derivative, double T, double* Q) {
  /* linear_impl.h:606-618 */
  memset(Q, 0, sizeof(double) * N * N);
  for (int col = 0; col < N - derivative; ++col)
    for (int row = 0; row < N - derivative; ++row) {
      const double exponent = (N - 1 - derivative) * 2 + 1 - row - col;
      Q[(N - 1 - row) * N + (N - 1 - col)] =
          mto_base_coeff(derivative, N - 1 - row) * mto_base_coeff(derivative, N - 1 - col) * pow(T, exponent) * 2.0 / exponent;
    }
}

void mto_segment_hessian(int derivative, double T, double* Hout, double* Ainv_out) {
  /* H = Ai^T * Q * Ai with Ai the inverse mapping matrix (linear_impl.h:318-320) */
  double A[N * N], Ai[N * N], Q[N * N], tmp[N * N];
  mto_mapping_matrix(T, A);
  mto_invert_mapping_matrix(A, Ai);
  mto_cost_matrix(derivative, T, Q);
  for (int i = 0; i < N; ++i) /* tmp = Ai^T Q */
    for (int j = 0; j < N; ++j) {
      double s = 0.0;
      for (int k = 0; k < N; ++k) s += Ai[k * N + i] * Q[k * N + j];
      tmp[i * N + j] = s;
    }
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      double s = 0.0;
      for (int k = 0; k < N; ++k) s += tmp[i * N + k] * Ai[k * N + j];
      Hout[i * N + j] = s;
    }
  if (Ainv_out) memcpy(Ainv_out, Ai, sizeof(Ai));
}

/* dense Householder QR solve of M x = B (M n x n, B n x nrhs, both overwritten; X returned in B) */
static int qr_solve(double* M, int n, double* B, int nrhs) {
  const mto_scratch_state mark = mto_scratch_mark();
  double* v = (double*)mto_scratch_alloc(sizeof(double) * (size_t)n, 0);
  if (!v) return -1;
  for (int c = 0; c < n; ++c) {
    double norm = 0.0;
    for (int r = c; r < n; ++r) norm += M[r * n + c] * M[r * n + c];
    norm = sqrt(norm);
    if (norm == 0.0) {
      mto_scratch_release(mark);
      return -2;
    }
    const double alpha = (M[c * n + c] > 0) ? -norm : norm;
    for (int r = c; r < n; ++r) v[r] = M[r * n + c];
    v[c] -= alpha;
    double vnorm2 = 0.0;
    for (int r = c; r < n; ++r) vnorm2 += v[r] * v[r];
    if (vnorm2 > 0.0) {
      for (int j = c; j < n; ++j) {
        double dot = 0.0;
        for (int r = c; r < n; ++r) dot += v[r] * M[r * n + j];
        const double f = 2.0 * dot / vnorm2;
        for (int r = c; r < n; ++r) M[r * n + j] -= f * v[r];
      }
      for (int j = 0; j < nrhs; ++j) {
        double dot = 0.0;
        for (int r = c; r < n; ++r) dot += v[r] * B[r * nrhs + j];
        const double f = 2.0 * dot / vnorm2;
        for (int r = c; r < n; ++r) B[r * nrhs + j] -= f * v[r];
      }
    }
  }
  for (int j = 0; j < nrhs; ++j)
    for (int r = n - 1; r >= 0; --r) {
      double s = B[r * nrhs + j];
      for (int k = r + 1; k < n; ++k) s -= M[r * n + k] * B[k * nrhs + j];
      B[r * nrhs + j] = s / M[r * n + r];
    }
  mto_scratch_release(mark);
  return 0;
}

int mto_count_free_constraints(const mto_path* path) {
  /* n_free_constraints_: every (vertex, derivative 0..4) without a constraint (linear_impl.h:191-254) */
  int n = 0;
  for (int i = 0; i < HALF * (path->n_seg + 1); ++i) n += path->fixed_mask[i] ? 0 : 1;
  return n;
}

int mto_solve_linear(const mto_path* path, const double* seg_times, double* coeffs_out) {
  return mto_solve_linear_free(path, seg_times, coeffs_out, NULL);
}

int mto_coeffs_from_free_constraints(const mto_path* path, const double* seg_times, const double* free_in, double* coeffs_out) {
  /* setFreeConstraints (linear_impl.h:515-522) + updateSegmentsFromCompactConstraints (:264-282):
   * no solve, d = [d_f; d_p] with d_p given; free_in [4][n_free], free constraints ordered by (vertex, derivative) */
  const int S = path->n_seg, V = S + 1;
  if (S < 1 || S > MTO_MAX_SEG) return -1;
  const int n_all = HALF * V, n_free = mto_count_free_constraints(path);
  const mto_scratch_state mark = mto_scratch_mark();
  int* fidx = (int*)mto_scratch_alloc(sizeof(int) * (size_t)n_all, 0);
  if (!fidx) return -1;
  for (int i = 0, cp = 0; i < n_all; ++i) fidx[i] = path->fixed_mask[i] ? -1 : cp++;
  for (int i = 0; i < S; ++i) {
    double A[N * N], Ai[N * N];
    mto_mapping_matrix(seg_times[i], A);
    mto_invert_mapping_matrix(A, Ai);
    for (int k = 0; k < DIM; ++k) {
      double dseg[N];
      for (int r = 0; r < N; ++r) {
        const int u = (i + r / HALF) * HALF + r % HALF;
        dseg[r] = path->fixed_mask[u] ? path->fixed_values[(size_t)u * DIM + k] : free_in[(size_t)k * n_free + fidx[u]];
      }
      double* c = coeffs_out + ((size_t)i * DIM + k) * N;
      for (int r = 0; r < N; ++r) {
        double acc = 0.0;
        for (int q = 0; q < N; ++q) acc += Ai[r * N + q] * dseg[q];
        c[r] = acc;
      }
    }
  }
  mto_scratch_release(mark);
  return 0;
}

int mto_solve_linear_free(const mto_path* path, const double* seg_times, double* coeffs_out, double* free_out) {
  /* free_out (may be NULL): getFreeConstraints, [4][n_free] (free_constraints_compact_, linear_impl.h:360-369) */
  const int S = path->n_seg, V = S + 1, d = path->derivative_to_optimize;
  if (S < 1 || S > MTO_MAX_SEG) return -1;
  const int n_all = HALF * V;
  /* column index of each (vertex, slot): fixed ones first, then free, each sorted by
   * (vertex, derivative) like the std::set<Constraint> walk at linear_impl.h:191-254 */
  const mto_scratch_state mark = mto_scratch_mark();
  int* col = (int*)mto_scratch_alloc(sizeof(int) * (size_t)n_all, 0);
  int n_fixed = 0, n_free = 0;
  for (int i = 0; i < n_all; ++i) n_fixed += path->fixed_mask[i] ? 1 : 0;
  n_free = n_all - n_fixed;
  {
    int cf = 0, cp = n_fixed;
    for (int i = 0; i < n_all; ++i) col[i] = path->fixed_mask[i] ? cf++ : cp++;
  }
  /* R = C^T blkdiag(H_i) C: row (10 i + r) of C selects unknown (vertex i + r/5, slot r%5) */
  double* R = (double*)mto_scratch_alloc(sizeof(double) * (size_t)n_all * (size_t)n_all, 1);
  double* Ainv = (double*)mto_scratch_alloc(sizeof(double) * (size_t)S * N * N, 0);
  double* dall = (double*)mto_scratch_alloc(sizeof(double) * (size_t)n_all * DIM, 1);
  if (!col || !R || !Ainv || !dall) {
    mto_scratch_release(mark);
    return -1;
  }
  double Hm[N * N];
  for (int i = 0; i < S; ++i) {
    mto_segment_hessian(d, seg_times[i], Hm, Ainv + (size_t)i * N * N);
    for (int r = 0; r < N; ++r) {
      const int cr = col[(i + r / HALF) * HALF + r % HALF];
      for (int c = 0; c < N; ++c) {
        const int cc = col[(i + c / HALF) * HALF + c % HALF];
        R[(size_t)cr * n_all + cc] += Hm[r * N + c];
      }
    }
  }
  /* d_f per dimension, then d_p = -Rpp^-1 Rpf d_f  (linear_impl.h:360-369) */
  for (int i = 0; i < n_all; ++i)
    if (path->fixed_mask[i])
      for (int k = 0; k < DIM; ++k) dall[(size_t)col[i] * DIM + k] = path->fixed_values[(size_t)i * DIM + k];
  int rc = 0;
  if (n_free > 0) {
    double* Rpp = (double*)mto_scratch_alloc(sizeof(double) * (size_t)n_free * (size_t)n_free, 0);
    double* rhs = (double*)mto_scratch_alloc(sizeof(double) * (size_t)n_free * DIM, 0);
    if (!Rpp || !rhs) {
      mto_scratch_release(mark);
      return -1;
    }
    for (int r = 0; r < n_free; ++r) {
      for (int c = 0; c < n_free; ++c) Rpp[(size_t)r * n_free + c] = R[(size_t)(n_fixed + r) * n_all + n_fixed + c];
      for (int k = 0; k < DIM; ++k) {
        double s = 0.0;
        for (int c = 0; c < n_fixed; ++c) s += -R[(size_t)(n_fixed + r) * n_all + c] * dall[(size_t)c * DIM + k];
        rhs[(size_t)r * DIM + k] = s;
      }
    }
    rc = qr_solve(Rpp, n_free, rhs, DIM);
    for (int r = 0; r < n_free; ++r)
      for (int k = 0; k < DIM; ++k) {
        dall[(size_t)(n_fixed + r) * DIM + k] = rhs[(size_t)r * DIM + k];
        if (free_out) free_out[(size_t)k * n_free + r] = rhs[(size_t)r * DIM + k];
      }
  }
  /* coefficients: c = A_i^-1 (C_i d)   linear_impl.h:264-282 */
  for (int i = 0; i < S; ++i)
    for (int k = 0; k < DIM; ++k) {
      double dseg[N];
      for (int r = 0; r < N; ++r) dseg[r] = dall[(size_t)col[(i + r / HALF) * HALF + r % HALF] * DIM + k];
      double* c = coeffs_out + ((size_t)i * DIM + k) * N;
      for (int r = 0; r < N; ++r) {
        double s = 0.0;
        for (int q = 0; q < N; ++q) s += Ainv[(size_t)i * N * N + r * N + q] * dseg[q];
        c[r] = s;
      }
    }
  mto_scratch_release(mark);
  return rc;
}

double mto_compute_cost(int n_seg, int derivative, const double* seg_times, const double* coeffs) {
  /* 0.5 * sum_seg sum_dim c^T Q c   linear_impl.h:128-141 */
  double cost = 0.0, Q[N * N];
  for (int i = 0; i < n_seg; ++i) {
    mto_cost_matrix(derivative, seg_times[i], Q);
    for (int k = 0; k < DIM; ++k) {
      const double* c = coeffs + ((size_t)i * DIM + k) * N;
      double partial = 0.0;
      for (int r = 0; r < N; ++r) {
        double s = 0.0;
        for (int q = 0; q < N; ++q) s += Q[r * N + q] * c[q];
        partial += c[r] * s;
      }
      cost += partial;
    }
  }
  return 0.5 * cost;
}
