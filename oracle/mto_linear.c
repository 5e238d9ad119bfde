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
#include <pthread.h>
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

/* ---- second arithmetic route: exactly rounded unit-time constants -------------------------------------------
 * The route above is the reference's: A(T) inverted numerically, Q(T) from pow(), two dense products -- its results carry
 * an error of cond(A) eps ~ 1e-10 .. 1e-8, which the forward-difference gradient of the outer loop (h = 0.1)
 * amplifies.  With mto_set_arithmetic(1) every per-segment matrix comes from the unit-time tables instead:
 *     H(T)[a][b]    = HBAR_d[a][b] T^(a%5 + b%5 + 1 - 2d),   HBAR_d = ABAR^-T QBAR_d ABAR^-1
 *     A^-1(T)[k][j] = ABAR^-1[k][j] T^(j%5 - k)
 * (the same matrices, mathematically), the tables computed once in 113-bit arithmetic (__float128) from the reference's own
 * formulas and rounded to double.  Everything downstream (dense R, QR, coefficients, cost) is unchanged.  Tests use it to
 * separate "the HIP path differs from the reference's algorithm" from "the reference-style arithmetic is noisy". */
static int g_arith = 0;
void mto_set_arithmetic(int mode) { g_arith = (mode == 1 || mode == 2) ? mode : 0; }
int mto_get_arithmetic(void) { return g_arith; }

typedef __float128 q_t;
static double g_abar_inv[N][N], g_hbar[HALF][N][N];
static q_t g_abar_inv_q[N][N], g_hbar_q[HALF][N][N];
static pthread_once_t g_tables_once = PTHREAD_ONCE_INIT;

static void build_unit_tables(void) {
  q_t A[N][2 * N];
  /* ABAR: rows 0..4 = derivative r at t = 0, rows 5..9 = derivative r at t = 1 (setupMappingMatrix with T = 1) */
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < 2 * N; ++j) A[i][j] = (j == N + i) ? 1 : 0;
  for (int r = 0; r < HALF; ++r) {
    A[r][r] = (q_t)mto_base_coeff(r, r);
    for (int j = r; j < N; ++j) A[HALF + r][j] = (q_t)mto_base_coeff(r, j);
  }
  for (int c = 0; c < N; ++c) { /* Gauss-Jordan with partial pivoting on [A | I] */
    int piv = c;
    for (int r = c + 1; r < N; ++r) {
      const q_t a = A[r][c] < 0 ? -A[r][c] : A[r][c], b = A[piv][c] < 0 ? -A[piv][c] : A[piv][c];
      if (a > b) piv = r;
    }
    for (int j = 0; j < 2 * N; ++j) {
      const q_t t = A[c][j];
      A[c][j] = A[piv][j];
      A[piv][j] = t;
    }
    const q_t inv = (q_t)1 / A[c][c];
    for (int j = 0; j < 2 * N; ++j) A[c][j] *= inv;
    for (int r = 0; r < N; ++r) {
      if (r == c) continue;
      const q_t f = A[r][c];
      for (int j = 0; j < 2 * N; ++j) A[r][j] -= f * A[c][j];
    }
  }
  q_t Ai[N][N];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      Ai[i][j] = A[i][N + j];
      g_abar_inv_q[i][j] = Ai[i][j];
      g_abar_inv[i][j] = (double)Ai[i][j];
    }
  for (int d = 0; d < HALF; ++d) {
    q_t Q[N][N], tmp[N][N];
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j)
        Q[i][j] = (i >= d && j >= d) ? (q_t)mto_base_coeff(d, i) * (q_t)mto_base_coeff(d, j) * 2 / (q_t)(i + j - 2 * d + 1) : 0;
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        q_t acc = 0;
        for (int k = 0; k < N; ++k) acc += Ai[k][i] * Q[k][j];
        tmp[i][j] = acc;
      }
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        q_t acc = 0;
        for (int k = 0; k < N; ++k) acc += tmp[i][k] * Ai[k][j];
        g_hbar_q[d][i][j] = acc;
        g_hbar[d][i][j] = (double)acc;
      }
  }
}

void mto_unit_tables(double* abar_inv_out, double* hbar_out) {
  pthread_once(&g_tables_once, build_unit_tables);
  if (abar_inv_out) memcpy(abar_inv_out, g_abar_inv, sizeof(g_abar_inv));
  if (hbar_out) memcpy(hbar_out, g_hbar, sizeof(g_hbar));
}

/* T^e for e in [-9, 9] */
static void time_powers(double T, double* tp /* [19], tp[9 + e] */) {
  tp[9] = 1.0;
  for (int e = 1; e <= 9; ++e) tp[9 + e] = tp[9 + e - 1] * T;
  const double ti = 1.0 / T;
  for (int e = 1; e <= 9; ++e) tp[9 - e] = tp[9 - e + 1] * ti;
}

static void inverse_mapping_from_tables(double T, double* Ai) {
  pthread_once(&g_tables_once, build_unit_tables);
  double tp[19];
  time_powers(T, tp);
  for (int k = 0; k < N; ++k)
    for (int j = 0; j < N; ++j) Ai[k * N + j] = g_abar_inv[k][j] * tp[9 + (j % HALF) - k];
}

/* A^-1(T) in the arithmetic route in force */
static void segment_inverse_mapping(double T, double* Ai) {
  if (g_arith) {  /* (route 2 as well: mto_coeffs_from_free_constraints is a plain product) */
    inverse_mapping_from_tables(T, Ai);
    return;
  }
  double A[N * N];
  mto_mapping_matrix(T, A);
  mto_invert_mapping_matrix(A, Ai);
}

void mto_segment_hessian(int derivative, double T, double* Hout, double* Ainv_out) {
  if (g_arith) {
    pthread_once(&g_tables_once, build_unit_tables);
    double tp[19];
    time_powers(T, tp);
    for (int a = 0; a < N; ++a)
      for (int b = 0; b < N; ++b)
        Hout[a * N + b] = g_hbar[derivative][a][b] * tp[9 + (a % HALF) + (b % HALF) + 1 - 2 * derivative];
    if (Ainv_out) inverse_mapping_from_tables(T, Ainv_out);
    return;
  }
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

/* Dense Householder QR solve of M x = B (M n x n, B n x nrhs, both overwritten; X returned in B).
 *
 * R_pp is block-tridiagonal: outside a band of a few entries either side of the diagonal it holds EXACT
 * zeros (never written after the zero fill).  The reflection of column c only mixes rows c .. c + bl (bl = lower band
 * width) and, in those rows, columns c .. c + bl + bu (bu = upper band width: the fill of earlier reflections included);
 * every operation outside that window multiplies by, adds or subtracts an exact zero and leaves its operand as it was.  The
 * loops below are the dense algorithm with exactly those operations left out -- the same values to the last bit (only the
 * sign of a structural zero can differ), in O(n (bl + bu)^2) instead of O(n^3), which is what lets the oracle follow the
 * product to 256-segment paths (n = 1020).  mto_set_dense_qr(1) runs every loop over its full dense range again
 * (tests/test_oracle_properties.py compares the two bit for bit). */
static int g_dense_qr = 0;
void mto_set_dense_qr(int on) { g_dense_qr = on; }

static void band_widths(const double* M, int n, int* bl_out, int* bu_out) {
  int bl = 0, bu = 0;
  for (int r = 0; r < n; ++r) {
    const double* row = M + (size_t)r * n;
    for (int c = 0; c < r - bl; ++c)
      if (row[c] != 0.0) {
        bl = r - c;
        break;
      }
    for (int c = n - 1; c > r + bu; --c)
      if (row[c] != 0.0) {
        bu = c - r;
        break;
      }
  }
  *bl_out = bl;
  *bu_out = bu;
}

static int qr_solve(double* M, int n, double* B, int nrhs) {
  const mto_scratch_state mark = mto_scratch_mark();
  double* v = (double*)mto_scratch_alloc(sizeof(double) * (size_t)n, 0);
  if (!v) return -1;
  int bl = n, bu = n;
  if (!g_dense_qr) band_widths(M, n, &bl, &bu);
  for (int c = 0; c < n; ++c) {
    const int rend = (c + bl < n - 1) ? c + bl + 1 : n;                 /* rows c .. rend - 1 */
    const int jend = ((long)c + bl + bu < n - 1) ? c + bl + bu + 1 : n; /* columns c .. jend - 1 */
    double norm = 0.0;
    for (int r = c; r < rend; ++r) norm += M[(size_t)r * n + c] * M[(size_t)r * n + c];
    norm = sqrt(norm);
    if (norm == 0.0) {
      mto_scratch_release(mark);
      return -2;
    }
    const double alpha = (M[(size_t)c * n + c] > 0) ? -norm : norm;
    for (int r = c; r < rend; ++r) v[r] = M[(size_t)r * n + c];
    v[c] -= alpha;
    double vnorm2 = 0.0;
    for (int r = c; r < rend; ++r) vnorm2 += v[r] * v[r];
    if (vnorm2 > 0.0) {
      for (int j = c; j < jend; ++j) {
        double dot = 0.0;
        for (int r = c; r < rend; ++r) dot += v[r] * M[(size_t)r * n + j];
        const double f = 2.0 * dot / vnorm2;
        for (int r = c; r < rend; ++r) M[(size_t)r * n + j] -= f * v[r];
      }
      for (int j = 0; j < nrhs; ++j) {
        double dot = 0.0;
        for (int r = c; r < rend; ++r) dot += v[r] * B[(size_t)r * nrhs + j];
        const double f = 2.0 * dot / vnorm2;
        for (int r = c; r < rend; ++r) B[(size_t)r * nrhs + j] -= f * v[r];
      }
    }
  }
  for (int j = 0; j < nrhs; ++j)
    for (int r = n - 1; r >= 0; --r) {
      const int kend = ((long)r + bl + bu < n - 1) ? r + bl + bu + 1 : n;
      double s = B[(size_t)r * nrhs + j];
      for (int k = r + 1; k < kend; ++k) s -= M[(size_t)r * n + k] * B[(size_t)k * nrhs + j];
      B[(size_t)r * nrhs + j] = s / M[(size_t)r * n + r];
    }
  mto_scratch_release(mark);
  return 0;
}

/* ---- third route: the whole linear solve in 113-bit arithmetic ------------------------------------------------
 * mto_set_arithmetic(2): H(T) and A^-1(T) from the unit-time tables kept in __float128, R, the QR solve, the coefficients
 * and the cost 0.5 c^T Q c all in __float128; results rounded to double once.  This is the reference's ALGORITHM with
 * (for the purposes of a double-precision comparison) no rounding error: what both the reference-style route and the HIP
 * path approximate.  ~50x slower than route 0; for parity sweeps that ask which of the two is closer. */
static __thread struct {
  const double* coeffs;
  double cost;
  int valid;
} tl_quad_cost;

static void band_widths_q(const q_t* M, int n, int* bl_out, int* bu_out) {
  int bl = 0, bu = 0;
  for (int r = 0; r < n; ++r) {
    const q_t* row = M + (size_t)r * n;
    for (int c = 0; c < r - bl; ++c)
      if (row[c] != 0) {
        bl = r - c;
        break;
      }
    for (int c = n - 1; c > r + bu; --c)
      if (row[c] != 0) {
        bu = c - r;
        break;
      }
  }
  *bl_out = bl;
  *bu_out = bu;
}

/* (the band-limited loops of qr_solve, see there) */
static int qr_solve_q(q_t* M, int n, q_t* B, int nrhs, q_t* v) {
  int bl = n, bu = n;
  if (!g_dense_qr) band_widths_q(M, n, &bl, &bu);
  for (int c = 0; c < n; ++c) {
    const int rend = (c + bl < n - 1) ? c + bl + 1 : n;
    const int jend = ((long)c + bl + bu < n - 1) ? c + bl + bu + 1 : n;
    q_t norm2 = 0;
    for (int r = c; r < rend; ++r) norm2 += M[(size_t)r * n + c] * M[(size_t)r * n + c];
    if (norm2 == 0) return -2;
    /* sqrt by Newton from the double estimate */
    q_t norm = (q_t)sqrt((double)norm2);
    for (int it = 0; it < 3; ++it) norm = (norm + norm2 / norm) / 2;
    const q_t alpha = (M[(size_t)c * n + c] > 0) ? -norm : norm;
    for (int r = c; r < rend; ++r) v[r] = M[(size_t)r * n + c];
    v[c] -= alpha;
    q_t vnorm2 = 0;
    for (int r = c; r < rend; ++r) vnorm2 += v[r] * v[r];
    if (vnorm2 > 0) {
      for (int j = c; j < jend; ++j) {
        q_t dot = 0;
        for (int r = c; r < rend; ++r) dot += v[r] * M[(size_t)r * n + j];
        const q_t f = 2 * dot / vnorm2;
        for (int r = c; r < rend; ++r) M[(size_t)r * n + j] -= f * v[r];
      }
      for (int j = 0; j < nrhs; ++j) {
        q_t dot = 0;
        for (int r = c; r < rend; ++r) dot += v[r] * B[(size_t)r * nrhs + j];
        const q_t f = 2 * dot / vnorm2;
        for (int r = c; r < rend; ++r) B[(size_t)r * nrhs + j] -= f * v[r];
      }
    }
  }
  for (int j = 0; j < nrhs; ++j)
    for (int r = n - 1; r >= 0; --r) {
      const int kend = ((long)r + bl + bu < n - 1) ? r + bl + bu + 1 : n;
      q_t s = B[(size_t)r * nrhs + j];
      for (int k = r + 1; k < kend; ++k) s -= M[(size_t)r * n + k] * B[(size_t)k * nrhs + j];
      B[(size_t)r * nrhs + j] = s / M[(size_t)r * n + r];
    }
  return 0;
}

static int solve_linear_quad(const mto_path* path, const double* seg_times, double* coeffs_out, double* free_out) {
  pthread_once(&g_tables_once, build_unit_tables);
  const int S = path->n_seg, V = S + 1, d = path->derivative_to_optimize;
  if (S < 1 || S > MTO_MAX_SEG) return -1;
  const int n_all = HALF * V;
  const mto_scratch_state mark = mto_scratch_mark();
  int* col = (int*)mto_scratch_alloc(sizeof(int) * (size_t)n_all, 0);
  int n_fixed = 0;
  for (int i = 0; i < n_all; ++i) n_fixed += path->fixed_mask[i] ? 1 : 0;
  const int n_free = n_all - n_fixed;
  q_t* R = (q_t*)mto_scratch_alloc(sizeof(q_t) * (size_t)n_all * (size_t)n_all, 1);
  q_t* tp = (q_t*)mto_scratch_alloc(sizeof(q_t) * (size_t)S * 19, 0);
  q_t* dall = (q_t*)mto_scratch_alloc(sizeof(q_t) * (size_t)n_all * DIM, 1);
  q_t* Rpp = (q_t*)mto_scratch_alloc(sizeof(q_t) * (size_t)(n_free > 0 ? n_free : 1) * (size_t)(n_free > 0 ? n_free : 1), 0);
  q_t* rhs = (q_t*)mto_scratch_alloc(sizeof(q_t) * (size_t)(n_free > 0 ? n_free : 1) * DIM, 0);
  q_t* hv = (q_t*)mto_scratch_alloc(sizeof(q_t) * (size_t)(n_free > 0 ? n_free : 1), 0);
  if (!col || !R || !tp || !dall || !Rpp || !rhs || !hv) {
    mto_scratch_release(mark);
    return -1;
  }
  {
    int cf = 0, cp = n_fixed;
    for (int i = 0; i < n_all; ++i) col[i] = path->fixed_mask[i] ? cf++ : cp++;
  }
  for (int i = 0; i < S; ++i) {
    q_t* t = tp + (size_t)i * 19;  /* t[9 + e] = T^e */
    const q_t T = (q_t)seg_times[i];
    t[9] = 1;
    for (int e = 1; e <= 9; ++e) t[9 + e] = t[9 + e - 1] * T;
    for (int e = 1; e <= 9; ++e) t[9 - e] = 1 / t[9 + e];
    for (int r = 0; r < N; ++r) {
      const int cr = col[(i + r / HALF) * HALF + r % HALF];
      for (int c = 0; c < N; ++c) {
        const int cc = col[(i + c / HALF) * HALF + c % HALF];
        R[(size_t)cr * n_all + cc] += g_hbar_q[d][r][c] * t[9 + (r % HALF) + (c % HALF) + 1 - 2 * d];
      }
    }
  }
  for (int i = 0; i < n_all; ++i)
    if (path->fixed_mask[i])
      for (int k = 0; k < DIM; ++k) dall[(size_t)col[i] * DIM + k] = (q_t)path->fixed_values[(size_t)i * DIM + k];
  int rc = 0;
  if (n_free > 0) {
    for (int r = 0; r < n_free; ++r) {
      for (int c = 0; c < n_free; ++c) Rpp[(size_t)r * n_free + c] = R[(size_t)(n_fixed + r) * n_all + n_fixed + c];
      for (int k = 0; k < DIM; ++k) {
        q_t acc = 0;
        for (int c = 0; c < n_fixed; ++c) acc += -R[(size_t)(n_fixed + r) * n_all + c] * dall[(size_t)c * DIM + k];
        rhs[(size_t)r * DIM + k] = acc;
      }
    }
    rc = qr_solve_q(Rpp, n_free, rhs, DIM, hv);
    for (int r = 0; r < n_free; ++r)
      for (int k = 0; k < DIM; ++k) {
        dall[(size_t)(n_fixed + r) * DIM + k] = rhs[(size_t)r * DIM + k];
        if (free_out) free_out[(size_t)k * n_free + r] = (double)rhs[(size_t)r * DIM + k];
      }
  }
  /* coefficients and the cost 0.5 sum c^T Q c, Q[i][j] = B[d][i] B[d][j] 2 T^(i+j-2d+1) / (i+j-2d+1)  (linear_impl.h:606-618) */
  q_t cost = 0;
  for (int i = 0; i < S; ++i) {
    const q_t* t = tp + (size_t)i * 19;
    for (int k = 0; k < DIM; ++k) {
      q_t dseg[N], c[N];
      for (int r = 0; r < N; ++r) dseg[r] = dall[(size_t)col[(i + r / HALF) * HALF + r % HALF] * DIM + k];
      for (int r = 0; r < N; ++r) {
        q_t acc = 0;
        for (int q = 0; q < N; ++q) acc += g_abar_inv_q[r][q] * t[9 + (q % HALF) - r] * dseg[q];
        c[r] = acc;
        coeffs_out[((size_t)i * DIM + k) * N + r] = (double)acc;
      }
      q_t partial = 0;
      for (int r = d; r < N; ++r)
        for (int q = d; q < N; ++q) {
          const int e = r + q - 2 * d + 1;  /* 1 .. 19 - 2d: T^e as T^9 * T^(e - 9) where e > 9 */
          const q_t te = (e <= 9) ? t[9 + e] : t[18] * t[9 + e - 9];
          partial += c[r] * c[q] * ((q_t)mto_base_coeff(d, r) * (q_t)mto_base_coeff(d, q) * 2 * te / (q_t)e);
        }
      cost += partial;
    }
  }
  tl_quad_cost.coeffs = coeffs_out;
  tl_quad_cost.cost = (double)(cost / 2);
  tl_quad_cost.valid = 1;
  mto_scratch_release(mark);
  return rc;
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
    double Ai[N * N];
    segment_inverse_mapping(seg_times[i], Ai);
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
  if (g_arith == 2) return solve_linear_quad(path, seg_times, coeffs_out, free_out);
  tl_quad_cost.valid = 0;
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
  /* (113-bit route: the cost of the solve that has just filled `coeffs` was formed before they were rounded to double) */
  if (g_arith == 2 && tl_quad_cost.valid && tl_quad_cost.coeffs == coeffs) return tl_quad_cost.cost;
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
