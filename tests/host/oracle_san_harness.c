/* oracle_san_harness.c -- CPU test program (test infrastructure): the C oracle under AddressSanitizer +
 * UndefinedBehaviorSanitizer or ThreadSanitizer.  Every numeric result the parity tests rest on comes out of
 * oracle/libmrs_tg_oracle.so; this program runs the same entry points from sanitised objects:
 *   - mto_solve_batch in every time-allocation mode (-1 linear, 2 Mellinger, 0 / 1 gradient-free, 3 / 4 with free
 *     constraints), uniform and ragged batches, with sampling -- on ONE thread and through the persistent pthread pool
 *     (8 workers, per-thread scratch stacks of mto_scratch.c), twice through the pool: the three results must be equal
 *     bit for bit (a race or a stale scratch block would show as a difference even where the sanitizer is silent);
 *   - the three arithmetic routes of the linear solve (reference, exact tables, 113 bit) and the dense / band-limited QR;
 *   - mto_optimize_path (policy loop), Jenkins-Traub on random polynomials.
 * It prints one checksum line; tests/test_host_sanitizers.py compares it with the line of the unsanitised build. */
#include <inttypes.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/mrs_tg_oracle.h"

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static double uniform(double lo, double hi) { /* splitmix64 */
  uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return lo + (hi - lo) * ((double)(z >> 11) * (1.0 / 9007199254740992.0));
}

static uint64_t fnv(uint64_t h, const void* data, size_t bytes) {
  const unsigned char* p = (const unsigned char*)data;
  for (size_t i = 0; i < bytes; ++i) {
    h ^= p[i];
    h *= 0x100000001B3ull;
  }
  return h;
}

typedef struct {
  int P, nS;
  int32_t* so;
  double *wp, *vals, *lim;
  uint8_t* mask;
} batch;

static batch make_batch(int P, int seg_lo, int seg_hi, int deriv, int with_stops) {
  batch b;
  b.P = P;
  b.so = (int32_t*)calloc((size_t)P + 1, sizeof(int32_t));
  for (int p = 0; p < P; ++p) b.so[p + 1] = b.so[p] + seg_lo + (int)uniform(0, seg_hi - seg_lo + 0.999);
  b.nS = b.so[P];
  const int nV = b.nS + P;
  b.wp = (double*)calloc((size_t)nV * 4, sizeof(double));
  b.vals = (double*)calloc((size_t)nV * 20, sizeof(double));
  b.mask = (uint8_t*)calloc((size_t)nV * 5, 1);
  b.lim = (double*)calloc((size_t)P * 9, sizeof(double));
  for (int p = 0; p < P; ++p) {
    const int v0 = b.so[p] + p, V = b.so[p + 1] - b.so[p] + 1;
    double last = 0;
    for (int i = 0; i < V; ++i) {
      double* w = b.wp + (size_t)(v0 + i) * 4;
      w[0] = uniform(-10, 10), w[1] = uniform(-10, 10), w[2] = uniform(1, 10);
      w[3] = mto_unwrap_heading(uniform(-M_PI, M_PI), i ? last : 0.0);
      last = w[3];
      uint8_t* m = b.mask + (size_t)(v0 + i) * 5;
      m[0] = 1;
      memcpy(b.vals + (size_t)(v0 + i) * 20, w, sizeof(double) * 4);
      if (i == 0 || i == V - 1) {
        for (int k = 1; k <= deriv; ++k) m[k] = 1;
        if (i == 0 && with_stops && p % 3 == 1) { /* a moving start */
          m[1] = m[2] = m[3] = 1;
          for (int k = 0; k < 3; ++k) b.vals[(size_t)(v0 + i) * 20 + 4 + k] = uniform(-1, 1);
        }
      } else if (with_stops && (i % 4 == 2)) {
        m[1] = m[2] = m[3] = 1;
      }
    }
    const double l[9] = {2, 2, 1, 2, 2, 2, 20, 20, 20};
    memcpy(b.lim + (size_t)p * 9, l, sizeof(l));
  }
  return b;
}

static void free_batch(batch* b) {
  free(b->so);
  free(b->wp);
  free(b->vals);
  free(b->mask);
  free(b->lim);
}

static int g_fail = 0;

static uint64_t solve_three_ways(const batch* b, int deriv, int mode, uint64_t h) {
  mto_options o;
  memset(&o, 0, sizeof(o));
  o.derivative_to_optimize = deriv;
  o.time_alloc_method = mode;
  o.estimate_times = 1;
  o.nlopt.max_iterations = mode == 2 ? 10 : 40;
  o.nlopt.f_rel = 0.05, o.nlopt.f_abs = -1, o.nlopt.x_rel = 0.1, o.nlopt.x_abs = -1;
  o.sampling_dt = 0.2;
  o.time_penalty = 100.0, o.use_soft_constraints = 1, o.soft_constraint_weight = 1.5, o.initial_stepsize_rel = 0.1;
  const int cap = 256;
  const size_t nc = (size_t)b->nS * 40, ns = (size_t)b->P * cap * 4;
  double* res[3][4];
  int32_t* ires[3][2];
  const int threads[3] = {1, 8, 8};
  for (int r = 0; r < 3; ++r) {
    res[r][0] = (double*)calloc((size_t)b->nS, sizeof(double));
    res[r][1] = (double*)calloc(nc, sizeof(double));
    res[r][2] = (double*)calloc((size_t)b->P, sizeof(double));
    res[r][3] = (double*)calloc(ns, sizeof(double));
    ires[r][0] = (int32_t*)calloc((size_t)b->P, sizeof(int32_t));
    ires[r][1] = (int32_t*)calloc((size_t)b->P, sizeof(int32_t));
    mto_solve_batch(b->P, b->so, b->wp, b->mask, b->vals, b->lim, &o, res[r][0], res[r][1], ires[r][0], res[r][2], ires[r][1],
                    res[r][3], cap, threads[r]);
  }
  for (int r = 1; r < 3; ++r) {
    int same = memcmp(res[0][0], res[r][0], sizeof(double) * (size_t)b->nS) == 0 && memcmp(res[0][1], res[r][1], sizeof(double) * nc) == 0 &&
               memcmp(res[0][2], res[r][2], sizeof(double) * (size_t)b->P) == 0 && memcmp(res[0][3], res[r][3], sizeof(double) * ns) == 0 &&
               memcmp(ires[0][0], ires[r][0], sizeof(int32_t) * (size_t)b->P) == 0 && memcmp(ires[0][1], ires[r][1], sizeof(int32_t) * (size_t)b->P) == 0;
    if (!same) {
      fprintf(stderr, "MISMATCH: mode %d deriv %d: the pool's result (run %d) differs from the single thread's\n", mode, deriv, r);
      ++g_fail;
    }
  }
  h = fnv(h, res[0][0], sizeof(double) * (size_t)b->nS);
  h = fnv(h, res[0][1], sizeof(double) * nc);
  h = fnv(h, ires[0][0], sizeof(int32_t) * (size_t)b->P);
  h = fnv(h, ires[0][1], sizeof(int32_t) * (size_t)b->P);
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 4; ++k) free(res[r][k]);
    free(ires[r][0]);
    free(ires[r][1]);
  }
  return h;
}

int main(void) {
  uint64_t h = 0xCBF29CE484222325ull;
  /* Mellinger + linear: uniform, ragged, constrained slots, every objective order */
  for (int deriv = 2; deriv <= 4; ++deriv) {
    batch u = make_batch(48, 10, 10, deriv, 0), r = make_batch(32, 3, 30, deriv, 1);
    h = solve_three_ways(&u, deriv, -1, h);
    h = solve_three_ways(&u, deriv, 2, h);
    h = solve_three_ways(&r, deriv, -1, h);
    h = solve_three_ways(&r, deriv, 2, h);
    free_batch(&u);
    free_batch(&r);
  }
  /* gradient-free modes */
  for (int mode = 0; mode <= 4; ++mode) {
    if (mode == 2) continue;
    batch s = make_batch(24, 2, 6, 4, mode >= 3);
    h = solve_three_ways(&s, 4, mode, h);
    free_batch(&s);
  }
  /* arithmetic routes and the two QR loops; a path of the largest accepted length */
  {
    batch s = make_batch(6, 5, 40, 4, 1), big = make_batch(1, MTO_MAX_SEG, MTO_MAX_SEG, 2, 0);
    for (int arith = 0; arith <= 2; ++arith)
      for (int dense = 0; dense <= 1; ++dense) {
        mto_set_arithmetic(arith);
        mto_set_dense_qr(dense);
        h = solve_three_ways(&s, 4, -1, h);
      }
    mto_set_arithmetic(0);
    mto_set_dense_qr(0);
    h = solve_three_ways(&big, 2, -1, h);
    free_batch(&s);
    free_batch(&big);
  }
  /* the policy loop */
  {
    mto_policy_params prm;
    mto_default_policy_params(&prm);
    mto_options o;
    memset(&o, 0, sizeof(o));
    o.derivative_to_optimize = 2, o.time_alloc_method = 2, o.estimate_times = 1;
    o.nlopt.max_iterations = 10, o.nlopt.f_rel = 0.05, o.nlopt.f_abs = -1, o.nlopt.x_rel = 0.1, o.nlopt.x_abs = -1;
    o.sampling_dt = 0.2, o.initial_stepsize_rel = 0.1;
    const double lim[9] = {2, 2, 1, 2, 2, 2, 20, 20, 20};
    double* smp = (double*)calloc(4096 * 4, sizeof(double));
    for (int q = 0; q < 12; ++q) {
      const int n = 3 + q % 6;
      double wp[16 * 4];
      uint8_t stop[16] = {0};
      double x = 0, y = 0, bearing = uniform(-M_PI, M_PI);
      for (int i = 0; i < n; ++i) {
        wp[i * 4] = x, wp[i * 4 + 1] = y, wp[i * 4 + 2] = 5 + uniform(-0.1, 0.1), wp[i * 4 + 3] = bearing;
        bearing += uniform(-0.4, 0.4);
        const double d = uniform(0.5, 2.0);
        x += d * cos(bearing), y += d * sin(bearing);
        stop[i] = (q % 4 == 1 && i == n / 2);
      }
      int ns = 0, nw = 0, it = 0;
      double md = 0;
      prm.fallback_sampling = (q % 5 == 4);
      const int ok = mto_optimize_path(wp, stop, n, NULL, lim, q % 3 == 0, &o, &prm, smp, 4096, &ns, &md, &nw, &it);
      h = fnv(h, &ok, sizeof(ok));
      h = fnv(h, &ns, sizeof(ns));
      h = fnv(h, &nw, sizeof(nw));
      h = fnv(h, smp, sizeof(double) * 4 * (size_t)(ns > 0 ? ns : 0));
    }
    free(smp);
  }
  /* Jenkins-Traub */
  for (int q = 0; q < 200; ++q) {
    const int n = 3 + q % 14;
    double c[16], re[16], im[16];
    for (int i = 0; i < n; ++i) c[i] = uniform(-3, 3);
    if (fabs(c[n - 1]) < 1e-3) c[n - 1] = 1.0;
    const int nr = mto_find_roots_jenkins_traub(c, n, re, im);
    h = fnv(h, &nr, sizeof(nr));
    if (nr > 0) h = fnv(h, re, sizeof(double) * (size_t)nr);
  }
  if (g_fail) {
    fprintf(stderr, "%d mismatches\n", g_fail);
    return 1;
  }
  printf("OK checksum %016" PRIx64 "\n", h);
  return 0;
}
