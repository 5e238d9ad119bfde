/*
 * mto_poly.c -- CPU ORACLE (test infrastructure): polynomial primitives and the real-polynomial
 * root finder.  See mrs_tg_oracle.h for the rules that apply to everything under oracle/.
 *
 * Follows (relative to /root/reference/):
 *   include/eth_trajectory_generation/polynomial.h:108-163,208-237
 *   src/eth_trajectory_generation/polynomial.cpp:155-192
 *   src/eth_trajectory_generation/rpoly/rpoly_ak1.cpp:59-120 (wrapper), :148-932 (TOMS 493)
 *
 * The root finder is the three-stage Jenkins-Traub RPOLY algorithm (ACM TOMS 493).  It is restated
 * here from the published algorithm with the same stage limits and tolerances as the reference's
 * translation (5 no-shift steps, <= 20 shifts of 94 degrees, 20*shift fixed-shift steps, 20 quadratic
 * / 10 linear variable-shift steps, 10*eps degeneracy tests), organised around one state struct.
 */
#include "mrs_tg_oracle.h"

#include <float.h>
#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------- */
/* derivative table                                                                            */

double mto_base_coeff(int r, int k) {
  /* polynomial.cpp:155-170 builds row n from row n-1 by multiplying with (order-DEG+i) = (i-n+1);
   * the product telescopes to k!/(k-r)!.  Same multiplication order => same doubles (all exact). */
  if (k < r) return 0.0;
  double v = 1.0;
  for (int n = 1; n <= r; ++n) v = (double)(k - n + 1) * v;
  return v;
}

double mto_poly_eval(const double* c, int n, double t, int derivative) {
  /* polynomial.h:150-163 */
  if (derivative >= n) return 0.0;
  const int top = n - 1;
  double acc = mto_base_coeff(derivative, top) * c[top];
  for (int j = top - 1; j >= derivative; --j) {
    acc *= t;
    acc += mto_base_coeff(derivative, j) * c[j];
  }
  return acc;
}

void mto_poly_derivative(const double* c, int n, int derivative, double* out) {
  /* polynomial.h:108-119: head(n-derivative) = tail(n-derivative) .* B[derivative, derivative..] */
  for (int i = 0; i < n; ++i) out[i] = 0.0;
  if (derivative == 0) {
    for (int i = 0; i < n; ++i) out[i] = c[i];
    return;
  }
  for (int i = 0; i < n - derivative; ++i) out[i] = c[i + derivative] * mto_base_coeff(derivative, i + derivative);
}

void mto_convolve(const double* data, int nd, const double* kernel, int nk, double* out) {
  /* polynomial.cpp:176-192: out[i] = sum over kernel_idx of reverse(kernel)[kernel_idx] * data[i-nk+1+kernel_idx] */
  const int len = nd + nk - 1;
  for (int i = 0; i < len; ++i) {
    const int data_idx = i - nk + 1;
    const int lo = (-data_idx > 0) ? -data_idx : 0;
    const int hi = (nk < nd - data_idx) ? nk : nd - data_idx;
    double acc = 0.0;
    for (int ki = lo; ki < hi; ++ki) acc += kernel[nk - 1 - ki] * data[data_idx + ki];
    out[i] = acc;
  }
}

/* ------------------------------------------------------------------------------------------- */
/* Jenkins-Traub (TOMS 493)                                                                    */

#define JT_MAXDEG 100

typedef struct {
  int n;  /* degree of the current (deflated) polynomial */
  double p[JT_MAXDEG + 1], qp[JT_MAXDEG + 1];
  double k[JT_MAXDEG + 1], qk[JT_MAXDEG + 1], svk[JT_MAXDEG + 1];
  double u, v;                 /* current quadratic z^2 + u z + v              */
  double a, b;                 /* remainder of p / quadratic                   */
  double c, d, e, f, g, h;     /* scalars of the K-polynomial recurrences      */
  double a1, a3, a7;
  double szr, szi, lzr, lzi;   /* smaller / larger zero found                  */
} jt_t;

/* divide poly[0..m) (decreasing powers) by z^2+u z+v: quotient q, remainder (a, b).  rpoly_ak1.cpp:543-559 */
static void jt_quad_div(int m, double u, double v, const double* poly, double* q, double* a, double* b) {
  double bb = poly[0], aa;
  q[0] = bb;
  aa = -(bb * u) + poly[1];
  q[1] = aa;
  for (int i = 2; i < m; ++i) {
    const double t = -(aa * u + bb * v) + poly[i];
    q[i] = t;
    bb = aa;
    aa = t;
  }
  *a = aa;
  *b = bb;
}

/* zeros of a z^2 + b1 z + c, overflow-safe.  rpoly_ak1.cpp:881-932 */
static void jt_quadratic(double a, double b1, double c, double* sr, double* si, double* lr, double* li) {
  *sr = *si = *lr = *li = 0.0;
  if (a == 0) {
    if (b1 != 0) *sr = -(c / b1);
    return;
  }
  if (c == 0) {
    *lr = -(b1 / a);
    return;
  }
  const double b = b1 / 2.0;
  double d, e;
  if (fabs(b) < fabs(c)) {
    e = (c >= 0) ? a : -a;
    e = -e + b * (b / fabs(c));
    d = sqrt(fabs(e)) * sqrt(fabs(c));
  } else {
    e = -((a / b) * (c / b)) + 1.0;
    d = sqrt(fabs(e)) * fabs(b);
  }
  if (e >= 0) { /* real */
    if (b >= 0) d = -d;
    *lr = (-b + d) / a;
    if (*lr != 0) *sr = (c / (*lr)) / a;
  } else { /* complex pair */
    *lr = *sr = -(b / a);
    *si = fabs(d / a);
    *li = -(*si);
  }
}

/* scalars for the next K polynomial; returns the normalisation type (3 = quadratic ~ factor of K).
 * rpoly_ak1.cpp:561-602 */
static int jt_scalars(jt_t* s) {
  const int n = s->n;
  jt_quad_div(n, s->u, s->v, s->k, s->qk, &s->c, &s->d);
  if (fabs(s->c) <= 10.0 * DBL_EPSILON * fabs(s->k[n - 1])) {
    if (fabs(s->d) <= 10.0 * DBL_EPSILON * fabs(s->k[n - 2])) return 3;
  }
  s->h = s->v * s->b;
  if (fabs(s->d) >= fabs(s->c)) {
    s->e = s->a / s->d;
    s->f = s->c / s->d;
    s->g = s->u * s->b;
    s->a3 = s->e * (s->g + s->a) + s->h * (s->b / s->d);
    s->a1 = -s->a + s->f * s->b;
    s->a7 = s->h + (s->f + s->u) * s->a;
    return 2;
  }
  s->e = s->a / s->c;
  s->f = s->d / s->c;
  s->g = s->e * s->u;
  s->a3 = s->e * s->a + (s->g + s->h / s->c) * s->b;
  s->a1 = -(s->a * (s->d / s->c)) + s->b;
  s->a7 = s->g * s->d + s->h * s->f + s->a;
  return 1;
}

/* next K polynomial.  rpoly_ak1.cpp:604-645 */
static void jt_next_k(jt_t* s, int type) {
  const int n = s->n;
  if (type == 3) {
    s->k[0] = s->k[1] = 0.0;
    for (int i = 2; i < n; ++i) s->k[i] = s->qk[i - 2];
    return;
  }
  const double ref = (type == 1) ? s->b : s->a;
  if (fabs(s->a1) > 10.0 * DBL_EPSILON * fabs(ref)) {
    s->a7 /= s->a1;
    s->a3 /= s->a1;
    s->k[0] = s->qp[0];
    s->k[1] = -(s->a7 * s->qp[0]) + s->qp[1];
    for (int i = 2; i < n; ++i) s->k[i] = -(s->a7 * s->qp[i - 1]) + s->a3 * s->qk[i - 2] + s->qp[i];
  } else {
    s->k[0] = 0.0;
    s->k[1] = -s->a7 * s->qp[0];
    for (int i = 2; i < n; ++i) s->k[i] = -(s->a7 * s->qp[i - 1]) + s->a3 * s->qk[i - 2];
  }
}

/* new estimate (uu, vv) of the quadratic.  rpoly_ak1.cpp:647-683 */
static void jt_new_estimate(const jt_t* s, int type, double* uu, double* vv) {
  *uu = *vv = 0.0;
  if (type == 3) return;
  const int n = s->n;
  double a4, a5;
  if (type != 2) {
    a4 = s->a + s->u * s->b + s->h * s->f;
    a5 = s->c + (s->u + s->v * s->f) * s->d;
  } else {
    a4 = (s->a + s->g) * s->f + s->h;
    a5 = (s->f + s->u) * s->c + s->v * s->d;
  }
  const double b1 = -s->k[n - 1] / s->p[n];
  const double b2 = -(s->k[n - 2] + b1 * s->p[n - 1]) / s->p[n];
  const double c1 = s->v * b2 * s->a1;
  const double c2 = b1 * s->a7;
  const double c3 = b1 * b1 * s->a3;
  const double c4 = -(c2 + c3) + c1;
  const double den = -c4 + a5 + b1 * a4;
  if (den != 0.0) {
    *uu = -((s->u * (c3 + c2) + s->v * (b1 * s->a1 + b2 * s->a7)) / den) + s->u;
    *vv = s->v * (1.0 + c4 / den);
  }
}

/* variable-shift quadratic iteration; returns number of zeros found (0 or 2).  rpoly_ak1.cpp:685-783 */
static int jt_quad_iteration(jt_t* s, double uu, double vv) {
  const int n = s->n, nn = n + 1;
  int j = 0, tried = 0, type;
  double relstp = 0.0, omp = 0.0, mp, ui, vi;
  s->u = uu;
  s->v = vv;
  for (;;) {
    jt_quadratic(1.0, s->u, s->v, &s->szr, &s->szi, &s->lzr, &s->lzi);
    /* bail out unless the two zeros are (nearly) equimodular */
    if (fabs(fabs(s->szr) - fabs(s->lzr)) > 0.01 * fabs(s->lzr)) return 0;
    jt_quad_div(nn, s->u, s->v, s->p, s->qp, &s->a, &s->b);
    mp = fabs(-(s->szr * s->b) + s->a) + fabs(s->szi * s->b);
    /* rigorous rounding-error bound for p at the zero */
    const double zm = sqrt(fabs(s->v));
    double ee = 2.0 * fabs(s->qp[0]);
    const double t = -(s->szr * s->b);
    for (int i = 1; i < n; ++i) ee = ee * zm + fabs(s->qp[i]);
    ee = ee * zm + fabs(s->a + t);
    ee = (9.0 * ee + 2.0 * fabs(t) - 7.0 * (fabs(s->a + t) + zm * fabs(s->b))) * DBL_EPSILON;
    if (mp <= 20.0 * ee) return 2;
    if (++j > 20) return 0;
    if (j >= 2 && relstp <= 0.01 && mp >= omp && !tried) {
      /* a cluster stalls convergence: five fixed-shift steps near it */
      relstp = (relstp < DBL_EPSILON) ? sqrt(DBL_EPSILON) : sqrt(relstp);
      s->u -= s->u * relstp;
      s->v += s->v * relstp;
      jt_quad_div(nn, s->u, s->v, s->p, s->qp, &s->a, &s->b);
      for (int i = 0; i < 5; ++i) {
        type = jt_scalars(s);
        jt_next_k(s, type);
      }
      tried = 1;
      j = 0;
    }
    omp = mp;
    type = jt_scalars(s);
    jt_next_k(s, type);
    type = jt_scalars(s);
    jt_new_estimate(s, type, &ui, &vi);
    if (vi == 0) return 0; /* not converging */
    relstp = fabs((-s->v + vi) / vi);
    s->u = ui;
    s->v = vi;
  }
}

/* variable-shift real iteration; returns zeros found (0/1); *near_double set when a cluster near the
 * real axis is met (then *sss holds the iterate).  rpoly_ak1.cpp:785-879 */
static int jt_real_iteration(jt_t* s, double* sss, int* near_double) {
  const int n = s->n, nn = n + 1, nm1 = n - 1;
  int j = 0;
  double x = *sss, t = 0.0, omp = 0.0;
  *near_double = 0;
  for (;;) {
    double pv = s->p[0];
    s->qp[0] = pv;
    for (int i = 1; i < nn; ++i) s->qp[i] = pv = pv * x + s->p[i];
    const double mp = fabs(pv);
    const double ms = fabs(x);
    double ee = 0.5 * fabs(s->qp[0]);
    for (int i = 1; i < nn; ++i) ee = ee * ms + fabs(s->qp[i]);
    if (mp <= 20.0 * DBL_EPSILON * (2.0 * ee - mp)) {
      s->szr = x;
      s->szi = 0.0;
      return 1;
    }
    if (++j > 10) return 0;
    if (j >= 2 && fabs(t) <= 0.001 * fabs(-t + x) && mp > omp) {
      *near_double = 1;
      *sss = x;
      return 0;
    }
    omp = mp;
    double kv = s->k[0];
    s->qk[0] = kv;
    for (int i = 1; i < n; ++i) s->qk[i] = kv = kv * x + s->k[i];
    if (fabs(kv) > fabs(s->k[nm1]) * 10.0 * DBL_EPSILON) {
      t = -(pv / kv);
      s->k[0] = s->qp[0];
      for (int i = 1; i < n; ++i) s->k[i] = t * s->qk[i - 1] + s->qp[i];
    } else {
      s->k[0] = 0.0;
      for (int i = 1; i < n; ++i) s->k[i] = s->qk[i - 1];
    }
    kv = s->k[0];
    for (int i = 1; i < n; ++i) kv = kv * x + s->k[i];
    t = (fabs(kv) > fabs(s->k[nm1]) * 10.0 * DBL_EPSILON) ? -(pv / kv) : 0.0;
    x += t;
  }
}

/* fixed-shift stage with up to l2 steps; launches a variable-shift iteration when one of the two
 * convergence sequences passes.  Returns the number of zeros found.  rpoly_ak1.cpp:389-541 */
static int jt_fixed_shift(jt_t* s, int l2, double sr, double bnd) {
  const int n = s->n, nn = n + 1;
  double betav = 0.25, betas = 0.25;
  double oss = sr, ovv = bnd, otv = 0.0, ots = 0.0;
  double ui, vi, sval = 0.0;
  s->u = -(2.0 * sr);
  s->v = bnd;
  const double u0 = s->u, v0 = s->v; /* the fixed quadratic of this stage */
  jt_quad_div(nn, s->u, s->v, s->p, s->qp, &s->a, &s->b);
  int type = jt_scalars(s);
  for (int j = 0; j < l2; ++j) {
    jt_next_k(s, type);
    type = jt_scalars(s);
    jt_new_estimate(s, type, &ui, &vi);
    const double vv = vi;
    const double ss = (s->k[n - 1] != 0.0) ? -(s->p[n] / s->k[n - 1]) : 0.0;
    double ts = 1.0, tv = 1.0;
    if (j != 0 && type != 3) {
      if (vv != 0.0) tv = fabs((vv - ovv) / vv);
      if (ss != 0.0) ts = fabs((ss - oss) / ss);
      const double tvv = (tv < otv) ? tv * otv : 1.0;
      const double tss = (ts < ots) ? ts * ots : 1.0;
      const int vpass = tvv < betav;
      const int spass = tss < betas;
      if (spass || vpass) {
        memcpy(s->svk, s->k, sizeof(double) * (size_t)n);
        sval = ss;
        int stry = 0, vtry = 0, first = 1;
        for (;;) {
          int run_real = 1;
          int prefer_real = 0;
          if (first) {
            first = 0;
            prefer_real = spass && (!vpass || tss < tvv);
          }
          if (!prefer_real) {
            if (jt_quad_iteration(s, ui, vi) > 0) return 2;
            /* quadratic iteration failed: tighten its criterion */
            vtry = 1;
            betav *= 0.25;
            if (stry || !spass) {
              run_real = 0;
            } else {
              memcpy(s->k, s->svk, sizeof(double) * (size_t)n);
            }
          }
          if (run_real) {
            int near_double = 0;
            if (jt_real_iteration(s, &sval, &near_double) > 0) return 1;
            stry = 1;
            betas *= 0.25;
            if (near_double) {
              /* almost-double real zero: try the quadratic iteration on (z - s)^2 */
              ui = -(sval + sval);
              vi = sval * sval;
              if (vpass && !vtry) continue;
              break;
            }
          }
          memcpy(s->k, s->svk, sizeof(double) * (size_t)n);
          if (!(vpass && !vtry)) break;
        }
        /* recompute qp and the scalars to continue the fixed-shift stage */
        s->u = u0;
        s->v = v0;
        jt_quad_div(nn, s->u, s->v, s->p, s->qp, &s->a, &s->b);
        type = jt_scalars(s);
      }
    }
    ovv = vv;
    oss = ss;
    otv = tv;
    ots = ts;
  }
  return 0;
}

/* driver: op[0..degree] decreasing powers; returns number of zeros found.  rpoly_ak1.cpp:148-387 */
static int jt_rpoly(const double* op, int degree, double* zr, double* zi) {
  if (degree > JT_MAXDEG) return -1;
  if (op[0] == 0) return 0; /* leading coefficient zero: reference reports degree 0 */
  static const double kDeg2Rad = 3.14159265358979323846 / 180.0;
  const double lb2 = log(2.0);
  const double lo = FLT_MIN / DBL_EPSILON;
  const double cosr = cos(94.0 * kDeg2Rad);
  const double sinr = sin(94.0 * kDeg2Rad);
  jt_t st;
  double pt[JT_MAXDEG + 1], ksave[JT_MAXDEG + 1];
  int n = degree, found_total = degree;
  double xx = sqrt(0.5), yy = -xx;
  int j = 0;
  while (op[n] == 0) { /* zeros at the origin */
    zr[j] = zi[j] = 0.0;
    --n;
    ++j;
  }
  int nn = n + 1;
  for (int i = 0; i < nn; ++i) st.p[i] = op[i];

  while (n >= 1) {
    if (n <= 2) {
      if (n < 2) {
        zr[degree - 1] = -(st.p[1] / st.p[0]);
        zi[degree - 1] = 0.0;
      } else {
        jt_quadratic(st.p[0], st.p[1], st.p[2], &zr[degree - 2], &zi[degree - 2], &zr[degree - 1], &zi[degree - 1]);
      }
      break;
    }
    /* scale to avoid overflow / undetected underflow */
    double mmax = 0.0, mmin = FLT_MAX;
    for (int i = 0; i < nn; ++i) {
      const double x = fabs(st.p[i]);
      if (x > mmax) mmax = x;
      if (x != 0 && x < mmin) mmin = x;
    }
    double sc = lo / mmin;
    if ((sc <= 1.0 && mmax >= 10) || (sc > 1.0 && FLT_MAX / sc >= mmax)) {
      if (sc == 0) sc = FLT_MIN;
      const int l = (int)(log(sc) / lb2 + 0.5);
      const double factor = pow(2.0, l);
      if (factor != 1.0)
        for (int i = 0; i < nn; ++i) st.p[i] *= factor;
    }
    /* lower bound on the moduli of the zeros */
    for (int i = 0; i < nn; ++i) pt[i] = fabs(st.p[i]);
    pt[n] = -pt[n];
    const int nm1 = n - 1;
    double x = exp((log(-pt[n]) - log(pt[0])) / (double)n);
    if (pt[nm1] != 0) {
      const double xm0 = -pt[n] / pt[nm1];
      if (xm0 < x) x = xm0;
    }
    double xm = x, ff;
    do {
      x = xm;
      xm = 0.1 * x;
      ff = pt[0];
      for (int i = 1; i < nn; ++i) ff = ff * xm + pt[i];
    } while (ff > 0);
    double dx = x;
    while (fabs(dx / x) > 0.005) {
      double df = pt[0];
      ff = pt[0];
      for (int i = 1; i < n; ++i) {
        ff = x * ff + pt[i];
        df = x * df + ff;
      }
      ff = x * ff + pt[n];
      dx = ff / df;
      x -= dx;
    }
    const double bnd = x;
    /* K0 = p'/n, then five no-shift steps */
    st.n = n;
    for (int i = 1; i < n; ++i) st.k[i] = (double)(n - i) * st.p[i] / (double)n;
    st.k[0] = st.p[0];
    const double aa = st.p[n], bb = st.p[nm1];
    int zerok = (st.k[nm1] == 0);
    for (int it = 0; it < 5; ++it) {
      const double cc = st.k[nm1];
      if (zerok) {
        for (int i = 0; i < nm1; ++i) st.k[nm1 - i] = st.k[nm1 - i - 1];
        st.k[0] = 0;
        zerok = (st.k[nm1] == 0);
      } else {
        const double t = -aa / cc;
        for (int i = 0; i < nm1; ++i) {
          const int jj = nm1 - i;
          st.k[jj] = t * st.k[jj - 1] + st.p[jj];
        }
        st.k[0] = st.p[0];
        zerok = (fabs(st.k[nm1]) <= fabs(bb) * DBL_EPSILON * 10.0);
      }
    }
    memcpy(ksave, st.k, sizeof(double) * (size_t)n);
    int shift, nz = 0;
    for (shift = 1; shift <= 20; ++shift) {
      const double xxx = -(sinr * yy) + cosr * xx;
      yy = sinr * xx + cosr * yy;
      xx = xxx;
      const double sr = bnd * xx;
      nz = jt_fixed_shift(&st, 20 * shift, sr, bnd);
      if (nz != 0) {
        const int at = degree - n;
        zr[at] = st.szr;
        zi[at] = st.szi;
        nn -= nz;
        n = nn - 1;
        for (int i = 0; i < nn; ++i) st.p[i] = st.qp[i];
        if (nz != 1) {
          zr[at + 1] = st.lzr;
          zi[at + 1] = st.lzi;
        }
        break;
      }
      memcpy(st.k, ksave, sizeof(double) * (size_t)n);
    }
    if (shift > 20) { /* no convergence with 20 shifts */
      found_total = degree - n;
      break;
    }
  }
  return found_total;
}

int mto_find_roots_jenkins_traub(const double* ci, int n_coeffs, double* re, double* im) {
  /* rpoly_ak1.cpp:59-120: strip trailing (highest-power) zeros, reverse, call rpoly */
  int last = -1;
  for (int i = n_coeffs - 1; i >= 0; --i) {
    if (fabs(ci[i]) >= DBL_MIN) {
      last = i;
      break;
    }
  }
  if (last == -1) return 0; /* all-zero polynomial: no roots */
  const int nc = last + 1;
  if (nc < 2) return 0; /* constant */
  double dec[JT_MAXDEG + 1];
  for (int i = 0; i < nc; ++i) dec[i] = ci[last - i];
  const int found = jt_rpoly(dec, nc - 1, re, im);
  return (found > 0) ? found : -1;
}
