// ref_harness.cpp -- C entry points over the EIGEN-ONLY part of the reference itself, for pinning the oracle.
//
// TEST INFRASTRUCTURE (see mrs_tg_oracle.h).  This file is OURS: it contains no reference code, it only CALLS the reference's
// public functions.  It is compiled by oracle/build_ref.sh together with the reference's own sources where they lie under
// /root/reference (never copied), and only when Eigen3 headers exist in the image -- which they do not in the image this was
// written in, so the file has never been compiled (REF_BUILD.md).  Outputs go to oracle/_ref/ (git-ignored).
//
// What each entry point pins (tests/test_oracle_vs_reference_build.py):
//   ref_find_roots            rpoly/rpoly_ak1.cpp:findRootsJenkinsTraub            -> oracle/mto_poly.c mto_find_roots_jenkins_traub
//   ref_min_max_candidates    polynomial.cpp:36-85 (incl. the |imag| <= eps filter, quirk B7) -> mto_poly.c candidate selection
//   ref_base_coeffs           polynomial.h:208-237 baseCoeffsWithTime              -> mto_poly.c base table / mto_linear.c A(T)
//   ref_convolve              polynomial.cpp convolve                                -> mto_poly.c products of derivative polynomials
//   ref_scale_in_time         polynomial.cpp scalePolynomialInTime                   -> mto_nonlinear.c coefficient scaling c_k s^-k
#include <eth_trajectory_generation/polynomial.h>
#include <eth_trajectory_generation/rpoly/rpoly_ak1.h>

#include <vector>

namespace etg = eth_trajectory_generation;

extern "C" {

// roots of sum_k c[k] t^k; returns the number of roots written (<= capacity), -1 when the reference reports failure
int ref_find_roots(const double* coeffs_increasing, int n, double* re, double* im, int capacity) {
  Eigen::VectorXd c = Eigen::Map<const Eigen::VectorXd>(coeffs_increasing, n);
  Eigen::VectorXcd roots;
  if (!etg::findRootsJenkinsTraub(c, &roots)) return -1;
  int m = 0;
  for (int i = 0; i < roots.size() && m < capacity; ++i, ++m) {
    re[m] = roots[i].real();
    im[m] = roots[i].imag();
  }
  return m;
}

// candidates for the extrema of derivative `derivative` of the polynomial on [t_start, t_end] (end points + real roots of the
// next derivative, as the reference selects them); returns the number written, -1 on failure
int ref_min_max_candidates(const double* coeffs_increasing, int n, double t_start, double t_end, int derivative, double* out,
                           int capacity) {
  etg::Polynomial p(Eigen::VectorXd(Eigen::Map<const Eigen::VectorXd>(coeffs_increasing, n)));
  std::vector<double> cand;
  if (!p.computeMinMaxCandidates(t_start, t_end, derivative, &cand)) return -1;
  int m = 0;
  for (size_t i = 0; i < cand.size() && m < capacity; ++i, ++m) out[m] = cand[i];
  return m;
}

// minimum and maximum (time, value) of derivative `derivative` on [t_start, t_end]; returns 0 on success
int ref_min_max(const double* coeffs_increasing, int n, double t_start, double t_end, int derivative, double* t_min, double* v_min,
                double* t_max, double* v_max) {
  etg::Polynomial p(Eigen::VectorXd(Eigen::Map<const Eigen::VectorXd>(coeffs_increasing, n)));
  std::pair<double, double> mn, mx;
  if (!p.computeMinMax(t_start, t_end, derivative, &mn, &mx)) return -1;
  *t_min = mn.first;
  *v_min = mn.second;
  *t_max = mx.first;
  *v_max = mx.second;
  return 0;
}

void ref_base_coeffs(int n, int derivative, double t, double* out) {
  const Eigen::VectorXd c = etg::Polynomial::baseCoeffsWithTime(n, derivative, t);
  for (int i = 0; i < n; ++i) out[i] = c[i];
}

// out has na + nb - 1 entries
void ref_convolve(const double* a, int na, const double* b, int nb, double* out) {
  const Eigen::VectorXd r = etg::Polynomial::convolve(Eigen::VectorXd(Eigen::Map<const Eigen::VectorXd>(a, na)),
                                                      Eigen::VectorXd(Eigen::Map<const Eigen::VectorXd>(b, nb)));
  for (int i = 0; i < r.size(); ++i) out[i] = r[i];
}

void ref_scale_in_time(const double* coeffs_increasing, int n, double scaling_factor, double* out) {
  etg::Polynomial p(Eigen::VectorXd(Eigen::Map<const Eigen::VectorXd>(coeffs_increasing, n)));
  p.scalePolynomialInTime(scaling_factor);
  const Eigen::VectorXd c = p.getCoefficients(0);
  for (int i = 0; i < n; ++i) out[i] = c[i];
}

}  // extern "C"
