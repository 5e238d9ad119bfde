// rsq_accuracy.hip -- relative error of v_rsq_f64 and of the refinement schemes used for the Cholesky pivots,
// against long double on the host.  Not part of the library.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const double* x, double* raw, double* n1, double* n2, double* gs_sqrt, double* gs_inv, double* cubic, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double y = __builtin_amdgcn_rsq(v);
  raw[i] = y;
  double y1 = y * fma(-0.5 * v * y, y, 1.5);
  n1[i] = y1;
  n2[i] = y1 * fma(-0.5 * v * y1, y1, 1.5);
  // coupled (Goldschmidt) step: sqrt and rsqrt together
  const double g = v * y, h = 0.5 * y;
  const double r = fma(-g, h, 0.5);
  gs_sqrt[i] = fma(g, r, g);
  gs_inv[i] = 2.0 * fma(h, r, h);
  // one third-order step: y (1 + e/2 + 3 e^2 / 8), e = 1 - v y^2 (four dependent operations after v_rsq_f64)
  const double e = fma(-(v * y), y, 1.0);
  cubic[i] = fma(y * e, fma(e, 0.375, 0.5), y);
}

int main() {
  const int n = 1 << 20;
  std::vector<double> x(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double m = 1.0 + (s >> 11) * (1.0 / 9007199254740992.0);
    x[i] = std::ldexp(m, (int)(s % 120) - 60);
  }
  double *dx, *d[6];
  hipMalloc(&dx, n * 8);
  for (auto& p : d) hipMalloc(&p, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d[0], d[1], d[2], d[3], d[4], d[5], n);
  const char* names[6] = {"v_rsq_f64 raw", "1 Newton step", "2 Newton steps", "coupled step: sqrt", "coupled step: 1/sqrt",
                          "1 third-order step"};
  std::vector<double> h(n);
  for (int kx = 0; kx < 6; ++kx) {
    hipMemcpy(h.data(), d[kx], n * 8, hipMemcpyDeviceToHost);
    long double worst = 0;
    for (int i = 0; i < n; ++i) {
      const long double ref = (kx == 3) ? sqrtl((long double)x[i]) : 1.0L / sqrtl((long double)x[i]);
      const long double e = fabsl(((long double)h[i] - ref) / ref);
      if (e > worst) worst = e;
    }
    printf("%-22s max relative error %.3Le  (%.2Lf ulp of 2^-53)\n", names[kx], worst, worst / 1.1102230246251565e-16L);
  }
  return 0;
}
