// pipeline_phases.hip -- where the time of solve_rows_pipeline_kernel goes (shader-clock stamps of the middle workgroup,
// MRS_TG_PHASE_CLOCKS), beside the separate launches it replaces.  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mrs_uav_trajectory_generation_amd/csrc -I include \
//         scripts/pipeline_phases.hip -o scripts/pipeline_phases.bin && scripts/pipeline_phases.bin [P] [S]
#define MRS_TG_PHASE_CLOCKS 1
#include "../mrs_uav_trajectory_generation_amd/csrc/mrs_tg_kernels.hip"
#include "../mrs_uav_trajectory_generation_amd/csrc/mrs_tg_tile.hip"
#include "../mrs_uav_trajectory_generation_amd/csrc/mrs_tg_rows.hip"
#include "../mrs_uav_trajectory_generation_amd/csrc/mrs_tg_quad.hip"
#include "../mrs_uav_trajectory_generation_amd/csrc/mrs_tg_general.hip"
#include "../mrs_uav_trajectory_generation_amd/csrc/mrs_tg_nonlinear.hip"
#include "../mrs_uav_trajectory_generation_amd/csrc/mrs_tg_wave.hip"
#include "../mrs_uav_trajectory_generation_amd/csrc/mrs_tg_dfo.hip"
#include "../mrs_uav_trajectory_generation_amd/csrc/mrs_tg_pool.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

using namespace mrs_tg;

template <typename T>
static T* to_dev(const std::vector<T>& h) {
  T* d;
  CK(hipMalloc(&d, h.size() * sizeof(T)));
  CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 1024, S = argc > 2 ? atoi(argv[2]) : 10, d = 4, cap = 512;
  const int V = S + 1;
  std::vector<int32_t> so(P + 1), order(P), slot(S + 1);
  for (int p = 0; p <= P; ++p) so[p] = p * S;
  for (int p = 0; p < P; ++p) order[p] = p;
  for (int j = 0; j <= S; ++j) slot[j] = j * P;
  std::vector<uint8_t> mask((size_t)P * V * 5, 0);
  std::vector<double> vals((size_t)P * V * 5 * 4, 0.0), T((size_t)P * S);
  unsigned rng = 12345u;
  auto uni = [&] { rng = rng * 1664525u + 1013904223u; return (rng >> 8) * (1.0 / 16777216.0); };
  for (int p = 0; p < P; ++p)
    for (int v = 0; v < V; ++v) {
      const size_t u = ((size_t)p * V + v) * 5;
      const bool end = (v == 0 || v == S);
      for (int k = 0; k < 5; ++k) mask[u + k] = (k == 0 || end) ? 1 : 0;
      for (int dim = 0; dim < 4; ++dim) vals[u * 4 + dim] = uni() * 20.0 - 10.0;
    }
  for (int p = 0; p < P; ++p)
    for (int i = 0; i < S; ++i) {
      double d2 = 0;
      for (int dim = 0; dim < 3; ++dim) {
        const double a = vals[(((size_t)p * V + i) * 5) * 4 + dim], bb = vals[(((size_t)p * V + i + 1) * 5) * 4 + dim];
        d2 += (a - bb) * (a - bb);
      }
      T[(size_t)p * S + i] = fmax(0.5, sqrt(d2) / 2.0);
    }
  BatchView b{P, P * S, S, S, to_dev(so), to_dev(order), to_dev(slot)};
  uint8_t* dmask = to_dev(mask);
  double *dvals = to_dev(vals), *dT = to_dev(T), *coeffs, *cost, *samples, *Tw, *maxima, *sum_t0;
  int32_t *status, *opt_status, *ns;
  CK(hipMalloc(&coeffs, (size_t)P * S * 40 * 8));
  CK(hipMalloc(&cost, (size_t)P * 8));
  CK(hipMalloc(&sum_t0, (size_t)P * 8));
  CK(hipMalloc(&status, (size_t)P * 4));
  CK(hipMalloc(&opt_status, (size_t)P * 4));
  CK(hipMalloc(&ns, (size_t)P * 4));
  CK(hipMalloc(&samples, (size_t)P * cap * 4 * 8));
  CK(hipMalloc(&Tw, (size_t)P * S * 8));
  CK(hipMalloc(&maxima, (size_t)P * S * 9 * 8));
  std::vector<double> lim((size_t)P * 9), st0((size_t)P, 1e9);
  const double l9[9] = {2, 2, 1, 2, 2, 2, 20, 20, 20};
  for (size_t i = 0; i < lim.size(); ++i) lim[i] = l9[i % 9];
  double* dlim = to_dev(lim);
  CK(hipMemcpy(sum_t0, st0.data(), (size_t)P * 8, hipMemcpyHostToDevice));
  std::vector<int32_t> os((size_t)P, 3);
  CK(hipMemcpy(opt_status, os.data(), (size_t)P * 4, hipMemcpyHostToDevice));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  RowsTail tail;
  tail.limits = dlim;
  tail.opt_status = opt_status;
  tail.sum_t0 = sum_t0;
  tail.seg_times_out = Tw;
  tail.sampling_dt = 0.2;
  tail.sample_capacity = cap;
  tail.n_samples = ns;
  tail.samples = samples;
  const int n = 50;
  auto timed = [&](const char* what, auto&& body) {
    float total = 0;
    for (int i = 0; i < n + 3; ++i) {
      CK(hipMemcpyAsync(Tw, dT, (size_t)P * S * 8, hipMemcpyDeviceToDevice, st));
      CK(hipEventRecord(e0, st));
      body();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (i >= 3) total += ms;
    }
    printf("P=%d S=%d %s: %.2f us\n", P, S, what, total * 1e3 / n);
  };
  timed("separate launches (solve, maxima, solve with scaling + sampling)", [&] {
    RowsTail t2 = tail;
    t2.maxima = maxima;
    CK(launch_solve_rows(b, d, dmask, dvals, Tw, coeffs, nullptr, nullptr, nullptr, st));
    CK(launch_segment_maxima(b, coeffs, Tw, maxima, st));
    CK(launch_solve_rows(b, d, dmask, dvals, Tw, coeffs, status, cost, opt_status, st, t2));
  });
  std::vector<double> c_sep((size_t)P * S * 40), c_one((size_t)P * S * 40);
  CK(hipMemcpy(c_sep.data(), coeffs, c_sep.size() * 8, hipMemcpyDeviceToHost));
  timed("one launch", [&] {
    RowsTail t2 = tail;
    t2.maxima_in_launch = true;
    CK(launch_solve_rows(b, d, dmask, dvals, Tw, coeffs, status, cost, opt_status, st, t2));
  });
  CK(hipMemcpy(c_one.data(), coeffs, c_one.size() * 8, hipMemcpyDeviceToHost));
  size_t diff = 0;
  for (size_t i = 0; i < c_sep.size(); ++i) diff += c_sep[i] != c_one[i];
  long long clk[32];
  CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_phase_clock), sizeof(clk)));
  std::vector<int32_t> hn(P);
  CK(hipMemcpy(hn.data(), ns, (size_t)P * 4, hipMemcpyDeviceToHost));
  printf("  coefficients that differ between the two: %zu; middle workgroup (%d samples), shader cycles: stage %lld | first solve + recover %lld | "
         "maxima (both wavefronts) %lld | scaling %lld | second solve + recover + cost %lld | sampling %lld | whole %lld\n",
         diff, hn[P / 2], clk[1] - clk[0], clk[6] - clk[1], clk[7] - clk[6], clk[14] - clk[7], clk[5] - clk[14], clk[8] - clk[5],
         clk[8] - clk[0]);
  return 0;
}
