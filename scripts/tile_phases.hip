// tile_phases.hip -- where the time of solve_tile_kernel goes: shader-clock stamps at the phase boundaries of
// one workgroup (MRS_TG_PHASE_CLOCKS), plus the launch-to-launch time of the kernel.  Not part of the library.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mrs_uav_trajectory_generation_amd/csrc -I include \
//         scripts/tile_phases.hip -o scripts/tile_phases.bin && scripts/tile_phases.bin [P] [S]
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
  const int P = argc > 1 ? atoi(argv[1]) : 1024, S = argc > 2 ? atoi(argv[2]) : 10, d = 4;
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
  // segment times as the Euclidean estimator would give them at 2 m/s (so that the outer loop runs a realistic
  // number of ticks)
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
  double *dvals = to_dev(vals), *dT = to_dev(T), *H, *A, *coeffs, *cost;
  int32_t* status;
  CK(hipMalloc(&H, (size_t)P * S * 100 * 8));
  CK(hipMalloc(&A, (size_t)P * S * 100 * 8));
  CK(hipMalloc(&coeffs, (size_t)P * S * 40 * 8));
  CK(hipMalloc(&cost, (size_t)P * 8));
  CK(hipMalloc(&status, (size_t)P * 4));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  CK(launch_assemble(b, d, dT, H, A, st));
  if (!tile_kernel_applies(b, false)) {
    printf("tile kernel does not apply to P=%d S=%d\n", P, S);
    return 0;
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int fused = 0; fused < 2; ++fused) {
    for (int i = 0; i < 10; ++i) CK(launch_solve_tile(b, d, fused, dmask, dvals, dT, H, A, coeffs, status, cost, nullptr, st));
    CK(hipStreamSynchronize(st));
    const int n = 200;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < n; ++i) CK(launch_solve_tile(b, d, fused, dmask, dvals, dT, H, A, coeffs, status, cost, nullptr, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    long long clk[32];
    CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_phase_clock), sizeof(clk)));
    printf("P=%d S=%d %s: %.2f us per launch back to back; phase clocks (shader cycles since kernel entry of the "
           "middle workgroup): setup %lld  A0 %lld  A1 %lld  B %lld  C %lld  total %lld\n",
           P, S, fused ? "fused" : "blocks", ms * 1e3 / n, clk[1] - clk[0], clk[2] - clk[1], clk[3] - clk[2], clk[4] - clk[3],
           clk[5] - clk[4], clk[5] - clk[0]);
    printf("   worker 0 in A1 (cycles after the A0 barrier as seen by thread 0): item start %lld  u, qf done %lld  blocks stored %lld; A^-1 requested %lld cycles into phase B\n",
           clk[16] - clk[2], clk[17] - clk[2], clk[18] - clk[2], clk[19] - clk[3]);
    printf("   inside B: entry %lld  elimination loop %lld  hand-over %lld  middle %lld  fence %lld  back substitution %lld  exit %lld\n",
           clk[10] - clk[3], clk[11] - clk[10], clk[12] - clk[11], clk[13] - clk[12], clk[14] - clk[13], clk[15] - clk[14],
           clk[4] - clk[15]);
  }
  // the rows kernel (one lane per unknown): the library's fused linear solve
  for (int ppw = 1; ppw <= 2; ++ppw) {
    setenv("MRS_TG_ROWS_PPW", ppw == 1 ? "1" : "2", 1);
    for (int i = 0; i < 10; ++i) CK(launch_solve_rows(b, d, dmask, dvals, dT, coeffs, status, cost, nullptr, st));
    CK(hipStreamSynchronize(st));
    const int n = 200;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < n; ++i) CK(launch_solve_rows(b, d, dmask, dvals, dT, coeffs, status, cost, nullptr, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    long long clk[32];
    CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_phase_clock), sizeof(clk)));
    printf("P=%d S=%d rows kernel, %d path(s) per wavefront: %.2f us per launch back to back; middle wavefront (shader cycles): stage %lld  "
           "build %lld  forward %lld  middle vertex %lld  backward %lld  recover %lld  total %lld\n",
           P, S, ppw, ms * 1e3 / n, clk[1] - clk[0], clk[10] - clk[1], clk[11] - clk[10], clk[12] - clk[11], clk[4] - clk[12],
           clk[5] - clk[4], clk[5] - clk[0]);
    printf("   inside stage (last path of the wavefront): requests issued %lld  vertices in LDS %lld  segments in LDS %lld  fence %lld\n",
           clk[20] - clk[0], clk[21] - clk[20], clk[22] - clk[21], clk[1] - clk[22]);
  }
  // sampler: serial walk (lane 0) and parallel evaluation, dt 0.2, capacity 512
  {
    const int cap = 512;
    double* samples;
    int32_t* ns;
    CK(hipMalloc(&samples, (size_t)P * cap * 4 * 8));
    CK(hipMalloc(&ns, (size_t)P * 4));
    for (int i = 0; i < 5; ++i) CK(launch_sample(b, coeffs, dT, 0.2, cap, ns, samples, st));
    CK(hipStreamSynchronize(st));
    const int n = 50;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < n; ++i) CK(launch_sample(b, coeffs, dT, 0.2, cap, ns, samples, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    long long clk[32];
    CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_phase_clock), sizeof(clk)));
    std::vector<int32_t> hn(P);
    CK(hipMemcpy(hn.data(), ns, (size_t)P * 4, hipMemcpyDeviceToHost));
    printf("sample_kernel: %.2f us per launch; middle workgroup (%d samples): load %lld  walk + evaluate %lld cycles\n",
           ms * 1e3 / n, hn[P / 2], clk[1] - clk[0], clk[2] - clk[1]);
  }
  // outer loop (mode 2): one optimiser tick = one objective evaluation (S + 1 forward sweeps) + bookkeeping
  {
    NonlinearPlan nl;
    nonlinear_plan_build(nl, so, order);
    NonlinearParams prm{d, 10, 0.05, -1.0, 0.1, -1.0};
    std::vector<double> lim((size_t)P * 9);
    const double l9[9] = {2, 2, 1, 2, 2, 2, 20, 20, 20};
    for (size_t i = 0; i < lim.size(); ++i) lim[i] = l9[i % 9];
    double* dlim = to_dev(lim);
    double* Tw;
    CK(hipMalloc(&Tw, (size_t)P * S * 8));
    float total = 0;
    const int n = 20;
    for (int i = 0; i < n + 3; ++i) {
      CK(hipMemcpyAsync(Tw, dT, (size_t)P * S * 8, hipMemcpyDeviceToDevice, st));
      CK(hipEventRecord(e0, st));
      CK(launch_nonlinear(nl, b, prm, dmask, dvals, dlim, Tw, coeffs, status, cost, st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (i >= 3) total += ms;
    }
    long long clk[32];
    CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_phase_clock), sizeof(clk)));
    printf("nonlinear pipeline (outer loop + 2 solves + maxima + scaling): %.2f us; optimize_kernel middle workgroup: staging %lld, whole kernel %lld cycles\n",
           total * 1e3 / n, clk[1] - clk[0], clk[5] - clk[0]);
    printf("  tick 1 in detail: accept step %lld, barrier %lld, direction + trial point %lld, flag + barrier %lld\n",
           clk[28] - clk[8], clk[29] - clk[28], clk[30] - clk[29], clk[9] - clk[30]);
    printf("    accept step: vectors read %lld, four sums %lld, stopping rules %lld, pair stored %lld, rest %lld | direction: pairs read %lld, recursion %lld, rest %lld\n",
           clk[23] - clk[8], clk[24] - clk[23], clk[25] - clk[24], clk[26] - clk[25], clk[28] - clk[26], clk[27] - clk[29],
           clk[31] - clk[27], clk[30] - clk[31]);
    for (int t = 0; t < 6; ++t)
      printf("  tick %d: evaluation %lld cycles, bookkeeping + direction %lld\n", t,
             clk[6 + 2 * t] - (t == 0 ? clk[1] : clk[5 + 2 * t]), clk[7 + 2 * t] - clk[6 + 2 * t]);
  }
  std::vector<double> hc((size_t)P);
  CK(hipMemcpy(hc.data(), cost, (size_t)P * 8, hipMemcpyDeviceToHost));
  printf("cost[0..2] = %g %g %g\n", hc[0], hc[1], hc[2]);
  return 0;
}
