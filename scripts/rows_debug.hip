// rows_debug.hip -- dumps the per-lane columns of solve_rows_kernel after the build and after the middle vertex for one
// small path (development aid, not part of the library)
#define MRS_TG_ROWS_DEBUG 1
#include "../mrs_uav_trajectory_generation_amd/csrc/mrs_tg_rows.hip"
#include "../mrs_uav_trajectory_generation_amd/csrc/mrs_tg_pool.hip"
#include <cstdio>
#include <vector>
using namespace mrs_tg;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <typename T> static T* to_dev(const std::vector<T>& h) { T* d; hipMalloc(&d, h.size() * sizeof(T)); hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice); return d; }
int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 3, P = 1, V = S + 1, d = 4;
  std::vector<int32_t> so = {0, S}, order = {0}, slot(S + 1, 0);
  std::vector<uint8_t> mask(V * 5, 0);
  std::vector<double> vals(V * 20, 0.0), T(S);
  for (int v = 0; v < V; ++v) {
    const bool end = v == 0 || v == S;
    for (int k = 0; k < 5; ++k) mask[v * 5 + k] = (k == 0 || end) ? 1 : 0;
    for (int dim = 0; dim < 4; ++dim) vals[v * 20 + dim] = (v + 1) * (dim + 1) * 0.5 + v * v * 0.25;
  }
  for (int i = 0; i < S; ++i) T[i] = 1.0 + 0.5 * i;
  BatchView b{P, S, S, S, to_dev(so), to_dev(order), to_dev(slot)};
  double *coeffs, *cost; int32_t* status;
  CK(hipMalloc(&coeffs, S * 40 * 8)); CK(hipMalloc(&cost, 8)); CK(hipMalloc(&status, 4));
  CK(launch_solve_rows(b, d, to_dev(mask), to_dev(vals), to_dev(T), coeffs, status, cost, nullptr, 0));
  CK(hipDeviceSynchronize());
  std::vector<double> dbg(64 * 24 * 2);
  CK(hipMemcpyFromSymbol(dbg.data(), HIP_SYMBOL(g_rows_debug), dbg.size() * 8));
  for (int phase = 0; phase < 2; ++phase) {
    printf("---- %s\n", phase ? "after middle" : "after build");
    for (int lane = 0; lane < 64; ++lane) {
      if ((lane >> 4) & 1) continue;
      const double* r = &dbg[phase * 64 * 24 + lane * 24];
      printf("lane %2d (dir %d quad %d k %d) nact %g:", lane, lane >> 5, (lane >> 2) & 3, lane & 3, r[20]);
      for (int j = 0; j < 16; ++j) printf(" %s%.5g", j % 4 == 0 ? "| " : "", r[j]);
      printf(" || B %.5g %.5g %.5g %.5g  c %.5g %.5g %.5g\n", r[16], r[17], r[18], r[19], r[21], r[22], r[23]);
    }
  }
  std::vector<double> hc(S * 40);
  CK(hipMemcpy(hc.data(), coeffs, hc.size() * 8, hipMemcpyDeviceToHost));
  for (int i = 0; i < S; ++i) { printf("seg %d dim0 c:", i); for (int k = 0; k < 10; ++k) printf(" %.6g", hc[i * 40 + k]); printf("\n"); }
  return 0;
}
