// request_latency_host.cpp -- the latency of ONE request through the C ABI, from a host without HIP or torch: the call a
// drop-in nodelet makes per path (mrs_tg_find_trajectory: vertices, time estimate, Mellinger outer loop, feasibility scaling,
// sampling; src/mrs_trajectory_generation.cpp:857-1209 of the reference), host buffers in and out.
//   g++ -std=c++17 -O2 -I include examples/request_latency_host.cpp -o request_latency_host -L mrs_uav_trajectory_generation_amd \
//       -lmrs_tg -Wl,-rpath,$PWD/mrs_uav_trajectory_generation_amd && ./request_latency_host [n_waypoints] [calls]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mrs_tg.h"

int main(int argc, char** argv) {
  const int n_wp = argc > 1 ? std::atoi(argv[1]) : 11, calls = argc > 2 ? std::atoi(argv[2]) : 300, cap = 1024;
  mrs_tg_ctx* ctx = nullptr;
  if (mrs_tg_create(0, &ctx) != MRS_TG_OK) {
    std::fprintf(stderr, "mrs_tg_create: %s\n", mrs_tg_last_error(nullptr));
    return 1;
  }
  std::vector<mrs_tg_waypoint> wp(n_wp);
  unsigned rng = 12345u;
  auto uni = [&] {
    rng = rng * 1664525u + 1013904223u;
    return (rng >> 8) * (1.0 / 16777216.0);
  };
  for (int i = 0; i < n_wp; ++i) {
    wp[i].coords[0] = uni() * 20.0 - 10.0;
    wp[i].coords[1] = uni() * 20.0 - 10.0;
    wp[i].coords[2] = uni() * 4.0 + 1.0;
    wp[i].coords[3] = uni() * 6.0 - 3.0;
    wp[i].stop_at = 0;
  }
  const double limits[9] = {4.0, 2.0, 1.0, 2.0, 1.0, 1.0, 20.0, 20.0, 10.0};
  mrs_tg_options opt;
  mrs_tg_default_options(&opt);
  opt.derivative_to_optimize = 4;
  opt.time_alloc_method = MRS_TG_TIME_ALLOC_MELLINGER;
  opt.sampling_dt = 0.2;
  opt.sample_capacity = cap;
  const int S = n_wp - 1;
  std::vector<double> times(S), coeffs((size_t)S * 40), samples((size_t)cap * 4), us;
  int32_t status = 0, n_samples = 0;
  for (int i = 0; i < calls + 20; ++i) {
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = mrs_tg_find_trajectory(ctx, wp.data(), n_wp, nullptr, limits, &opt, 0, times.data(), coeffs.data(), &status,
                                          &n_samples, samples.data());
    const auto t1 = std::chrono::steady_clock::now();
    if (rc != MRS_TG_OK) {
      std::fprintf(stderr, "mrs_tg_find_trajectory: %s\n", mrs_tg_last_error(ctx));
      return 1;
    }
    if (i >= 20) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
  }
  std::sort(us.begin(), us.end());
  double total = 0.0;
  for (int i = 0; i < S; ++i) total += times[i];
  std::printf("one request, %d segments (status %d, %d samples, %.3f s of trajectory): median %.1f us, min %.1f us, 90%% %.1f us over %d calls\n",
              S, status, n_samples, total, us[us.size() / 2], us.front(), us[us.size() * 9 / 10], calls);
  mrs_tg_destroy(ctx);
  return 0;
}
