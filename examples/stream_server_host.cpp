// stream_server_host.cpp -- a C++ host that keeps several batches in flight on one MI355X through the C ABI.
//
// What a planning server for a swarm does with the library: a fixed batch shape (here 1024 paths of 10 segments), K batches
// in flight, each with its own context = HIP stream, plan and device buffers; the solves are bound once
// (mrs_tg_plan_bind_solve) and issued round-robin by the library's own loop (mrs_tg_bound_solve_launch_many); the options
// carry MRS_TG_FLAG_SHARED_DEVICE because several batches share the device.  The host needs the HIP runtime only to own
// its device buffers and streams.  Prints one JSON object: throughput, and a checksum of the first batch's coefficients
// against a plain mrs_tg_solve_batch of the same paths.
//
//   g++ -std=c++17 -O1 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/stream_server_host.cpp \
//       -L mrs_uav_trajectory_generation_amd -lmrs_tg -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,... -o stream_server_host
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mrs_tg.h"

#define HIP_OK(x)                                                                      \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      std::exit(1);                                                                    \
    }                                                                                  \
  } while (0)
#define TG_OK(ctx, x)                                                                          \
  do {                                                                                         \
    int rc_ = (x);                                                                             \
    if (rc_ != MRS_TG_OK) {                                                                    \
      std::fprintf(stderr, "%s:%d rc %d: %s\n", __FILE__, __LINE__, rc_, mrs_tg_last_error(ctx)); \
      std::exit(1);                                                                            \
    }                                                                                          \
  } while (0)

template <typename T>
static T* to_device(const std::vector<T>& h) {
  T* d = nullptr;
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&d), h.size() * sizeof(T)));
  HIP_OK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  const int P = argc > 1 ? std::atoi(argv[1]) : 1024, S = 10, K = argc > 2 ? std::atoi(argv[2]) : 4;
  const int steps = argc > 3 ? std::atoi(argv[3]) : 2000;
  const size_t nS = (size_t)P * S, nV = nS + P;
  // rest-to-rest paths through pseudo-random waypoints (a linear congruential generator: no dependence on a library)
  std::vector<int32_t> so(P + 1);
  for (int p = 0; p <= P; ++p) so[p] = p * S;
  std::vector<double> wp(nV * 4), vals(nV * 20, 0.0), limits((size_t)P * 9), times(nS);
  std::vector<uint8_t> mask(nV * 5, 0);
  uint64_t state = 12345;
  auto uniform = [&]() {
    state = state * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(state >> 11) / 9007199254740992.0;
  };
  for (int p = 0; p < P; ++p)
    for (int v = 0; v <= S; ++v) {
      const size_t g = (size_t)p * (S + 1) + v;
      for (int dd = 0; dd < 4; ++dd) {
        wp[g * 4 + dd] = (dd < 3 ? 20.0 : 1.0) * (uniform() - 0.5);
        vals[g * 20 + dd] = wp[g * 4 + dd];
      }
      const bool end = v == 0 || v == S;
      for (int k = 0; k < 5; ++k) mask[g * 5 + k] = (k == 0 || end) ? 1 : 0;  // position everywhere, rest at both ends
    }
  const double lim[9] = {2, 2, 1, 2, 2, 2, 20, 20, 20};  // [speed, acceleration, jerk] x [horizontal, vertical, heading]
  for (int p = 0; p < P; ++p)
    for (int k = 0; k < 9; ++k) limits[(size_t)p * 9 + k] = lim[k];

  // reference result of one batch through the one-call host interface (also gives the Euclidean segment times)
  mrs_tg_ctx* ctx0 = nullptr;
  TG_OK(nullptr, mrs_tg_create(0, &ctx0));
  mrs_tg_options opt;
  mrs_tg_default_options(&opt);
  opt.estimate_times = 1;
  std::vector<double> ref_coeffs(nS * 40), cost(P);
  std::vector<int32_t> status(P);
  TG_OK(ctx0, mrs_tg_solve_batch(ctx0, P, so.data(), wp.data(), mask.data(), vals.data(), limits.data(), &opt, times.data(),
                                 ref_coeffs.data(), status.data(), cost.data(), nullptr, nullptr));
  for (int p = 0; p < P; ++p)
    if (status[p] != MRS_TG_STATUS_SUCCESS) {
      std::fprintf(stderr, "path %d: status %d\n", p, status[p]);
      return 1;
    }

  // K batches in flight
  struct Lane {
    hipStream_t stream;
    mrs_tg_ctx* ctx;
    mrs_tg_plan* plan;
    double *times, *coeffs, *cost;
    int32_t* status;
    mrs_tg_bound_solve* bound;
  };
  uint8_t* d_mask = to_device(mask);
  double* d_vals = to_device(vals);
  std::vector<Lane> lanes(K);
  std::vector<mrs_tg_bound_solve*> bound(K);
  mrs_tg_default_options(&opt);
  opt.flags = MRS_TG_FLAG_SHARED_DEVICE;
  for (int k = 0; k < K; ++k) {
    Lane& l = lanes[k];
    // lane 0 on the default stream, the others on non-blocking streams of their own: the runtime spreads streams over four
    // hardware queues and keeps one of them for the default stream (measured: four created streams 5.8 us per step, the
    // default stream + three 3.9)
    l.stream = nullptr;
    if (k > 0) HIP_OK(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
    TG_OK(nullptr, mrs_tg_create(0, &l.ctx));
    TG_OK(l.ctx, mrs_tg_set_stream(l.ctx, l.stream));
    TG_OK(l.ctx, mrs_tg_plan_create(l.ctx, P, so.data(), &l.plan));
    l.times = to_device(times);
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&l.coeffs), nS * 40 * sizeof(double)));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&l.cost), (size_t)P * sizeof(double)));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&l.status), (size_t)P * sizeof(int32_t)));
    TG_OK(l.ctx, mrs_tg_plan_bind_solve(l.plan, nullptr, d_mask, d_vals, nullptr, &opt, l.times, l.coeffs, l.status, l.cost,
                                        nullptr, nullptr, &l.bound));
    bound[k] = l.bound;
  }
  TG_OK(lanes[0].ctx, mrs_tg_bound_solve_launch_many(bound.data(), K, 400));  // warm-up
  HIP_OK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  TG_OK(lanes[0].ctx, mrs_tg_bound_solve_launch_many(bound.data(), K, steps));
  HIP_OK(hipDeviceSynchronize());
  const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  // every lane holds the reference result
  double worst = 0.0;
  std::vector<double> got(nS * 40);
  for (int k = 0; k < K; ++k) {
    HIP_OK(hipMemcpy(got.data(), lanes[k].coeffs, got.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < got.size(); ++i) worst = std::fmax(worst, std::fabs(got[i] - ref_coeffs[i]));
  }
  std::printf("{\"paths\": %d, \"segments\": %d, \"in_flight\": %d, \"steps\": %d, \"us_per_step\": %.3f, "
              "\"trajectories_per_s\": %.1f, \"max_abs_diff_vs_one_call_interface\": %.3e}\n",
              P, S, K, steps, el / steps * 1e6, (double)P * steps / el, worst);
  for (Lane& l : lanes) {
    mrs_tg_bound_solve_destroy(l.bound);
    mrs_tg_plan_destroy(l.plan);
    mrs_tg_destroy(l.ctx);
    (void)hipFree(l.times);
    (void)hipFree(l.coeffs);
    (void)hipFree(l.cost);
    (void)hipFree(l.status);
    if (l.stream) (void)hipStreamDestroy(l.stream);
  }
  (void)hipFree(d_mask);
  (void)hipFree(d_vals);
  mrs_tg_destroy(ctx0);
  return worst == 0.0 ? 0 : 2;
}
