// mrs_tg_policy.hip -- the path-policy layer's C entry points: mrs_tg_policy_host.hpp (what
// MrsTrajectoryGeneration::optimize() does around findTrajectory(), /root/reference/src/mrs_trajectory_generation.cpp:620-851,
// as pure host logic behind a solve callback) instantiated with the batched GPU solve -- all still-active paths of a round go
// out as ONE call of the C ABI's own solver, the arrays of the round in the context's pinned scratch block.
#include <hip/hip_runtime.h>

#include "mrs_tg_policy_host.hpp"
#include "mrs_tg_launch.h"

namespace {

struct GpuHost {  // the Host of mrs_tg::policy::optimize_paths
  mrs_tg_ctx* ctx;
  void* scratch(size_t bytes) { return mrs_tg::ctx_host_scratch(ctx, bytes); }
  int solve(int32_t n_paths, const int32_t* seg_offsets, const double* wp, const uint8_t* mask, const double* vals, const double* lim,
            const mrs_tg_options* opt, double* times, int32_t* status, int32_t* n_samples, double* samples) {
    return mrs_tg::solve_batch_samples_only(ctx, n_paths, seg_offsets, wp, mask, vals, lim, opt, times, status, n_samples, samples);
  }
  int fail(int code, const char* message) { return mrs_tg::report_error(ctx, code, "%s", message); }
  // the rounds' batch-sized work on the device (mrs_tg_abi.hip::policy_round_device); MRS_TG_POLICY_DEVICE=0: the host route
  // above (vertices, gates and validation on the policy's host threads), as until round 5
  // From 64 active requests on: below, a round's two extra launches and the second copy kernel cost a lone request more than
  // its 36 KB of values and samples on the wire (one 5-round request: 1.15 ms on the host route, 1.97 on the device route).
  bool device_round_enabled(size_t active_paths) const {
    static const int min_paths = [] {
      const char* e = std::getenv("MRS_TG_POLICY_DEVICE");   // 0: never; n >= 1: from n active requests on
      return e == nullptr ? 64 : std::atoi(e);
    }();
    return min_paths > 0 && active_paths >= (size_t)min_paths;
  }
  int round(const mrs_tg::PolicyRoundIn& in) { return mrs_tg::policy_round_device(ctx, in); }
};

}  // namespace

extern "C" {

void mrs_tg_default_policy_options(mrs_tg_policy_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  mrs_tg_default_options(&o->solver);
  mrs_tg::policy::default_policy_fields(o);
}

int mrs_tg_optimize_paths(mrs_tg_ctx* ctx, int32_t n_paths, const int32_t* wp_offsets, const mrs_tg_waypoint* waypoints,
                          const mrs_tg_initial_state* initial_states, const uint8_t* has_initial_state, const double* limits,
                          const uint8_t* relax_heading, const mrs_tg_policy_options* opt, int32_t sample_capacity,
                          int32_t* success_out, int32_t* n_samples_out, double* samples_out, double* max_deviation_out,
                          int32_t* n_waypoints_out, int32_t* iterations_out) {
  if (!ctx) return mrs_tg::report_error(nullptr, MRS_TG_ERR_INVALID_ARG, "ctx is NULL");
  try {  // (host containers sized by the batch, on the calling thread and on the policy's worker threads, whose exceptions
         // mrs_tg::policy::parallel_ranges carries back to this one: nothing crosses the C boundary)
    GpuHost host{ctx};
    return mrs_tg::policy::optimize_paths(host, n_paths, wp_offsets, waypoints, initial_states, has_initial_state, limits,
                                          relax_heading, opt, sample_capacity, success_out, n_samples_out, samples_out,
                                          max_deviation_out, n_waypoints_out, iterations_out);
  } catch (const std::bad_alloc&) {
    return mrs_tg::report_error(ctx, MRS_TG_ERR_NOMEM, "out of host memory for %d requests of up to %d samples", n_paths, sample_capacity);
  } catch (const std::exception& ex) {
    return mrs_tg::report_error(ctx, MRS_TG_ERR_HIP, "mrs_tg_optimize_paths: %s", ex.what());
  }
}

// getWaypointInTrajectoryIdxs :1461-1499 for one path: returns the number of indices written
int32_t mrs_tg_waypoint_trajectory_idxs(const double* samples, int32_t n_samples, const mrs_tg_waypoint* waypoints,
                                        int32_t n_waypoints, int32_t* idxs_out) {
  return mrs_tg::policy::waypoint_trajectory_idxs(samples, n_samples, waypoints, n_waypoints, idxs_out);
}

// estimateSegmentTimesBaca (src/eth_trajectory_generation/vertex.cpp:301-485) for one path's vertices, as findTrajectory calls it
// (:1048-1049): waypoints [n_waypoints][4] with headings ALREADY unwrapped, limits9 after relax_heading
int mrs_tg_estimate_times_baca(const double* waypoints, int32_t n_waypoints, const double* limits9, double* seg_times_out) {
  if (!waypoints || !limits9 || !seg_times_out || n_waypoints < 2)
    return mrs_tg::report_error(nullptr, MRS_TG_ERR_INVALID_ARG, "mrs_tg_estimate_times_baca: need >= 2 waypoints, limits and an output array");
  try {
    std::vector<double> t;
    mrs_tg::policy::estimate_times_baca(n_waypoints - 1, waypoints, limits9, t);
    std::memcpy(seg_times_out, t.data(), sizeof(double) * t.size());
  } catch (const std::bad_alloc&) {
    return mrs_tg::report_error(nullptr, MRS_TG_ERR_NOMEM, "out of host memory");
  }
  return MRS_TG_OK;
}

}  // extern "C"
