// mrs_tg_service.hpp -- the nodelet's service layer without ROS: PathSrv / GetPathSrv in, TrajectoryReference out,
// for one request or a batch of requests, over the C ABI (mrs_tg.h).
//
// It restates the host logic of MrsTrajectoryGeneration::callbackPathSrv / callbackGetPathSrv
// (/root/reference/src/mrs_trajectory_generation.cpp:1968-2190, 2196-2451) that sits above optimize():
//   * request validation and the reference's error strings ("missing constraints", "received an empty message",
//     "invalid path", "the path is empty (after postprocessing)", "failed to find trajectory");
//   * the Path fields read at :2061-2095 (use_heading, fly_now, stop_at_waypoints, loop, relax_heading,
//     override_constraints + override_max_*, max_deviation_from_path, dont_prepend_current_state, input_id);
//   * the constraint override with its feasibility test against the initial state (:997-1026), including the
//     reference's assignment of the horizontal jerk override to the vertical jerk limit (:2071, quirk B2);
//   * prepending the current state as the initial condition (:660-674);
//   * the n_attempts loop with the fallback sampler on the last attempt (:2131-2150);
//   * TrajectoryReference assembly (getTrajectoryReference :1560-1606) and getWaypointInTrajectoryIdxs (:1461-1499).
// What needs ROS is left to the caller: message <-> struct conversion, tf (transformPath), time stamps / "path from
// the future" splicing of the MPC prediction, publishing.  All requests of a call that share a policy are solved in
// ONE batched GPU call per attempt (mrs_tg_optimize_paths); the reference serves one request at a time.
//
// Plain structs stand in for the mrs_msgs types so that the header compiles anywhere; INTEGRATION.md shows the
// field-by-field conversion.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <limits>
#include <chrono>
#include <map>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "mrs_tg.h"

namespace mrs_tg {

struct Reference {  // mrs_msgs::Reference
  double x = 0, y = 0, z = 0, heading = 0;
};

struct Path {  // mrs_msgs::Path, the fields read at src/...cpp:2061-2095
  std::string frame_id;
  uint64_t input_id = 0;
  std::vector<Reference> points;
  bool fly_now = false, use_heading = false, stop_at_waypoints = false, loop = false, relax_heading = false;
  bool override_constraints = false;
  double override_max_velocity_horizontal = 0, override_max_velocity_vertical = 0;
  double override_max_acceleration_horizontal = 0, override_max_acceleration_vertical = 0;
  double override_max_jerk_horizontal = 0, override_max_jerk_vertical = 0;
  double max_deviation_from_path = 0;
  double max_execution_time = 0;  // > 0: this request's budget (:1862-1866); else the max_time parameter
  bool dont_prepend_current_state = false;
};

struct TrajectoryReference {  // mrs_msgs::TrajectoryReference as filled at :1564-1603, :2159
  std::string frame_id;
  uint64_t input_id = 0;
  bool use_heading = false, fly_now = false, loop = false;
  double dt = 0;
  std::vector<Reference> points;
};

struct GetPathResponse {  // mrs_msgs::GetPathSrv::Response (PathSrv::Response = success + message)
  bool success = false;
  std::string message;
  TrajectoryReference trajectory;
  std::vector<int32_t> waypoint_trajectory_idxs;
  double max_deviation = 0;  // final max trajectory-path deviation (:729)
};

struct Constraints {  // mrs_msgs::DynamicsConstraints, the fields read at :985-1037
  double horizontal_speed = 0, horizontal_acceleration = 0, horizontal_jerk = 0;
  double vertical_ascending_speed = 0, vertical_descending_speed = 0;
  double vertical_ascending_acceleration = 0, vertical_descending_acceleration = 0;
  double vertical_ascending_jerk = 0, vertical_descending_jerk = 0;
  double heading_speed = 0, heading_acceleration = 0, heading_jerk = 0;
};

struct CurrentState {  // the mrs_msgs::TrackerCommand fields used as the initial condition (:925-957)
  Reference position;  // position + heading
  std::array<double, 4> velocity{}, acceleration{}, jerk{};  // xyz + heading rate / acceleration / jerk
};

struct ServiceParams {  // config/{public,private}/trajectory_generation.yaml
  int n_attempts = 3;                    // n_attempts
  bool fallback_sampling_enabled = true; // fallback_sampling/enabled
  double max_time = 0.5;                 // max_time [s] (config/public/trajectory_generation.yaml:4); <= 0: no deadline
  int sample_capacity = 8192;            // capacity of one trajectory in samples (host buffer size, not a reference parameter)
  mrs_tg_policy_options policy{};        // everything optimize() / findTrajectory() read
  ServiceParams() {
    mrs_tg_default_policy_options(&policy);
    policy.solver.time_alloc_method = MRS_TG_TIME_ALLOC_MELLINGER;  // time_allocation: 2
    policy.solver.derivative_to_optimize = 2;                       // derivative_to_optimize: 0 -> acceleration
    policy.solver.sampling_dt = 0.2;                                // sampling_dt
  }
};

class PathService {
public:
  explicit PathService(int device = 0, const ServiceParams& params = ServiceParams()) : params_(params) {
    if (mrs_tg_create(device, &ctx_) != MRS_TG_OK) throw std::runtime_error(mrs_tg_last_error(nullptr));
  }
  ~PathService() { mrs_tg_destroy(ctx_); }
  PathService(const PathService&) = delete;
  PathService& operator=(const PathService&) = delete;

  ServiceParams& params() { return params_; }
  void setConstraints(const Constraints& c) { constraints_ = c; }     // sh_constraints_
  void setCurrentState(const CurrentState& s) { state_ = s; }         // sh_tracker_cmd_ / prepareInitialCondition
  void clearCurrentState() { state_.reset(); }

  // callbackGetPathSrv for one request
  GetPathResponse getPath(const Path& path) { return getPaths({path}).front(); }

  // ... and for a batch of independent requests: one GPU call per attempt and per distinct deviation limit
  std::vector<GetPathResponse> getPaths(const std::vector<Path>& requests) {
    const size_t R = requests.size();
    std::vector<GetPathResponse> res(R);
    std::vector<Job> jobs;
    jobs.reserve(R);
    for (size_t r = 0; r < R; ++r) {
      const Path& path = requests[r];
      GetPathResponse& out = res[r];
      if (!constraints_) {  // :1976-1984
        out.message = "missing constraints";
        continue;
      }
      if (path.points.empty()) {  // :2033-2041
        out.message = "received an empty message";
        continue;
      }
      Job job;
      job.request = r;
      bool finite = true;
      for (const Reference& p : path.points) {  // checkNaN :2097-2113
        finite = finite && std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z) && std::isfinite(p.heading);
        job.waypoints.push_back(make_waypoint(p, path.stop_at_waypoints));
      }
      if (!finite) {
        out.message = "invalid path";
        continue;
      }
      if (path.loop) job.waypoints.push_back(job.waypoints.front());  // :2118-2120
      job.n_requested = job.waypoints.size();
      // the initial condition is prepended when there is one (:660-674); without one a "fly now" is dropped
      job.fly_now = path.fly_now;
      job.relax_heading = path.relax_heading;
      if (state_ && !path.dont_prepend_current_state) {
        job.has_initial_state = true;
        job.initial_state.heading = state_->position.heading;
        for (int k = 0; k < 4; ++k) {
          job.initial_state.velocity[k] = state_->velocity[k];
          job.initial_state.acceleration[k] = state_->acceleration[k];
          job.initial_state.jerk[k] = state_->jerk[k];
        }
        job.waypoints.insert(job.waypoints.begin(), make_waypoint(state_->position, false));
      } else if (!path.dont_prepend_current_state) {
        job.fly_now = false;
      }
      job.limits = limits_for(path, job.has_initial_state ? &job.initial_state : nullptr);
      job.max_deviation = path.max_deviation_from_path > 0 ? path.max_deviation_from_path : params_.policy.max_deviation;  // :2085-2089
      job.max_execution_time = path.max_execution_time > 0 ? path.max_execution_time : params_.max_time;                   // :1862-1872
      jobs.push_back(std::move(job));
    }

    std::vector<size_t> pending(jobs.size());
    for (size_t j = 0; j < jobs.size(); ++j) pending[j] = j;
    const int attempts = params_.n_attempts > 0 ? params_.n_attempts : 1;
    for (int attempt = 0; attempt < attempts && !pending.empty(); ++attempt) {
      // the last attempt uses the fallback sampler when it is enabled (:2136)
      const bool fallback = (attempts > 1) && (attempt == attempts - 1) && params_.fallback_sampling_enabled;
      std::map<double, std::vector<size_t>> by_deviation;  // requests sharing a policy share a GPU call
      for (size_t j : pending) by_deviation[jobs[j].max_deviation].push_back(j);
      std::vector<size_t> still;
      for (auto& [max_deviation, group] : by_deviation) {
        // start_time_total_ (:2008) is set when the callback starts to work on a request, not while the request waits in
        // the service queue: a request's clock starts when its group's first GPU call is set up and runs on through the
        // later attempts.  The requests of one GPU call share that call, so it gets the tightest time left of its group;
        // a group that is already late is not given up -- the policy layer runs the fallback sampler for it, as
        // optimize() does when overtime() holds (:711-713).
        const auto now = std::chrono::steady_clock::now();
        double budget = 0;
        for (size_t j : group) {
          Job& job = jobs[j];
          if (!job.started) {
            job.started = true;
            job.t_start = now;
          }
          if (job.max_execution_time <= 0) continue;
          double left = job.max_execution_time - std::chrono::duration<double>(now - job.t_start).count();
          if (left <= 0) left = 1e-9;
          if (budget <= 0 || left < budget) budget = left;
        }
        solve_group(jobs, group, max_deviation, fallback, budget);
        for (size_t j : group)
          if (!jobs[j].success) still.push_back(j);
      }
      pending.swap(still);
    }

    for (Job& job : jobs) {
      GetPathResponse& out = res[job.request];
      const Path& path = requests[job.request];
      out.success = job.success;
      out.message = job.message;
      out.max_deviation = job.max_deviation_out;
      if (!job.success) continue;
      TrajectoryReference& t = out.trajectory;  // getTrajectoryReference :1564-1603
      t.frame_id = path.frame_id;
      t.input_id = path.input_id;                // :2159
      t.use_heading = path.use_heading;
      t.fly_now = job.fly_now;
      t.loop = path.loop;
      t.dt = params_.policy.solver.sampling_dt;
      t.points = std::move(job.samples);
      // getWaypointInTrajectoryIdxs over the waypoints as requested (initial condition excluded, :2392)
      std::vector<int32_t> idx(job.waypoints.size() + 1);
      std::vector<double> flat(t.points.size() * 4);
      for (size_t i = 0; i < t.points.size(); ++i) {
        flat[4 * i] = t.points[i].x;
        flat[4 * i + 1] = t.points[i].y;
        flat[4 * i + 2] = t.points[i].z;
        flat[4 * i + 3] = t.points[i].heading;
      }
      const mrs_tg_waypoint* first = job.waypoints.data() + (job.has_initial_state ? 1 : 0);
      const int32_t n = mrs_tg_waypoint_trajectory_idxs(flat.data(), static_cast<int32_t>(t.points.size()), first,
                                                        static_cast<int32_t>(job.n_requested), idx.data());
      out.waypoint_trajectory_idxs.assign(idx.begin(), idx.begin() + n);
    }
    return res;
  }

private:
  struct Job {
    size_t request = 0, n_requested = 0;
    std::vector<mrs_tg_waypoint> waypoints;
    bool has_initial_state = false, fly_now = false, relax_heading = false;
    mrs_tg_initial_state initial_state{};
    std::array<double, 9> limits{};
    double max_deviation = 0, max_deviation_out = 0;
    double max_execution_time = 0;  // this request's budget [s], <= 0: none
    bool started = false;           // t_start is set: the request's clock runs (start_time_total_, :2008)
    std::chrono::steady_clock::time_point t_start{};
    bool success = false;
    std::string message;
    std::vector<Reference> samples;
  };

  static mrs_tg_waypoint make_waypoint(const Reference& p, bool stop_at) {
    mrs_tg_waypoint w{};
    w.coords[0] = p.x;
    w.coords[1] = p.y;
    w.coords[2] = p.z;
    w.coords[3] = p.heading;
    w.stop_at = stop_at ? 1 : 0;
    return w;
  }

  // limits9 = {v, a, j} x {horizontal, vertical, heading} as findTrajectory() derives them (:985-1037)
  std::array<double, 9> limits_for(const Path& path, const mrs_tg_initial_state* init) const {
    const Constraints& c = *constraints_;
    double vh = c.horizontal_speed, ah = c.horizontal_acceleration, jh = c.horizontal_jerk;
    double vv = std::min(c.vertical_ascending_speed, c.vertical_descending_speed);
    double av = std::min(c.vertical_ascending_acceleration, c.vertical_descending_acceleration);
    double jv = std::min(c.vertical_ascending_jerk, c.vertical_descending_jerk);
    if (path.override_constraints) {
      // the reference stores the HORIZONTAL jerk override in its vertical jerk member (:2071)
      const double o_jv = path.override_max_jerk_horizontal;
      bool can_change = true;
      if (init) {  // :1001-1007
        can_change = std::hypot(init->velocity[0], init->velocity[1]) < path.override_max_velocity_horizontal &&
                     std::hypot(init->acceleration[0], init->acceleration[1]) < path.override_max_acceleration_horizontal &&
                     std::hypot(init->jerk[0], init->jerk[1]) < path.override_max_jerk_horizontal &&
                     std::fabs(init->velocity[2]) < path.override_max_velocity_vertical &&
                     std::fabs(init->acceleration[2]) < path.override_max_acceleration_vertical && std::fabs(init->jerk[2]) < o_jv;
      }
      if (can_change) {
        vh = path.override_max_velocity_horizontal;
        ah = path.override_max_acceleration_horizontal;
        jh = path.override_max_jerk_horizontal;
        vv = path.override_max_velocity_vertical;
        av = path.override_max_acceleration_vertical;
        jv = o_jv;
      }
    }
    // relax_heading is applied by the library (limits of float max, :1030-1034) from the per-path flag
    return {vh, vv, c.heading_speed, ah, av, c.heading_acceleration, jh, jv, c.heading_jerk};
  }

  void solve_group(std::vector<Job>& jobs, const std::vector<size_t>& group, double max_deviation, bool fallback, double budget_s) {
    const int32_t P = static_cast<int32_t>(group.size());
    std::vector<int32_t> off(P + 1, 0);
    std::vector<mrs_tg_waypoint> wps;
    std::vector<mrs_tg_initial_state> inits(P);
    std::vector<uint8_t> has(P), relax(P);
    std::vector<double> limits(static_cast<size_t>(P) * 9);
    for (int32_t p = 0; p < P; ++p) {
      const Job& job = jobs[group[p]];
      wps.insert(wps.end(), job.waypoints.begin(), job.waypoints.end());
      off[p + 1] = static_cast<int32_t>(wps.size());
      inits[p] = job.initial_state;
      has[p] = job.has_initial_state ? 1 : 0;
      relax[p] = job.relax_heading ? 1 : 0;
      for (int k = 0; k < 9; ++k) limits[static_cast<size_t>(p) * 9 + k] = job.limits[k];
    }
    mrs_tg_policy_options pol = params_.policy;
    pol.max_deviation = max_deviation;
    pol.fallback_sampling = fallback ? 1 : pol.fallback_sampling;
    pol.max_execution_time_s = budget_s;
    const int cap = params_.sample_capacity;
    std::vector<int32_t> success(P), n_samples(P);
    std::vector<double> samples(static_cast<size_t>(P) * cap * 4), max_dev(P);
    const int rc = mrs_tg_optimize_paths(ctx_, P, off.data(), wps.data(), inits.data(), has.data(), limits.data(), relax.data(), &pol,
                                         cap, success.data(), n_samples.data(), samples.data(), max_dev.data(), nullptr, nullptr);
    for (int32_t p = 0; p < P; ++p) {
      Job& job = jobs[group[p]];
      if (rc != MRS_TG_OK) {
        job.success = false;
        job.message = std::string("failed to find trajectory: ") + mrs_tg_last_error(ctx_);
        continue;
      }
      job.success = success[p] != 0;
      job.max_deviation_out = max_dev[p];
      if (!job.success) {
        // optimize() reports an unusable path and a failed solve differently (:678, :724)
        job.message = (job.waypoints.size() <= 1) ? "the path is empty (after postprocessing)" : "failed to find trajectory";
        continue;
      }
      job.message = "trajectory generated";  // :846
      job.samples.resize(n_samples[p]);
      const double* s = samples.data() + static_cast<size_t>(p) * cap * 4;
      for (int32_t i = 0; i < n_samples[p]; ++i) job.samples[i] = {s[4 * i], s[4 * i + 1], s[4 * i + 2], s[4 * i + 3]};
    }
  }

private:
  mrs_tg_ctx* ctx_ = nullptr;
  ServiceParams params_;
  std::optional<Constraints> constraints_;
  std::optional<CurrentState> state_;
};

}  // namespace mrs_tg
