// mrs_tg.hpp -- thin header-only C++17 wrapper over the C ABI (mrs_tg.h) with the vocabulary of the
// reference's solver adapter, for a ROS/C++ host.  It mirrors the argument list of
// MrsTrajectoryGeneration::findTrajectory() (/root/reference/src/mrs_trajectory_generation.cpp:857-859):
//
//   std::optional<eth_mav_msgs::EigenTrajectoryPoint::Vector>
//   findTrajectory(const std::vector<Waypoint_t>& waypoints, const std::optional<mrs_msgs::TrackerCommand>& initial_state,
//                  const double& sampling_dt, const bool& relax_heading);
//
// No Eigen / ROS types appear here so that the header compiles anywhere; INTEGRATION.md shows the
// three-line conversion inside the nodelet.
#pragma once
#include <array>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "mrs_tg.h"

namespace mrs_tg {

struct Waypoint {  // Waypoint_t, src/mrs_trajectory_generation.cpp:66-70
  std::array<double, 4> coords;  // x, y, z, heading
  bool stop_at = false;
};

struct InitialState {  // the mrs_msgs::TrackerCommand fields read at src/...cpp:925-957
  double heading = 0;
  std::array<double, 4> velocity{}, acceleration{}, jerk{};
};

struct DynamicsConstraints {  // the mrs_msgs::DynamicsConstraints fields read at src/...cpp:985-1037
  double horizontal_speed, horizontal_acceleration, horizontal_jerk;
  double vertical_ascending_speed, vertical_descending_speed;
  double vertical_ascending_acceleration, vertical_descending_acceleration;
  double vertical_ascending_jerk, vertical_descending_jerk;
  double heading_speed, heading_acceleration, heading_jerk;
  // limits9 = {v,a,j} x {horizontal, vertical, heading}; vertical = min(ascending, descending) (:985-995)
  std::array<double, 9> limits9() const {
    auto mn = [](double a, double b) { return a < b ? a : b; };
    return {horizontal_speed, mn(vertical_ascending_speed, vertical_descending_speed), heading_speed,
            horizontal_acceleration, mn(vertical_ascending_acceleration, vertical_descending_acceleration), heading_acceleration,
            horizontal_jerk, mn(vertical_ascending_jerk, vertical_descending_jerk), heading_jerk};
  }
};

struct TrajectoryPoint {  // what the nodelet reads of EigenTrajectoryPoint (src/...cpp:1582-1599)
  double x, y, z, heading;
};

class TrajectoryGenerator {
public:
  explicit TrajectoryGenerator(int device = 0) {
    if (mrs_tg_create(device, &ctx_) != MRS_TG_OK) throw std::runtime_error(mrs_tg_last_error(nullptr));
    mrs_tg_default_options(&opt_);
    opt_.time_alloc_method = MRS_TG_TIME_ALLOC_MELLINGER;  // config/private/trajectory_generation.yaml:7
    opt_.derivative_to_optimize = 2;                       // :11 (0 -> acceleration)
  }
  ~TrajectoryGenerator() { mrs_tg_destroy(ctx_); }
  TrajectoryGenerator(const TrajectoryGenerator&) = delete;
  TrajectoryGenerator& operator=(const TrajectoryGenerator&) = delete;

  mrs_tg_options& options() { return opt_; }

  // nullopt = failure, exactly where the reference returns {}: a rejected optimiser code (:1146-1149) and a sampled
  // trajectory that fails the temporal sanity check against the Baca estimate (:1178-1199; the factors are
  // options().max_trajectory_len_factor / min_trajectory_len_factor, the reference's parameters of the same names) --
  // both gates are applied inside mrs_tg_find_trajectory, which builds the vertices and their Baca estimate itself.
  // rejection() says which gate, lastError() carries the reference's message, bacaTotalTime() is initial_total_time_baca.
  std::optional<std::vector<TrajectoryPoint>> findTrajectory(const std::vector<Waypoint>& waypoints,
                                                             const std::optional<InitialState>& initial_state,
                                                             const DynamicsConstraints& constraints, double sampling_dt,
                                                             bool relax_heading, int sample_capacity = 8192) {
    std::vector<mrs_tg_waypoint> wp(waypoints.size());
    for (size_t i = 0; i < waypoints.size(); ++i) {
      for (int k = 0; k < 4; ++k) wp[i].coords[k] = waypoints[i].coords[k];
      wp[i].stop_at = waypoints[i].stop_at ? 1 : 0;
    }
    mrs_tg_initial_state init{};
    if (initial_state) {
      init.heading = initial_state->heading;
      for (int k = 0; k < 4; ++k) {
        init.velocity[k] = initial_state->velocity[k];
        init.acceleration[k] = initial_state->acceleration[k];
        init.jerk[k] = initial_state->jerk[k];
      }
    }
    const auto lim = constraints.limits9();
    mrs_tg_options opt = opt_;
    opt.sampling_dt = sampling_dt;
    opt.sample_capacity = sample_capacity;
    const int S = static_cast<int>(waypoints.size()) - 1;
    if (S < 1) return std::nullopt;
    segment_times_.assign(S, 0.0);
    coefficients_.assign(static_cast<size_t>(S) * 40, 0.0);
    std::vector<double> samples(static_cast<size_t>(sample_capacity) * 4);
    int32_t n = 0;
    if (mrs_tg_find_trajectory(ctx_, wp.data(), static_cast<int32_t>(wp.size()), initial_state ? &init : nullptr, lim.data(),
                               &opt, relax_heading ? 1 : 0, segment_times_.data(), coefficients_.data(), &status_, &n,
                               samples.data()) != MRS_TG_OK) {
      last_error_ = mrs_tg_last_error(ctx_);
      return std::nullopt;
    }
    mrs_tg_find_trajectory_info(ctx_, &rejection_, &baca_total_time_);
    if (n <= 0) {  // one of the reference's two gates (or an empty sampling)
      last_error_ = mrs_tg_last_error(ctx_);
      return std::nullopt;
    }
    if (n > sample_capacity) {  // the trajectory passed the gates but does not fit: the caller's capacity is too small
      last_error_ = "the sampled trajectory needs more than sample_capacity samples";
      return std::nullopt;
    }
    std::vector<TrajectoryPoint> out(n);
    for (int i = 0; i < n; ++i) out[i] = {samples[4 * i], samples[4 * i + 1], samples[4 * i + 2], samples[4 * i + 3]};
    return out;
  }

  int status() const { return status_; }                                  // nlopt-style stopping reason
  int rejection() const { return rejection_; }                            // MRS_TG_FIND_* of the last call
  double bacaTotalTime() const { return baca_total_time_; }               // initial_total_time_baca (:1048-1056)
  const std::vector<double>& segmentTimes() const { return segment_times_; }  // Trajectory::getSegmentTimes
  const std::vector<double>& coefficients() const { return coefficients_; }   // [S][4][10], ascending powers
  const std::string& lastError() const { return last_error_; }

private:
  mrs_tg_ctx* ctx_ = nullptr;
  mrs_tg_options opt_{};
  int32_t status_ = MRS_TG_STATUS_FAILURE;
  int32_t rejection_ = MRS_TG_FIND_ACCEPTED;
  double baca_total_time_ = 0.0;
  std::vector<double> segment_times_, coefficients_;
  std::string last_error_;
};

}  // namespace mrs_tg
