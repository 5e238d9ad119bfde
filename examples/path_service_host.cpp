// path_service_host.cpp -- a C++ host over the C ABI, shaped like the reference's service callbacks: requests in,
// TrajectoryReference out (include/mrs_tg_service.hpp).  Prints one JSON object; tests/test_gpu_cpp_host.py builds
// it with g++ (no HIP, no torch: only libmrs_tg.so) and checks the output the way the reference's rostests do.
//
//   g++ -std=c++17 -I include examples/path_service_host.cpp -o path_service_host
//     -L mrs_uav_trajectory_generation_amd -lmrs_tg -Wl,-rpath,$PWD/mrs_uav_trajectory_generation_amd
#include <chrono>
#include <cmath>
#include <cstdio>

#include "mrs_tg_service.hpp"

using namespace mrs_tg;

static Path test_path() {  // the path of the reference's tests (test/get_path_before_takeoff/test.cpp:29-32)
  Path p;
  p.frame_id = "uav1/world_origin";
  p.input_id = 7;
  p.use_heading = true;
  p.fly_now = true;
  p.points = {{-5, -5, 5, 1}, {-5, 5, 5, 2}, {5, -5, 5, 3}, {5, 5, 5, 4}};
  return p;
}

static void print_response(const char* name, const GetPathResponse& r, bool last) {
  printf("\"%s\": {\"success\": %s, \"message\": \"%s\", \"dt\": %.3f, \"fly_now\": %s, \"use_heading\": %s, \"loop\": %s, "
         "\"input_id\": %llu, \"frame_id\": \"%s\", \"max_deviation\": %.6f, \"idxs\": [",
         name, r.success ? "true" : "false", r.message.c_str(), r.trajectory.dt, r.trajectory.fly_now ? "true" : "false",
         r.trajectory.use_heading ? "true" : "false", r.trajectory.loop ? "true" : "false",
         (unsigned long long)r.trajectory.input_id, r.trajectory.frame_id.c_str(), r.max_deviation);
  for (size_t i = 0; i < r.waypoint_trajectory_idxs.size(); ++i) printf("%s%d", i ? ", " : "", r.waypoint_trajectory_idxs[i]);
  printf("], \"points\": [");
  for (size_t i = 0; i < r.trajectory.points.size(); ++i) {
    const Reference& q = r.trajectory.points[i];
    printf("%s[%.9g, %.9g, %.9g, %.9g]", i ? ", " : "", q.x, q.y, q.z, q.heading);
  }
  printf("]}%s\n", last ? "" : ",");
}

int main() {
  PathService srv(0);
  printf("{\n");
  // 1. no constraints yet (:1976-1984)
  print_response("missing_constraints", srv.getPath(test_path()), false);

  Constraints c;
  c.horizontal_speed = 2.0;
  c.horizontal_acceleration = 2.0;
  c.horizontal_jerk = 20.0;
  c.vertical_ascending_speed = 2.0;
  c.vertical_descending_speed = 2.0;
  c.vertical_ascending_acceleration = 2.0;
  c.vertical_descending_acceleration = 2.0;
  c.vertical_ascending_jerk = 20.0;
  c.vertical_descending_jerk = 20.0;
  c.heading_speed = 1.0;
  c.heading_acceleration = 2.0;
  c.heading_jerk = 20.0;
  srv.setConstraints(c);
  CurrentState now;
  now.position = {0.0, 0.0, 3.0, 0.5};
  srv.setCurrentState(now);

  // 2. a batch of requests in one call: the reference's path; the same as a loop with stops at the waypoints; an empty
  //    message; a NaN; a faster user override of the limits with a tighter deviation bound
  Path plain = test_path();
  Path loop = test_path();
  loop.loop = true;
  loop.stop_at_waypoints = true;
  loop.input_id = 8;
  Path empty;
  Path nan = test_path();
  nan.points[2].y = std::nan("");
  Path fast = test_path();
  fast.override_constraints = true;
  fast.override_max_velocity_horizontal = 4.0;
  fast.override_max_velocity_vertical = 2.0;
  fast.override_max_acceleration_horizontal = 3.0;
  fast.override_max_acceleration_vertical = 2.0;
  fast.override_max_jerk_horizontal = 30.0;
  fast.override_max_jerk_vertical = 30.0;
  fast.max_deviation_from_path = 0.2;
  fast.input_id = 9;
  const auto res = srv.getPaths({plain, loop, empty, nan, fast});
  print_response("plain", res[0], false);
  print_response("loop_stop", res[1], false);
  print_response("empty", res[2], false);
  print_response("nan", res[3], false);
  print_response("override", res[4], false);

  // 3. without a current state the path is solved as given and "fly now" is dropped (:671-674)
  srv.clearCurrentState();
  print_response("no_state", srv.getPath(test_path()), false);

  // 4. the fallback sampler alone (enforce_fallback_solver / last attempt)
  srv.params().policy.fallback_sampling = 1;
  const GetPathResponse fb = srv.getPath(test_path());
  print_response("fallback", fb, false);
  srv.params().policy.fallback_sampling = 0;

  // 5. a request whose budget is spent before it starts: optimize() runs the fallback sampler for it ("executing fallback
  //    sampling, we are running over time", :711-713) -- the request succeeds with the fallback trajectory
  Path late = test_path();
  late.max_execution_time = 1e-6;
  const GetPathResponse lr = srv.getPath(late);
  print_response("late_request", lr, false);
  bool same = lr.trajectory.points.size() == fb.trajectory.points.size();
  for (size_t i = 0; same && i < fb.trajectory.points.size(); ++i)
    same = lr.trajectory.points[i].x == fb.trajectory.points[i].x && lr.trajectory.points[i].heading == fb.trajectory.points[i].heading;
  printf("\"late_equals_fallback\": %s,\n", same ? "true" : "false");

  // 6. many requests in one call with the default parameters (max_time 0.5 s per request, three deviation groups): every
  //    request is answered, by the solver or -- if its group ran late -- by the fallback sampler
  std::vector<Path> many;
  unsigned long long lcg = 12345;
  auto uni = [&]() {
    lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL;
    return (double)(lcg >> 11) / 9007199254740992.0;
  };
  for (int r = 0; r < 600; ++r) {
    Path p;
    p.frame_id = "uav1/world_origin";
    p.input_id = 100 + r;
    p.use_heading = true;
    p.dont_prepend_current_state = true;
    double x = 0, y = 0, bearing = 0;
    const int n = 4 + r % 9;
    for (int i = 0; i < n; ++i) {
      bearing += -0.4 + 0.8 * uni();
      const double step = 0.5 + 1.5 * uni();
      x += step * std::cos(bearing);
      y += step * std::sin(bearing);
      p.points.push_back({x, y, 5.0 + 0.2 * uni() - 0.1, bearing});
    }
    p.max_deviation_from_path = (r % 3 == 0) ? 0.0 : (r % 3 == 1 ? 0.1 : 0.3);
    many.push_back(p);
  }
  const auto t0 = std::chrono::steady_clock::now();
  const auto mres = srv.getPaths(many);
  const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  int n_ok = 0, n_id = 0;
  for (size_t r = 0; r < mres.size(); ++r) {
    n_ok += mres[r].success ? 1 : 0;
    n_id += (mres[r].trajectory.input_id == 100 + r && mres[r].trajectory.points.size() > 2) ? 1 : 0;
  }
  printf("\"many\": {\"requests\": %d, \"success\": %d, \"own_fields\": %d, \"elapsed_s\": %.4f}\n", (int)mres.size(), n_ok, n_id, el);
  printf("}\n");
  return 0;
}
