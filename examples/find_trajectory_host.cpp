// find_trajectory_host.cpp -- a C++17 host over include/mrs_tg.hpp: mrs_tg::TrajectoryGenerator::findTrajectory has the argument
// list of MrsTrajectoryGeneration::findTrajectory (/root/reference/src/mrs_trajectory_generation.cpp:857-859) and, since ABI 5,
// its WHOLE behaviour: nullopt exactly where the reference returns {} -- a rejected optimiser code (:1146-1149) or a sampled
// trajectory that fails the temporal sanity check against the Baca estimate (:1178-1199).
//
//   find_trajectory_host WAYPOINTS.txt [max_len_factor] [min_len_factor]      (one "x y z heading" line per waypoint)
// prints one JSON line: {"found": 0|1, "samples": n, "status": code, "rejection": MRS_TG_FIND_*, "baca": s, "message": "..."}
#include <cstdio>
#include <cstdlib>
#include <optional>
#include <string>
#include <vector>

#include "mrs_tg.hpp"

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s WAYPOINTS.txt [max_len_factor] [min_len_factor]\n", argv[0]);
    return 2;
  }
  std::vector<mrs_tg::Waypoint> wps;
  if (FILE* f = std::fopen(argv[1], "r")) {
    double x, y, z, h;
    while (std::fscanf(f, "%lf %lf %lf %lf", &x, &y, &z, &h) == 4) wps.push_back({{x, y, z, h}, false});
    std::fclose(f);
  }
  if (wps.size() < 2) {
    std::fprintf(stderr, "need at least two waypoints\n");
    return 2;
  }
  mrs_tg::TrajectoryGenerator tg(0);
  tg.options().derivative_to_optimize = 4;   // the BASELINE configs' objective (the nodelet's default is 2)
  if (argc > 2) tg.options().max_trajectory_len_factor = std::atof(argv[2]);
  if (argc > 3) tg.options().min_trajectory_len_factor = std::atof(argv[3]);
  // the limits of the synthetic configs: 2 m/s, 2 m/s^2, 20 m/s^3; heading 1 rad/s, 2 rad/s^2, 20 rad/s^3
  const mrs_tg::DynamicsConstraints dc{2.0, 2.0, 20.0, 2.0, 2.0, 2.0, 2.0, 20.0, 20.0, 1.0, 2.0, 20.0};
  const auto pts = tg.findTrajectory(wps, std::nullopt, dc, 0.2, false, 4096);
  std::string msg = tg.lastError();
  for (char& c : msg)
    if (c == '"' || c == '\\') c = '\'';
  std::printf("{\"found\": %d, \"samples\": %zu, \"status\": %d, \"rejection\": %d, \"baca\": %.17g, \"message\": \"%s\"}\n", pts ? 1 : 0,
              pts ? pts->size() : (size_t)0, tg.status(), tg.rejection(), tg.bacaTotalTime(), pts ? "" : msg.c_str());
  return 0;
}
