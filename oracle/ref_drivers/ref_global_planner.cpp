// Driver for oracle/_ref: the two global_planner sources that need nothing but the standard library --
// QuadraticCalculator (src/quadratic_calculator.cpp) with its PotentialCalculator base, and GridPath (src/grid_path.cpp) --
// compiled in place from /root/reference (nothing copied) behind a C ABI, so tests can pin the matching pieces of
// oracle/global_planner_oracle.hpp against the reference itself.  dijkstra.cpp, astar.cpp and gradient_path.cpp include
// planner_core.h (ros/ros.h, nav_core, dynamic_reconfigure): unbuildable here, no stand-in headers are written.
#include <algorithm>  // potential_calculator.h uses std::min without including <algorithm>
#include <utility>
#include <vector>

#include <global_planner/grid_path.h>
#include <global_planner/potential_calculator.h>
#include <global_planner/quadratic_calculator.h>

extern "C" {
// calculatePotential for every listed cell n (prev_potential < 0: the calculators look at the neighbours themselves)
void ref_gp_calculate_potential(int quadratic, float* potential, int nx, int ny, const unsigned char* cost, const int* cells, const float* prev,
                                int count, float* out) {
  global_planner::PotentialCalculator plain(nx, ny);
  global_planner::QuadraticCalculator quad(nx, ny);
  global_planner::PotentialCalculator* c = quadratic ? &quad : &plain;
  for (int i = 0; i < count; ++i) out[i] = c->calculatePotential(potential, cost[i], cells[i], prev[i]);
}
// GridPath::getPath; returns the number of points written (0: getPath returned false)
int ref_gp_grid_path(float* potential, int nx, int ny, double start_x, double start_y, double end_x, double end_y, float* path_xy, int cap) {
  global_planner::PotentialCalculator plain(nx, ny);
  global_planner::GridPath gp(&plain);
  gp.setSize(nx, ny);
  std::vector<std::pair<float, float> > path;
  if (!gp.getPath(potential, start_x, start_y, end_x, end_y, path)) return 0;
  const int n = (int)path.size();
  for (int i = 0; i < n && i < cap; ++i) {
    path_xy[2 * i] = path[i].first;
    path_xy[2 * i + 1] = path[i].second;
  }
  return n;
}
}
