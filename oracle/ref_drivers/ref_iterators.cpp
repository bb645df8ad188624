// Driver for oracle/_ref: exposes the REFERENCE's own header-only LineIterator and
// VelocityIterator (compiled in place from /root/reference, nothing copied) through a C ABI so
// tests can pin oracle/planner_oracle.hpp against them.  These two headers need nothing but the
// standard library; every other reference file on the path needs ROS/boost/Eigen/pcl headers
// that this image lacks, so nothing else is built (no stand-in headers are written).
#include <vector>  // velocity_iterator.h uses std::vector without including <vector>

#include <base_local_planner/line_iterator.h>
#include <base_local_planner/velocity_iterator.h>
#include <costmap_2d/cost_values.h>

extern "C" {
int ref_line_cells(int x0, int y0, int x1, int y1, int* out_xy, int cap) {
  int n = 0;
  for (base_local_planner::LineIterator line(x0, y0, x1, y1); line.isValid(); line.advance()) {
    if (n < cap) {
      out_xy[2 * n] = line.getX();
      out_xy[2 * n + 1] = line.getY();
    }
    ++n;
  }
  return n;
}
int ref_velocity_samples(double mn, double mx, int num, double* out, int cap) {
  int n = 0;
  for (base_local_planner::VelocityIterator it(mn, mx, num); !it.isFinished(); it++) {
    if (n < cap) out[n] = it.getVelocity();
    ++n;
  }
  return n;
}
void ref_cost_values(unsigned char* out4) {
  out4[0] = costmap_2d::NO_INFORMATION;
  out4[1] = costmap_2d::LETHAL_OBSTACLE;
  out4[2] = costmap_2d::INSCRIBED_INFLATED_OBSTACLE;
  out4[3] = costmap_2d::FREE_SPACE;
}
}
