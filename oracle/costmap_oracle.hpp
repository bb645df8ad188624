// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of the costmap half of the hot path of BadgerTechnologies/navigation
// (costmap_2d + voxel_grid).  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use anything in oracle/.  Every function cites the reference
// file:line it restates (paths relative to /root/reference).
//
// Parity pinning: the reference needs ROS/boost/Eigen/pcl headers that this image
// lacks, so it is NOT compiled here (no stand-in headers are written).  The
// restatement is pinned by the reference's own test expectations
// (tests/test_oracle_reference_fixtures.py) and, for the two header-only pieces
// that compile from their own files (LineIterator, VelocityIterator), by
// oracle/_ref built from the reference sources in place.
#pragma once
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <queue>
#include <vector>

namespace oracle {

// costmap_2d/include/costmap_2d/cost_values.h:42-45
constexpr uint8_t NO_INFORMATION = 255;
constexpr uint8_t LETHAL_OBSTACLE = 254;
constexpr uint8_t INSCRIBED_INFLATED_OBSTACLE = 253;
constexpr uint8_t FREE_SPACE = 0;

struct Pt2 {
  double x = 0, y = 0;
};
struct Pt3f {
  float x = 0, y = 0, z = 0;
};
struct CellXY {
  uint32_t x = 0, y = 0;
};

// costmap_2d/src/costmap_math.cpp:32-60 + include/costmap_2d/costmap_math.h (distance = hypot)
inline double dist2d(double x0, double y0, double x1, double y1) { return hypot(x1 - x0, y1 - y0); }
inline double distanceToLine(double pX, double pY, double x0, double y0, double x1, double y1) {
  double A = pX - x0, B = pY - y0, C = x1 - x0, D = y1 - y0;
  double dot = A * C + B * D;
  double len_sq = C * C + D * D;
  double param = dot / len_sq;
  double xx, yy;
  if (param < 0) {
    xx = x0;
    yy = y0;
  } else if (param > 1) {
    xx = x1;
    yy = y1;
  } else {
    xx = x0 + param * C;
    yy = y0 + param * D;
  }
  return dist2d(pX, pY, xx, yy);
}

// costmap_2d/src/footprint.cpp:41-67
inline void calculateMinAndMaxDistances(const std::vector<Pt2>& fp, double& min_dist, double& max_dist) {
  min_dist = std::numeric_limits<double>::max();
  max_dist = 0.0;
  if (fp.size() <= 2) return;
  for (size_t i = 0; i + 1 < fp.size(); ++i) {
    double vd = dist2d(0.0, 0.0, fp[i].x, fp[i].y);
    double ed = distanceToLine(0.0, 0.0, fp[i].x, fp[i].y, fp[i + 1].x, fp[i + 1].y);
    min_dist = std::min(min_dist, std::min(vd, ed));
    max_dist = std::max(max_dist, std::max(vd, ed));
  }
  double vd = dist2d(0.0, 0.0, fp.back().x, fp.back().y);
  double ed = distanceToLine(0.0, 0.0, fp.back().x, fp.back().y, fp.front().x, fp.front().y);
  min_dist = std::min(min_dist, std::min(vd, ed));
  max_dist = std::max(max_dist, std::max(vd, ed));
}

// costmap_2d/src/footprint.cpp:103-118
inline void transformFootprint(double x, double y, double theta, const std::vector<Pt2>& spec, std::vector<Pt2>& out) {
  out.clear();
  double cos_th = cos(theta), sin_th = sin(theta);
  for (const Pt2& p : spec) {
    Pt2 q;
    q.x = x + (p.x * cos_th - p.y * sin_th);
    q.y = y + (p.x * sin_th + p.y * cos_th);
    out.push_back(q);
  }
}

// ---------------------------------------------------------------------------------------------
// Grid2D restates costmap_2d::Costmap2D (include/costmap_2d/costmap_2d.h:60-466, src/costmap_2d.cpp)
// ---------------------------------------------------------------------------------------------
struct Grid2D {
  uint32_t size_x = 0, size_y = 0;
  double resolution = 0, origin_x = 0, origin_y = 0;
  uint8_t default_value = 0;
  std::vector<uint8_t> cells;

  void resize(uint32_t sx, uint32_t sy, double res, double ox, double oy) {  // costmap_2d.cpp:73-91
    size_x = sx;
    size_y = sy;
    resolution = res;
    origin_x = ox;
    origin_y = oy;
    cells.assign(size_t(sx) * sy, default_value);
  }
  void resetMaps() { std::fill(cells.begin(), cells.end(), default_value); }  // costmap_2d.cpp:87-91
  // costmap_2d.cpp:93-99 (row-wise memset of [x0,xn) x [y0,yn))
  void resetMap(uint32_t x0, uint32_t y0, uint32_t xn, uint32_t yn) {
    uint32_t len = xn - x0;
    for (uint32_t y = y0 * size_x + x0; y < yn * size_x + x0; y += size_x) memset(cells.data() + y, default_value, len);
  }
  inline uint32_t index(uint32_t mx, uint32_t my) const { return my * size_x + mx; }  // costmap_2d.h:171-174
  inline uint8_t cost(uint32_t mx, uint32_t my) const { return cells[index(mx, my)]; }
  void mapToWorld(uint32_t mx, uint32_t my, double& wx, double& wy) const {  // costmap_2d.cpp:202-206
    wx = origin_x + (mx + 0.5) * resolution;
    wy = origin_y + (my + 0.5) * resolution;
  }
  bool worldToMap(double wx, double wy, uint32_t& mx, uint32_t& my) const {  // costmap_2d.cpp:208-220
    if (wx < origin_x || wy < origin_y) return false;
    mx = (int)((wx - origin_x) / resolution);
    my = (int)((wy - origin_y) / resolution);
    return mx < size_x && my < size_y;
  }
  void worldToMapEnforceBounds(double wx, double wy, int& mx, int& my) const {  // costmap_2d.cpp:228-262
    if (wx < origin_x)
      mx = 0;
    else if (wx > resolution * (size_x - 1) + origin_x)
      mx = size_x - 1;
    else
      mx = (int)((wx - origin_x) / resolution);
    if (wy < origin_y)
      my = 0;
    else if (wy > resolution * (size_y - 1) + origin_y)
      my = size_y - 1;
    else
      my = (int)((wy - origin_y) / resolution);
  }
  uint32_t cellDistance(double world_dist) const {  // costmap_2d.cpp:181-185
    double cells_dist = std::max(0.0, ceil(world_dist / resolution));
    return (uint32_t)cells_dist;
  }
  double sizeInMetersX() const { return (size_x - 1 + 0.5) * resolution; }  // costmap_2d.cpp:440-443
  double sizeInMetersY() const { return (size_y - 1 + 0.5) * resolution; }  // costmap_2d.cpp:445-448

  // costmap_2d.cpp:264-313: updateOrigin (rolling window): keep the overlap, default everywhere else.
  // Returns the cell offsets so that sibling arrays (voxel columns) can be moved the same way.
  void updateOrigin(double new_origin_x, double new_origin_y, int* out_cell_ox = nullptr, int* out_cell_oy = nullptr) {
    int cell_ox = int((new_origin_x - origin_x) / resolution);
    int cell_oy = int((new_origin_y - origin_y) / resolution);
    double new_grid_ox = origin_x + cell_ox * resolution;
    double new_grid_oy = origin_y + cell_oy * resolution;
    int sx = size_x, sy = size_y;
    int lower_left_x = std::min(std::max(cell_ox, 0), sx), lower_left_y = std::min(std::max(cell_oy, 0), sy);
    int upper_right_x = std::min(std::max(cell_ox + sx, 0), sx), upper_right_y = std::min(std::max(cell_oy + sy, 0), sy);
    uint32_t cell_size_x = upper_right_x - lower_left_x, cell_size_y = upper_right_y - lower_left_y;
    std::vector<uint8_t> local((size_t)cell_size_x * cell_size_y);
    for (uint32_t i = 0; i < cell_size_y; ++i)  // copyMapRegion, costmap_2d.h:315-331
      memcpy(local.data() + (size_t)i * cell_size_x, cells.data() + (size_t)(lower_left_y + i) * size_x + lower_left_x, cell_size_x);
    resetMaps();
    origin_x = new_grid_ox;
    origin_y = new_grid_oy;
    int start_x = lower_left_x - cell_ox, start_y = lower_left_y - cell_oy;
    for (uint32_t i = 0; i < cell_size_y; ++i)
      memcpy(cells.data() + (size_t)(start_y + i) * size_x + start_x, local.data() + (size_t)i * cell_size_x, cell_size_x);
    if (out_cell_ox) *out_cell_ox = cell_ox;
    if (out_cell_oy) *out_cell_oy = cell_oy;
  }

  // costmap_2d.h:359-417: raytraceLine + bresenham2D; `at(offset)` is applied to every visited cell.
  template <class F>
  void raytraceLine(F&& at, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t max_length = UINT_MAX) const {
    int dx = x1 - x0, dy = y1 - y0;
    uint32_t abs_dx = abs(dx), abs_dy = abs(dy);
    int offset_dx = dx > 0 ? 1 : -1;                 // sign(0) == -1, costmap_2d.h:414-417
    int offset_dy = (dy > 0 ? 1 : -1) * (int)size_x;
    uint32_t offset = y0 * size_x + x0;
    double dist = hypot(dx, dy);
    double scale = (dist == 0.0) ? 1.0 : std::min(1.0, max_length / dist);
    if (abs_dx >= abs_dy)
      bresenham2D(at, abs_dx, abs_dy, abs_dx / 2, offset_dx, offset_dy, offset, (uint32_t)(scale * abs_dx));
    else
      bresenham2D(at, abs_dy, abs_dx, abs_dy / 2, offset_dy, offset_dx, offset, (uint32_t)(scale * abs_dy));
  }
  template <class F>
  static void bresenham2D(F&& at, uint32_t abs_da, uint32_t abs_db, int error_b, int offset_a, int offset_b, uint32_t offset,
                          uint32_t max_length) {
    uint32_t end = std::min(max_length, abs_da);
    for (uint32_t i = 0; i < end; ++i) {
      at(offset);
      offset += offset_a;
      error_b += abs_db;
      if ((uint32_t)error_b >= abs_da) {
        offset += offset_b;
        error_b -= abs_da;
      }
    }
    at(offset);
  }

  // costmap_2d.cpp:315-428: setConvexPolygonCost / polygonOutlineCells / convexFillCells,
  // including the bubble sort by x, the pairwise column walk on the live (growing) vector and the
  // `y < max_pt.y` fill that leaves the top cell to the outline.
  bool setConvexPolygonCost(const std::vector<Pt2>& polygon, uint8_t value) {
    std::vector<CellXY> mp;
    for (const Pt2& p : polygon) {
      CellXY c;
      if (!worldToMap(p.x, p.y, c.x, c.y)) return false;
      mp.push_back(c);
    }
    std::vector<CellXY> pc;
    convexFillCells(mp, pc);
    for (const CellXY& c : pc) cells[index(c.x, c.y)] = value;
    return true;
  }
  void polygonOutlineCells(const std::vector<CellXY>& poly, std::vector<CellXY>& out) const {
    auto gather = [&](uint32_t off) {
      CellXY c;
      c.y = off / size_x;  // indexToCells, costmap_2d.h:182-186
      c.x = off - c.y * size_x;
      out.push_back(c);
    };
    for (size_t i = 0; i + 1 < poly.size(); ++i) raytraceLine(gather, poly[i].x, poly[i].y, poly[i + 1].x, poly[i + 1].y);
    if (!poly.empty()) {
      size_t last = poly.size() - 1;
      raytraceLine(gather, poly[last].x, poly[last].y, poly[0].x, poly[0].y);
    }
  }
  void convexFillCells(const std::vector<CellXY>& poly, std::vector<CellXY>& pc) const {
    if (poly.size() < 3) return;
    polygonOutlineCells(poly, pc);
    size_t i = 0;
    while (i < pc.size() - 1) {
      if (pc[i].x > pc[i + 1].x) {
        std::swap(pc[i], pc[i + 1]);
        if (i > 0) --i;
      } else
        ++i;
    }
    i = 0;
    CellXY min_pt, max_pt;
    uint32_t min_x = pc[0].x, max_x = pc[pc.size() - 1].x;
    for (uint32_t x = min_x; x <= max_x; ++x) {
      if (i >= pc.size() - 1) break;
      if (pc[i].y < pc[i + 1].y) {
        min_pt = pc[i];
        max_pt = pc[i + 1];
      } else {
        min_pt = pc[i + 1];
        max_pt = pc[i];
      }
      i += 2;
      while (i < pc.size() && pc[i].x == x) {
        if (pc[i].y < min_pt.y)
          min_pt = pc[i];
        else if (pc[i].y > max_pt.y)
          max_pt = pc[i];
        ++i;
      }
      for (uint32_t y = min_pt.y; y < max_pt.y; ++y) {
        CellXY c;
        c.x = x;
        c.y = y;
        pc.push_back(c);
      }
    }
  }
};

struct Bounds {
  double min_x, min_y, max_x, max_y;
  void touch(double x, double y) {  // costmap_layer.cpp touch()
    min_x = std::min(x, min_x);
    min_y = std::min(y, min_y);
    max_x = std::max(x, max_x);
    max_y = std::max(y, max_y);
  }
};

// costmap_2d/include/costmap_2d/observation.h:46-103: origin (double xyz), float xyz cloud, ranges.
struct Observation {
  double ox = 0, oy = 0, oz = 0;
  std::vector<Pt3f> cloud;
  double obstacle_range = 2.5, raytrace_range = 3.0;
};

// costmap_2d/src/costmap_layer.cpp:62-124 merge helpers (layer grid -> master, same geometry)
inline void updateWithMax(const Grid2D& layer, Grid2D& master, int min_i, int min_j, int max_i, int max_j) {
  uint32_t span = master.size_x;
  for (int j = min_j; j < max_j; j++) {
    uint32_t it = j * span + min_i;
    for (int i = min_i; i < max_i; i++, it++) {
      uint8_t c = layer.cells[it];
      if (c == NO_INFORMATION) continue;
      uint8_t old_cost = master.cells[it];
      if (old_cost == NO_INFORMATION || old_cost < c) master.cells[it] = c;
    }
  }
}
inline void updateWithTrueOverwrite(const Grid2D& layer, Grid2D& master, int min_i, int min_j, int max_i, int max_j) {
  uint32_t span = master.size_x;
  for (int j = min_j; j < max_j; j++) {
    uint32_t it = span * j + min_i;
    for (int i = min_i; i < max_i; i++, it++) master.cells[it] = layer.cells[it];
  }
}
inline void updateWithOverwrite(const Grid2D& layer, Grid2D& master, int min_i, int min_j, int max_i, int max_j) {
  uint32_t span = master.size_x;
  for (int j = min_j; j < max_j; j++) {
    uint32_t it = span * j + min_i;
    for (int i = min_i; i < max_i; i++, it++)
      if (layer.cells[it] != NO_INFORMATION) master.cells[it] = layer.cells[it];
  }
}

// ---------------------------------------------------------------------------------------------
// Static layer, non-rolling branch: costmap_2d/plugins/static_layer.cpp:149-163 (interpretValue),
// :167-228 (incomingMap), :263-283 (updateBounds), :285-299 (updateCosts).
// ---------------------------------------------------------------------------------------------
struct StaticLayerOracle {
  Grid2D grid;
  bool map_received = false, has_updated_data = false, use_maximum = false;
  bool track_unknown_space = true, trinary_costmap = true;
  uint8_t lethal_threshold = 100, unknown_cost_value = (uint8_t)-1;
  uint32_t x_ = 0, y_ = 0, width_ = 0, height_ = 0;

  uint8_t interpretValue(uint8_t value) const {
    if (track_unknown_space && value == unknown_cost_value)
      return NO_INFORMATION;
    else if (!track_unknown_space && value == unknown_cost_value)
      return FREE_SPACE;
    else if (value >= lethal_threshold)
      return LETHAL_OBSTACLE;
    else if (trinary_costmap)
      return FREE_SPACE;
    double scale = (double)value / lethal_threshold;
    return scale * LETHAL_OBSTACLE;
  }
  // occupancy: int8 row-major (nav_msgs/OccupancyGrid.data), geometry taken from the master
  void incomingMap(const int8_t* occ, uint32_t sx, uint32_t sy, double res, double ox, double oy) {
    grid.default_value = 0;
    grid.resize(sx, sy, res, ox, oy);
    for (size_t i = 0; i < size_t(sx) * sy; ++i) grid.cells[i] = interpretValue((uint8_t)occ[i]);
    x_ = y_ = 0;
    width_ = sx;
    height_ = sy;
    map_received = true;
    has_updated_data = true;
  }
  // rolling window (static_layer.cpp:265-268: the early return is for non-rolling costmaps only): the layer's own
  // extent joins the bounds every cycle, so LayeredCostmap::updateMap rewrites the whole window
  bool rolling = false;
  double tf_basis[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, tf_origin[3] = {0, 0, 0};  // map_frame <- global_frame (tf::StampedTransform)
  void updateBounds(Bounds& b) {
    if (!rolling && (!map_received || !has_updated_data)) return;
    double wx, wy;
    grid.mapToWorld(x_, y_, wx, wy);
    b.min_x = std::min(wx, b.min_x);
    b.min_y = std::min(wy, b.min_y);
    grid.mapToWorld(x_ + width_, y_ + height_, wx, wy);
    b.max_x = std::max(wx, b.max_x);
    b.max_y = std::max(wy, b.max_y);
    has_updated_data = false;
  }
  void updateCosts(Grid2D& master, int min_i, int min_j, int max_i, int max_j) const {
    if (!map_received) return;
    if (!rolling) {
      if (!use_maximum)
        updateWithTrueOverwrite(grid, master, min_i, min_j, max_i, max_j);
      else
        updateWithMax(grid, master, min_i, min_j, max_i, max_j);
      return;
    }
    // rolling branch (static_layer.cpp:300-333): every master cell of the window -> world -> map frame -> static cell.
    // tf::Transform::operator() (LinearMath, not in the reference tree; its published form): basis row . point + origin
    for (unsigned int i = min_i; i < (unsigned int)max_i; ++i) {
      for (unsigned int j = min_j; j < (unsigned int)max_j; ++j) {
        double wx, wy;
        master.mapToWorld(i, j, wx, wy);
        const double px = tf_basis[0] * wx + tf_basis[1] * wy + tf_basis[2] * 0.0 + tf_origin[0];
        const double py = tf_basis[3] * wx + tf_basis[4] * wy + tf_basis[5] * 0.0 + tf_origin[1];
        unsigned int mx, my;
        if (grid.worldToMap(px, py, mx, my)) {
          if (!use_maximum)
            master.cells[master.index(i, j)] = grid.cost(mx, my);
          else
            master.cells[master.index(i, j)] = std::max(grid.cost(mx, my), master.cost(i, j));
        }
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------
// Obstacle layer: costmap_2d/plugins/obstacle_layer.cpp:340-448 (updateBounds/updateFootprint/
// updateCosts), :498-576 (raytraceFreespace), :602-610 (updateRaytraceBounds).
// Observations are injected like the reference tests do (addStaticObservation, :450-464).
// ---------------------------------------------------------------------------------------------
struct ObstacleLayerOracle {
  Grid2D grid;  // the layer's own costmap (CostmapLayer : Layer, Costmap2D)
  // CostmapLayer's extra bounds (costmap_layer.cpp:21-60): a box added by resetBoundingBox, taken up by the next updateBounds
  double extra_min_x = 1e6, extra_max_x = -1e6, extra_min_y = 1e6, extra_max_y = -1e6;
  bool has_extra_bounds = false;
  void addExtraBounds(double mx0, double my0, double mx1, double my1) {
    extra_min_x = std::min(mx0, extra_min_x);
    extra_max_x = std::max(mx1, extra_max_x);
    extra_min_y = std::min(my0, extra_min_y);
    extra_max_y = std::max(my1, extra_max_y);
    has_extra_bounds = true;
  }
  void resetBoundingBox(double min_x, double min_y, double max_x, double max_y) {  // costmap_layer.cpp:30-43 (the 2-D grid only, also for a VoxelLayer)
    int start_x, start_y, end_x, end_y;
    grid.worldToMapEnforceBounds(min_x, min_y, start_x, start_y);
    grid.worldToMapEnforceBounds(max_x, max_y, end_x, end_y);
    grid.resetMap(start_x, start_y, end_x, end_y);
    addExtraBounds(min_x, min_y, max_x, max_y);
  }
  void useExtraBounds(Bounds& b) {
    if (!has_extra_bounds) return;
    b.min_x = std::min(extra_min_x, b.min_x);
    b.min_y = std::min(extra_min_y, b.min_y);
    b.max_x = std::max(extra_max_x, b.max_x);
    b.max_y = std::max(extra_max_y, b.max_y);
    extra_min_x = 1e6;
    extra_min_y = 1e6;
    extra_max_x = -1e6;
    extra_max_y = -1e6;
    has_extra_bounds = false;
  }
  bool enabled = true, footprint_clearing_enabled = true;
  double max_obstacle_height = 2.0;
  int combination_method = 1;
  std::vector<Pt2> footprint_spec, transformed_footprint;

  void matchSize(const Grid2D& master, bool track_unknown) {  // obstacle_layer.cpp:60-66, costmap_layer.cpp matchSize
    grid.default_value = track_unknown ? NO_INFORMATION : FREE_SPACE;
    grid.resize(master.size_x, master.size_y, master.resolution, master.origin_x, master.origin_y);
  }

  void raytraceFreespace(const Observation& obs, Bounds& b) {
    double ox = obs.ox, oy = obs.oy;
    uint32_t x0, y0;
    if (!grid.worldToMap(ox, oy, x0, y0)) return;
    double origin_x = grid.origin_x, origin_y = grid.origin_y;
    double map_end_x = origin_x + grid.size_x * grid.resolution;
    double map_end_y = origin_y + grid.size_y * grid.resolution;
    b.touch(ox, oy);
    for (const Pt3f& p : obs.cloud) {
      double wx = p.x, wy = p.y;
      double a = wx - ox, bb = wy - oy;
      if (wx < origin_x) {
        double t = (origin_x - ox) / a;
        wx = origin_x;
        wy = oy + bb * t;
      }
      if (wy < origin_y) {
        double t = (origin_y - oy) / bb;
        wx = ox + a * t;
        wy = origin_y;
      }
      if (wx > map_end_x) {
        double t = (map_end_x - ox) / a;
        wx = map_end_x - .001;
        wy = oy + bb * t;
      }
      if (wy > map_end_y) {
        double t = (map_end_y - oy) / bb;
        wx = ox + a * t;
        wy = map_end_y - .001;
      }
      uint32_t x1, y1;
      if (!grid.worldToMap(wx, wy, x1, y1)) continue;
      uint32_t cell_raytrace_range = grid.cellDistance(obs.raytrace_range);
      uint8_t* cm = grid.cells.data();
      grid.raytraceLine([cm](uint32_t off) { cm[off] = FREE_SPACE; }, x0, y0, x1, y1, cell_raytrace_range);
      // updateRaytraceBounds, obstacle_layer.cpp:602-610
      double dx = wx - ox, dy = wy - oy;
      double full_distance = hypot(dx, dy);
      double scale = std::min(1.0, obs.raytrace_range / full_distance);
      b.touch(ox + dx * scale, oy + dy * scale);
    }
  }

  void updateBounds(double rx, double ry, double ryaw, const std::vector<Observation>& marking,
                    const std::vector<Observation>& clearing, Bounds& b) {
    if (!enabled) return;
    useExtraBounds(b);  // obstacle_layer.cpp:347
    for (const Observation& o : clearing) raytraceFreespace(o, b);
    for (const Observation& obs : marking) {
      double sq_obstacle_range = obs.obstacle_range * obs.obstacle_range;
      for (const Pt3f& p : obs.cloud) {
        double px = p.x, py = p.y, pz = p.z;
        if (pz > max_obstacle_height) continue;
        double sq_dist = (px - obs.ox) * (px - obs.ox) + (py - obs.oy) * (py - obs.oy) + (pz - obs.oz) * (pz - obs.oz);
        if (sq_dist >= sq_obstacle_range) continue;
        uint32_t mx, my;
        if (!grid.worldToMap(px, py, mx, my)) continue;
        grid.cells[grid.index(mx, my)] = LETHAL_OBSTACLE;
        b.touch(px, py);
      }
    }
    updateFootprint(rx, ry, ryaw, b);
  }
  void updateFootprint(double rx, double ry, double ryaw, Bounds& b) {  // obstacle_layer.cpp:415-425
    if (!footprint_clearing_enabled) return;
    transformFootprint(rx, ry, ryaw, footprint_spec, transformed_footprint);
    for (const Pt2& p : transformed_footprint) b.touch(p.x, p.y);
  }
  void updateCosts(Grid2D& master, int min_i, int min_j, int max_i, int max_j) {  // obstacle_layer.cpp:427-448
    if (!enabled) return;
    if (footprint_clearing_enabled) grid.setConvexPolygonCost(transformed_footprint, FREE_SPACE);
    switch (combination_method) {
      case 0: updateWithOverwrite(grid, master, min_i, min_j, max_i, max_j); break;
      case 1: updateWithMax(grid, master, min_i, min_j, max_i, max_j); break;
      default: break;
    }
  }
};

// ---------------------------------------------------------------------------------------------
// VoxelGrid: voxel_grid/include/voxel_grid/voxel_grid.h:65-434, src/voxel_grid.cpp
// ---------------------------------------------------------------------------------------------
struct VoxelGridOracle {
  uint32_t size_x = 0, size_y = 0, size_z = 0;
  std::vector<uint32_t> data;
  void resize(uint32_t sx, uint32_t sy, uint32_t sz) {  // voxel_grid.cpp:41-82
    size_x = sx;
    size_y = sy;
    size_z = std::min<uint32_t>(sz, 16);
    data.assign(size_t(sx) * sy, ~((uint32_t)0) >> 16);
  }
  void reset() { std::fill(data.begin(), data.end(), ~((uint32_t)0) >> 16); }
  static bool bitsBelowThreshold(uint32_t n, uint32_t thr) {  // voxel_grid.h:151-164
    uint32_t bit_count;
    for (bit_count = 0; n;) {
      ++bit_count;
      if (bit_count > thr) return false;
      n &= n - 1;
    }
    return true;
  }
  static uint32_t numBits(uint32_t n) {
    uint32_t c = 0;
    for (; n; ++c) n &= n - 1;
    return c;
  }
  void markVoxel(uint32_t x, uint32_t y, uint32_t z) {  // voxel_grid.h:90-98
    if (x >= size_x || y >= size_y || z >= size_z) return;
    uint32_t full_mask = ((uint32_t)1 << z << 16) | (1 << z);
    data[y * size_x + x] |= full_mask;
  }
  bool markVoxelInMap(uint32_t x, uint32_t y, uint32_t z, uint32_t marked_threshold) {  // voxel_grid.h:100-117
    if (x >= size_x || y >= size_y || z >= size_z) return false;
    uint32_t& col = data[y * size_x + x];
    uint32_t full_mask = ((uint32_t)1 << z << 16) | (1 << z);
    col |= full_mask;
    uint32_t marked_bits = col >> 16;
    return !bitsBelowThreshold(marked_bits, marked_threshold);
  }
  int getVoxel(uint32_t x, uint32_t y, uint32_t z) const {  // voxel_grid.h:177-200: 0 free, 1 unknown, 2 marked
    if (x >= size_x || y >= size_y || z >= size_z) return 1;
    uint32_t full_mask = ((uint32_t)1 << z << 16) | (1 << z);
    uint32_t bits = numBits(data[y * size_x + x] & full_mask);
    return bits < 2 ? (bits < 1 ? 0 : 1) : 2;
  }
  // voxel_grid.h:226-308: raytraceLine + bresenham3D.  `at(offset, z_mask)`.
  template <class F>
  void raytraceLine(F&& at, double x0, double y0, double z0, double x1, double y1, double z1, uint32_t max_length = UINT_MAX) {
    int dx = int(x1) - int(x0), dy = int(y1) - int(y0), dz = int(z1) - int(z0);
    uint32_t abs_dx = abs(dx), abs_dy = abs(dy), abs_dz = abs(dz);
    int offset_dx = dx > 0 ? 1 : -1;
    int offset_dy = (dy > 0 ? 1 : -1) * (int)size_x;
    int offset_dz = dz > 0 ? 1 : -1;
    uint32_t z_mask = ((1 << 16) | 1) << (uint32_t)z0;
    uint32_t offset = (uint32_t)y0 * size_x + (uint32_t)x0;
    double dist = sqrt((x0 - x1) * (x0 - x1) + (y0 - y1) * (y0 - y1) + (z0 - z1) * (z0 - z1));
    double scale = std::min(1.0, max_length / dist);
    auto grid_off = [&](int v) { offset += v; };
    auto z_off = [&](int v) { v > 0 ? z_mask <<= 1 : z_mask >>= 1; };
    auto run = [&](auto&& off_a, auto&& off_b, auto&& off_c, uint32_t abs_da, uint32_t abs_db, uint32_t abs_dc, int offset_a,
                   int offset_b, int offset_c, uint32_t max_len) {
      int error_b = abs_da / 2, error_c = abs_da / 2;
      uint32_t end = std::min(max_len, abs_da);
      for (uint32_t i = 0; i < end; ++i) {
        at(offset, z_mask);
        off_a(offset_a);
        error_b += abs_db;
        error_c += abs_dc;
        if ((uint32_t)error_b >= abs_da) {
          off_b(offset_b);
          error_b -= abs_da;
        }
        if ((uint32_t)error_c >= abs_da) {
          off_c(offset_c);
          error_c -= abs_da;
        }
      }
      at(offset, z_mask);
    };
    if (abs_dx >= std::max(abs_dy, abs_dz)) {
      run(grid_off, grid_off, z_off, abs_dx, abs_dy, abs_dz, offset_dx, offset_dy, offset_dz, (uint32_t)(scale * abs_dx));
      return;
    }
    if (abs_dy >= abs_dz) {
      run(grid_off, grid_off, z_off, abs_dy, abs_dx, abs_dz, offset_dy, offset_dx, offset_dz, (uint32_t)(scale * abs_dy));
      return;
    }
    run(z_off, grid_off, grid_off, abs_dz, abs_dx, abs_dy, offset_dz, offset_dx, offset_dy, (uint32_t)(scale * abs_dz));
  }
  bool endpointsOutOfBounds(double x0, double y0, double z0, double x1, double y1, double z1) const {
    return x0 >= size_x || y0 >= size_y || z0 >= size_z || x1 >= size_x || y1 >= size_y || z1 >= size_z;
  }
  void markVoxelLine(double x0, double y0, double z0, double x1, double y1, double z1, uint32_t max_length = UINT_MAX) {
    if (endpointsOutOfBounds(x0, y0, z0, x1, y1, z1)) return;  // voxel_grid.cpp:99-109
    uint32_t* d = data.data();
    raytraceLine([d](uint32_t off, uint32_t zm) { d[off] |= zm; }, x0, y0, z0, x1, y1, z1, max_length);
  }
  void clearVoxelLine(double x0, double y0, double z0, double x1, double y1, double z1, uint32_t max_length = UINT_MAX) {
    if (endpointsOutOfBounds(x0, y0, z0, x1, y1, z1)) return;  // voxel_grid.cpp:111-122
    uint32_t* d = data.data();
    raytraceLine([d](uint32_t off, uint32_t zm) { d[off] &= ~zm; }, x0, y0, z0, x1, y1, z1, max_length);
  }
  // voxel_grid.cpp:124-141 + ClearVoxelInMap functor voxel_grid.h:349-402
  void clearVoxelLineInMap(double x0, double y0, double z0, double x1, double y1, double z1, uint8_t* map_2d,
                           uint32_t unknown_threshold, uint32_t mark_threshold, uint8_t free_cost, uint8_t unknown_cost,
                           uint32_t max_length) {
    if (endpointsOutOfBounds(x0, y0, z0, x1, y1, z1)) return;
    uint32_t* d = data.data();
    raytraceLine(
        [=](uint32_t off, uint32_t zm) {
          uint32_t& col = d[off];
          col &= ~zm;
          uint32_t unknown_bits = uint16_t(col >> 16) ^ uint16_t(col);
          uint32_t marked_bits = col >> 16;
          if (bitsBelowThreshold(marked_bits, mark_threshold)) {
            if (bitsBelowThreshold(unknown_bits, unknown_threshold))
              map_2d[off] = free_cost;
            else
              map_2d[off] = unknown_cost;
          }
        },
        x0, y0, z0, x1, y1, z1, max_length);
  }
};

// ---------------------------------------------------------------------------------------------
// Voxel layer: costmap_2d/plugins/voxel_layer.cpp:78-91 (config), :116-213 (updateBounds),
// :266-383 (raytraceFreespace), include/costmap_2d/voxel_layer.h:107-133 (3-D conversions)
// ---------------------------------------------------------------------------------------------
struct VoxelLayerOracle : ObstacleLayerOracle {
  VoxelGridOracle vg;
  double z_resolution = 0.2, origin_z = 0.0;
  uint32_t unknown_threshold = 15, mark_threshold = 0, size_z = 10;

  void configure(uint32_t z_voxels, double oz, double zres, uint32_t unknown_thr_cfg, uint32_t mark_thr) {
    size_z = z_voxels;
    origin_z = oz;
    z_resolution = zres;
    unknown_threshold = unknown_thr_cfg + (16 - size_z);  // voxel_layer.cpp:89
    mark_threshold = mark_thr;
  }
  void matchSizeVoxel(const Grid2D& master, bool track_unknown) {  // voxel_layer.cpp:93-98
    matchSize(master, track_unknown);
    vg.resize(grid.size_x, grid.size_y, size_z);
  }
  bool worldToMap3DFloat(double wx, double wy, double wz, double& mx, double& my, double& mz) const {
    if (wx < grid.origin_x || wy < grid.origin_y || wz < origin_z) return false;
    mx = (wx - grid.origin_x) / grid.resolution;
    my = (wy - grid.origin_y) / grid.resolution;
    mz = (wz - origin_z) / z_resolution;
    return mx < grid.size_x && my < grid.size_y && mz < size_z;
  }
  bool worldToMap3D(double wx, double wy, double wz, uint32_t& mx, uint32_t& my, uint32_t& mz) const {
    if (wx < grid.origin_x || wy < grid.origin_y || wz < origin_z) return false;
    mx = (int)((wx - grid.origin_x) / grid.resolution);
    my = (int)((wy - grid.origin_y) / grid.resolution);
    mz = (int)((wz - origin_z) / z_resolution);
    return mx < grid.size_x && my < grid.size_y && mz < size_z;
  }
  void raytraceFreespaceVoxel(const Observation& obs, Bounds& b) {
    if (obs.cloud.empty()) return;
    double sensor_x, sensor_y, sensor_z;
    double ox = obs.ox, oy = obs.oy, oz = obs.oz;
    if (!worldToMap3DFloat(ox, oy, oz, sensor_x, sensor_y, sensor_z)) return;
    double map_end_x = grid.origin_x + grid.sizeInMetersX();
    double map_end_y = grid.origin_y + grid.sizeInMetersY();
    for (const Pt3f& p : obs.cloud) {
      double wpx = p.x, wpy = p.y, wpz = p.z;
      double distance = sqrt((ox - wpx) * (ox - wpx) + (oy - wpy) * (oy - wpy) + (oz - wpz) * (oz - wpz));  // voxel_layer.h dist()
      double scaling_fact = 1.0;
      scaling_fact = std::max(std::min(scaling_fact, (distance - 2 * grid.resolution) / distance), 0.0);
      wpx = scaling_fact * (wpx - ox) + ox;
      wpy = scaling_fact * (wpy - oy) + oy;
      wpz = scaling_fact * (wpz - oz) + oz;
      double a = wpx - ox, bb = wpy - oy, c = wpz - oz, t = 1.0;
      if (wpz > max_obstacle_height)
        t = std::max(0.0, std::min(t, (max_obstacle_height - 0.01 - oz) / c));
      else if (wpz < origin_z)
        t = std::min(t, (origin_z - oz) / c);
      if (wpx < grid.origin_x) t = std::min(t, (grid.origin_x - ox) / a);
      if (wpy < grid.origin_y) t = std::min(t, (grid.origin_y - oy) / bb);
      if (wpx > map_end_x) t = std::min(t, (map_end_x - ox) / a);
      if (wpy > map_end_y) t = std::min(t, (map_end_y - oy) / bb);
      wpx = ox + a * t;
      wpy = oy + bb * t;
      wpz = oz + c * t;
      double point_x, point_y, point_z;
      if (worldToMap3DFloat(wpx, wpy, wpz, point_x, point_y, point_z)) {
        uint32_t cell_raytrace_range = grid.cellDistance(obs.raytrace_range);
        vg.clearVoxelLineInMap(sensor_x, sensor_y, sensor_z, point_x, point_y, point_z, grid.cells.data(), unknown_threshold,
                               mark_threshold, FREE_SPACE, NO_INFORMATION, cell_raytrace_range);
        double dx = wpx - ox, dy = wpy - oy;
        double full_distance = hypot(dx, dy);
        double scale = std::min(1.0, obs.raytrace_range / full_distance);
        b.touch(ox + dx * scale, oy + dy * scale);
      }
    }
  }
  // voxel_layer.cpp:385-440: the 2-D grid and the voxel columns move together; both are reset first
  void updateOriginVoxel(double new_origin_x, double new_origin_y) {
    std::vector<uint32_t> oldv = vg.data;
    int cell_ox, cell_oy;
    grid.updateOrigin(new_origin_x, new_origin_y, &cell_ox, &cell_oy);
    vg.reset();
    const int sx = grid.size_x, sy = grid.size_y;
    for (int y = 0; y < sy; ++y)
      for (int x = 0; x < sx; ++x) {
        const int ox = x + cell_ox, oy = y + cell_oy;
        if (ox >= 0 && oy >= 0 && ox < sx && oy < sy) vg.data[(size_t)y * sx + x] = oldv[(size_t)oy * sx + ox];
      }
  }
  void updateBoundsVoxel(double rx, double ry, double ryaw, const std::vector<Observation>& marking,
                         const std::vector<Observation>& clearing, Bounds& b) {
    if (!enabled) return;
    useExtraBounds(b);  // voxel_layer.cpp:123
    for (const Observation& o : clearing) raytraceFreespaceVoxel(o, b);
    for (const Observation& obs : marking) {
      double sq_obstacle_range = obs.obstacle_range * obs.obstacle_range;
      for (const Pt3f& p : obs.cloud) {
        if (p.z > max_obstacle_height) continue;
        double sq_dist = (p.x - obs.ox) * (p.x - obs.ox) + (p.y - obs.oy) * (p.y - obs.oy) + (p.z - obs.oz) * (p.z - obs.oz);
        if (sq_dist >= sq_obstacle_range) continue;
        uint32_t mx, my, mz;
        if (p.z < origin_z) {
          if (!worldToMap3D(p.x, p.y, origin_z, mx, my, mz)) continue;
        } else if (!worldToMap3D(p.x, p.y, p.z, mx, my, mz))
          continue;
        if (vg.markVoxelInMap(mx, my, mz, mark_threshold)) {
          grid.cells[grid.index(mx, my)] = LETHAL_OBSTACLE;
          b.touch((double)p.x, (double)p.y);
        }
      }
    }
    updateFootprint(rx, ry, ryaw, b);
  }
};

// ---------------------------------------------------------------------------------------------
// Inflation layer: costmap_2d/plugins/inflation_layer.cpp:125-158 (updateBounds), :172-266
// (updateCosts), :277-293 (enqueue), :295-328 (computeCaches); inflation_layer.h:55-85 (CellData),
// :114-129 (computeCost)
// ---------------------------------------------------------------------------------------------
struct InflationOracle {
  double inflation_radius = 0, weight = 0, inscribed_radius = 0, resolution = 0;
  uint32_t cell_inflation_radius = 0;
  bool enabled = true, need_reinflation = false;
  double last_min_x = -FLT_MAX, last_min_y = -FLT_MAX, last_max_x = FLT_MAX, last_max_y = FLT_MAX;
  std::vector<double> cached_distances;  // (R+2)^2
  std::vector<uint8_t> cached_costs;

  uint8_t computeCost(double distance) const {
    uint8_t cost = 0;
    if (distance == 0)
      cost = LETHAL_OBSTACLE;
    else if (distance * resolution <= inscribed_radius)
      cost = INSCRIBED_INFLATED_OBSTACLE;
    else {
      double euclidean_distance = distance * resolution;
      double factor = exp(-1.0 * weight * (euclidean_distance - inscribed_radius));
      cost = (uint8_t)((INSCRIBED_INFLATED_OBSTACLE - 1) * factor);
    }
    return cost;
  }
  uint32_t stride() const { return cell_inflation_radius + 2; }
  void computeCaches() {
    if (cell_inflation_radius == 0) return;
    uint32_t n = stride();
    cached_distances.assign(size_t(n) * n, 0.0);
    cached_costs.assign(size_t(n) * n, 0);
    for (uint32_t i = 0; i < n; ++i)
      for (uint32_t j = 0; j < n; ++j) {
        cached_distances[i * n + j] = hypot(i, j);
        cached_costs[i * n + j] = computeCost(cached_distances[i * n + j]);
      }
  }
  // setInflationParameters (:362-376) + onFootprintChanged (:160-170) + matchSize (:110-123)
  void configure(const Grid2D& master, double radius, double cost_scaling_factor, double inscribed) {
    resolution = master.resolution;
    inflation_radius = radius;
    weight = cost_scaling_factor;
    inscribed_radius = inscribed;
    cell_inflation_radius = master.cellDistance(inflation_radius);
    computeCaches();
    need_reinflation = true;
  }
  void updateBounds(Bounds& b) {
    if (need_reinflation) {
      last_min_x = b.min_x;
      last_min_y = b.min_y;
      last_max_x = b.max_x;
      last_max_y = b.max_y;
      b.min_x = -std::numeric_limits<float>::max();
      b.min_y = -std::numeric_limits<float>::max();
      b.max_x = std::numeric_limits<float>::max();
      b.max_y = std::numeric_limits<float>::max();
      need_reinflation = false;
    } else {
      double tmin_x = last_min_x, tmin_y = last_min_y, tmax_x = last_max_x, tmax_y = last_max_y;
      last_min_x = b.min_x;
      last_min_y = b.min_y;
      last_max_x = b.max_x;
      last_max_y = b.max_y;
      b.min_x = std::min(tmin_x, b.min_x) - inflation_radius;
      b.min_y = std::min(tmin_y, b.min_y) - inflation_radius;
      b.max_x = std::max(tmax_x, b.max_x) + inflation_radius;
      b.max_y = std::max(tmax_y, b.max_y) + inflation_radius;
    }
  }

  struct CellData {
    double distance_;
    uint32_t index_, x_, y_, src_x_, src_y_;
  };
  struct Farther {  // operator< of inflation_layer.h:82-85 under std::less => min-distance on top
    bool operator()(const CellData& a, const CellData& b) const { return a.distance_ > b.distance_; }
  };

  // Reference algorithm: std::priority_queue (libstdc++ binary heap) carries the source cell.
  void updateCosts(Grid2D& master, int min_i, int min_j, int max_i, int max_j) const {
    if (!enabled) return;
    // (the reference dereferences NULL caches when cell_inflation_radius_ == 0; here R == 0 falls
    //  back to direct hypot()/computeCost(), i.e. only the seeds themselves are visited)
    uint8_t* arr = master.cells.data();
    uint32_t size_x = master.size_x, size_y = master.size_y;
    std::vector<uint8_t> seen(size_t(size_x) * size_y, 0);
    const uint32_t n = stride();
    const int R = (int)cell_inflation_radius;
    min_i -= R;
    min_j -= R;
    max_i += R;
    max_j += R;
    min_i = std::max(0, min_i);
    min_j = std::max(0, min_j);
    max_i = std::min(int(size_x), max_i);
    max_j = std::min(int(size_y), max_j);
    std::priority_queue<CellData, std::vector<CellData>, Farther> q;
    auto enqueue = [&](uint32_t index, uint32_t mx, uint32_t my, uint32_t sx, uint32_t sy) {
      if (seen[index]) return;
      uint32_t dx = abs((int)mx - (int)sx), dy = abs((int)my - (int)sy);
      double distance = cell_inflation_radius == 0 ? hypot(dx, dy) : cached_distances[dx * n + dy];
      if (distance > cell_inflation_radius) return;
      q.push(CellData{distance, index, mx, my, sx, sy});
    };
    for (int j = min_j; j < max_j; j++)
      for (int i = min_i; i < max_i; i++) {
        uint32_t idx = master.index(i, j);
        if (arr[idx] == LETHAL_OBSTACLE) enqueue(idx, i, j, i, j);
      }
    while (!q.empty()) {
      CellData c = q.top();
      q.pop();
      if (seen[c.index_]) continue;
      seen[c.index_] = 1;
      uint32_t dx = abs((int)c.x_ - (int)c.src_x_), dy = abs((int)c.y_ - (int)c.src_y_);
      uint8_t cost = cell_inflation_radius == 0 ? computeCost(hypot(dx, dy)) : cached_costs[dx * n + dy];
      uint8_t old_cost = arr[c.index_];
      if (old_cost == NO_INFORMATION && cost >= INSCRIBED_INFLATED_OBSTACLE)
        arr[c.index_] = cost;
      else
        arr[c.index_] = std::max(old_cost, cost);
      if (c.x_ > 0) enqueue(c.index_ - 1, c.x_ - 1, c.y_, c.src_x_, c.src_y_);
      if (c.y_ > 0) enqueue(c.index_ - size_x, c.x_, c.y_ - 1, c.src_x_, c.src_y_);
      if (c.x_ < size_x - 1) enqueue(c.index_ + 1, c.x_ + 1, c.y_, c.src_x_, c.src_y_);
      if (c.y_ < size_y - 1) enqueue(c.index_ + size_x, c.x_, c.y_ + 1, c.src_x_, c.src_y_);
    }
  }

  // Order-independent specification the GPU kernel is bit-exact against: every cell takes the
  // cost of its Euclidean-nearest LETHAL seed inside the grown box (windowed exact EDT), merged
  // with the same old-cost rule.  Never lower than updateCosts() above (SURVEY §7 hard part 1).
  void updateCostsExact(Grid2D& master, int min_i, int min_j, int max_i, int max_j) const {
    if (!enabled) return;
    uint32_t size_x = master.size_x, size_y = master.size_y;
    const uint32_t n = stride();
    const int R = (int)cell_inflation_radius;
    min_i = std::max(0, min_i - R);
    min_j = std::max(0, min_j - R);
    max_i = std::min(int(size_x), max_i + R);
    max_j = std::min(int(size_y), max_j + R);
    std::vector<uint8_t> best(size_t(size_x) * size_y, 0), hit(size_t(size_x) * size_y, 0);
    for (int j = min_j; j < max_j; j++)
      for (int i = min_i; i < max_i; i++) {
        if (master.cells[master.index(i, j)] != LETHAL_OBSTACLE) continue;
        for (int dy = -R; dy <= R; ++dy)
          for (int dx = -R; dx <= R; ++dx) {
            int x = i + dx, y = j + dy;
            if (x < 0 || y < 0 || x >= (int)size_x || y >= (int)size_y) continue;
            uint32_t ax = abs(dx), ay = abs(dy);
            double d = R == 0 ? hypot(ax, ay) : cached_distances[ax * n + ay];
            if (d > cell_inflation_radius) continue;
            uint8_t c = R == 0 ? computeCost(d) : cached_costs[ax * n + ay];
            size_t idx = master.index(x, y);
            hit[idx] = 1;
            best[idx] = std::max(best[idx], c);
          }
      }
    for (size_t idx = 0; idx < best.size(); ++idx) {
      if (!hit[idx]) continue;
      uint8_t old_cost = master.cells[idx], cost = best[idx];
      if (old_cost == NO_INFORMATION && cost >= INSCRIBED_INFLATED_OBSTACLE)
        master.cells[idx] = cost;
      else
        master.cells[idx] = std::max(old_cost, cost);
    }
  }
};

// ---------------------------------------------------------------------------------------------
// LayeredCostmap::updateMap, non-rolling: costmap_2d/src/layered_costmap.cpp:79-150
// plugin order static -> obstacle|voxel -> inflation (costmap_2d_ros.cpp:189-260)
// ---------------------------------------------------------------------------------------------
struct LayeredCostmapOracle {
  Grid2D master;
  bool track_unknown = false, rolling_window = false;
  bool has_static = false, has_obstacle = false, has_voxel = false, has_inflation = false;
  bool inflation_exact = false;  // use updateCostsExact instead of the PQ walk
  StaticLayerOracle slayer;
  VoxelLayerOracle olayer;  // voxel layer degenerates to the obstacle layer when !has_voxel
  InflationOracle ilayer;
  std::vector<Pt2> footprint;
  double inscribed_radius = 0, circumscribed_radius = 0;
  std::vector<Observation> marking, clearing;  // static observations (test hook), persist across cycles
  int bx0 = 0, bxn = 0, by0 = 0, byn = 0;

  void init(bool track_unknown_) {
    track_unknown = track_unknown_;
    master.default_value = track_unknown ? 255 : 0;  // layered_costmap.cpp:53-57
  }
  void resizeMap(uint32_t sx, uint32_t sy, double res, double ox, double oy) {  // layered_costmap.cpp:67-77
    master.resize(sx, sy, res, ox, oy);
    if (has_obstacle) {
      if (has_voxel)
        olayer.matchSizeVoxel(master, track_unknown);
      else
        olayer.matchSize(master, track_unknown);
    }
  }
  void setFootprint(const std::vector<Pt2>& fp) {  // layered_costmap.cpp:164-174
    footprint = fp;
    calculateMinAndMaxDistances(fp, inscribed_radius, circumscribed_radius);
    olayer.footprint_spec = fp;
    if (has_inflation) {
      ilayer.inscribed_radius = inscribed_radius;
      ilayer.resolution = master.resolution;
      ilayer.cell_inflation_radius = master.cellDistance(ilayer.inflation_radius);
      ilayer.computeCaches();
      ilayer.need_reinflation = true;
    }
  }
  void updateMap(double rx, double ry, double ryaw) {
    if (rolling_window) {  // layered_costmap.cpp:86-91
      double new_origin_x = rx - master.sizeInMetersX() / 2;
      double new_origin_y = ry - master.sizeInMetersY() / 2;
      master.updateOrigin(new_origin_x, new_origin_y);
    }
    if (!(has_static || has_obstacle || has_inflation)) return;
    Bounds b{1e30, 1e30, -1e30, -1e30};
    if (has_static) slayer.updateBounds(b);
    if (has_obstacle && rolling_window) {  // obstacle_layer.cpp:343-344 / voxel_layer.cpp:119-120
      if (has_voxel)
        olayer.updateOriginVoxel(rx - olayer.grid.sizeInMetersX() / 2, ry - olayer.grid.sizeInMetersY() / 2);
      else
        olayer.grid.updateOrigin(rx - olayer.grid.sizeInMetersX() / 2, ry - olayer.grid.sizeInMetersY() / 2);
    }
    if (has_obstacle) {
      if (has_voxel)
        olayer.updateBoundsVoxel(rx, ry, ryaw, marking, clearing, b);
      else
        olayer.updateBounds(rx, ry, ryaw, marking, clearing, b);
    }
    if (has_inflation) ilayer.updateBounds(b);
    int x0, xn, y0, yn;
    master.worldToMapEnforceBounds(b.min_x, b.min_y, x0, y0);
    master.worldToMapEnforceBounds(b.max_x, b.max_y, xn, yn);
    x0 = std::max(0, x0);
    xn = std::min(int(master.size_x), xn + 1);
    y0 = std::max(0, y0);
    yn = std::min(int(master.size_y), yn + 1);
    bx0 = x0;
    bxn = xn;
    by0 = y0;
    byn = yn;
    if (xn < x0 || yn < y0) return;
    master.resetMap(x0, y0, xn, yn);
    if (has_static) slayer.updateCosts(master, x0, y0, xn, yn);
    if (has_obstacle) olayer.updateCosts(master, x0, y0, xn, yn);
    if (has_inflation) {
      if (inflation_exact)
        ilayer.updateCostsExact(master, x0, y0, xn, yn);
      else
        ilayer.updateCosts(master, x0, y0, xn, yn);
    }
  }
};

}  // namespace oracle
