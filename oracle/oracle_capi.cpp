// ORACLE — TEST INFRASTRUCTURE ONLY (see costmap_oracle.hpp).  extern "C" surface for ctypes.
#include <chrono>
#include <thread>

#include "../include/navgpu.h"  // POD layouts only (navgpu_dwa_config, navgpu_plan_result)
#include "planner_oracle.hpp"
#include "global_planner_oracle.hpp"
#include "navfn_oracle.hpp"
#include "trajectory_planner_oracle.hpp"

using namespace oracle;

namespace {
DwaConfig toCfg(const navgpu_dwa_config& c) {
  DwaConfig d;
  d.max_trans_vel = c.max_trans_vel;
  d.min_trans_vel = c.min_trans_vel;
  d.max_vel_x = c.max_vel_x;
  d.min_vel_x = c.min_vel_x;
  d.max_vel_y = c.max_vel_y;
  d.min_vel_y = c.min_vel_y;
  d.max_rot_vel = c.max_rot_vel;
  d.min_rot_vel = c.min_rot_vel;
  d.acc_lim_x = c.acc_lim_x;
  d.acc_lim_y = c.acc_lim_y;
  d.acc_lim_theta = c.acc_lim_theta;
  d.sim_time = c.sim_time;
  d.sim_granularity = c.sim_granularity;
  d.angular_sim_granularity = c.angular_sim_granularity;
  d.sim_period = c.sim_period;
  d.vx_samples = c.vx_samples;
  d.vy_samples = c.vy_samples;
  d.vth_samples = c.vth_samples;
  d.use_dwa = c.use_dwa;
  d.discretize_by_time = c.discretize_by_time;
  d.sum_scores = c.sum_scores;
  d.path_distance_bias = c.path_distance_bias;
  d.goal_distance_bias = c.goal_distance_bias;
  d.occdist_scale = c.occdist_scale;
  d.forward_point_distance = c.forward_point_distance;
  d.cheat_factor = c.cheat_factor;
  d.oscillation_reset_dist = c.oscillation_reset_dist;
  d.oscillation_reset_angle = c.oscillation_reset_angle;
  d.allow_unknown = c.allow_unknown;
  d.rollout_trig = c.rollout_trig;
  return d;
}
std::vector<Pt2> toPts(const double* xy, uint32_t n) {
  std::vector<Pt2> v(n);
  for (uint32_t i = 0; i < n; ++i) {
    v[i].x = xy[2 * i];
    v[i].y = xy[2 * i + 1];
  }
  return v;
}
struct PlannerHandle {
  Grid2D grid;
  DwaPlannerOracle planner;
};
}  // namespace

extern "C" {

// ------------------------------------------------------------------ layered costmap
void* orc_lc_create(int track_unknown) {
  auto* lc = new LayeredCostmapOracle();
  lc->init(track_unknown != 0);
  return lc;
}
void orc_lc_destroy(void* h) { delete static_cast<LayeredCostmapOracle*>(h); }
void orc_lc_set_rolling(void* h, int rolling) { static_cast<LayeredCostmapOracle*>(h)->rolling_window = rolling != 0; }
void orc_lc_get_origin(void* h, double* xy) {
  auto* lc = static_cast<LayeredCostmapOracle*>(h);
  xy[0] = lc->master.origin_x;
  xy[1] = lc->master.origin_y;
}
void orc_lc_resize(void* h, uint32_t sx, uint32_t sy, double res, double ox, double oy) {
  static_cast<LayeredCostmapOracle*>(h)->resizeMap(sx, sy, res, ox, oy);
}
void orc_lc_add_static(void* h, const int8_t* occ, uint32_t sx, uint32_t sy, double res, double ox, double oy,
                       int track_unknown_space, int use_maximum) {
  auto* lc = static_cast<LayeredCostmapOracle*>(h);
  lc->has_static = true;
  lc->slayer.track_unknown_space = track_unknown_space != 0;
  lc->slayer.use_maximum = use_maximum != 0;
  // incomingMap resizes the layered costmap when geometry differs (static_layer.cpp:173-186)
  if (lc->master.size_x != sx || lc->master.size_y != sy || lc->master.resolution != res || lc->master.origin_x != ox ||
      lc->master.origin_y != oy)
    lc->resizeMap(sx, sy, res, ox, oy);
  lc->slayer.incomingMap(occ, sx, sy, res, ox, oy);
}
// StaticLayer under a rolling window: incomingMap resizes the LAYER only (static_layer.cpp:187-193), the master keeps its geometry
void orc_lc_add_static_rolling(void* h, const int8_t* occ, uint32_t sx, uint32_t sy, double res, double ox, double oy, int track_unknown_space,
                               int use_maximum, int trinary, int lethal_threshold, int unknown_cost_value) {
  auto* lc = static_cast<LayeredCostmapOracle*>(h);
  lc->has_static = true;
  lc->slayer.rolling = true;
  lc->slayer.track_unknown_space = track_unknown_space != 0;
  lc->slayer.use_maximum = use_maximum != 0;
  lc->slayer.trinary_costmap = trinary != 0;
  lc->slayer.lethal_threshold = (uint8_t)std::max(std::min(lethal_threshold, 100), 0);
  lc->slayer.unknown_cost_value = (uint8_t)unknown_cost_value;
  lc->slayer.incomingMap(occ, sx, sy, res, ox, oy);
}
void orc_lc_set_static_transform(void* h, const double* m12) {
  auto* lc = static_cast<LayeredCostmapOracle*>(h);
  for (int i = 0; i < 9; ++i) lc->slayer.tf_basis[i] = m12[i];
  for (int i = 0; i < 3; ++i) lc->slayer.tf_origin[i] = m12[9 + i];
}
void orc_lc_add_obstacle(void* h, int combination_method, int footprint_clearing, double max_obstacle_height) {
  auto* lc = static_cast<LayeredCostmapOracle*>(h);
  lc->has_obstacle = true;
  lc->olayer.combination_method = combination_method;
  lc->olayer.footprint_clearing_enabled = footprint_clearing != 0;
  lc->olayer.max_obstacle_height = max_obstacle_height;
  lc->olayer.matchSize(lc->master, lc->track_unknown);
  lc->olayer.footprint_spec = lc->footprint;
}
void orc_lc_add_voxel(void* h, int combination_method, int footprint_clearing, double max_obstacle_height, uint32_t z_voxels,
                      double origin_z, double z_resolution, uint32_t unknown_threshold, uint32_t mark_threshold) {
  auto* lc = static_cast<LayeredCostmapOracle*>(h);
  orc_lc_add_obstacle(h, combination_method, footprint_clearing, max_obstacle_height);
  lc->has_voxel = true;
  lc->olayer.configure(z_voxels, origin_z, z_resolution, unknown_threshold, mark_threshold);
  lc->olayer.matchSizeVoxel(lc->master, lc->track_unknown);
}
void orc_lc_add_inflation(void* h, double radius, double scaling, int exact) {
  auto* lc = static_cast<LayeredCostmapOracle*>(h);
  lc->has_inflation = true;
  lc->inflation_exact = exact != 0;
  lc->ilayer.configure(lc->master, radius, scaling, lc->inscribed_radius);
}
void orc_lc_set_inflation_exact(void* h, int exact) { static_cast<LayeredCostmapOracle*>(h)->inflation_exact = exact != 0; }
void orc_lc_set_footprint(void* h, const double* xy, uint32_t n) {
  static_cast<LayeredCostmapOracle*>(h)->setFootprint(toPts(xy, n));
}
double orc_lc_inscribed_radius(void* h) { return static_cast<LayeredCostmapOracle*>(h)->inscribed_radius; }
double orc_lc_circumscribed_radius(void* h) { return static_cast<LayeredCostmapOracle*>(h)->circumscribed_radius; }
void orc_lc_add_observation(void* h, double ox, double oy, double oz, const float* xyz, uint32_t n, double obstacle_range,
                            double raytrace_range, int marking, int clearing) {
  auto* lc = static_cast<LayeredCostmapOracle*>(h);
  Observation o;
  o.ox = ox;
  o.oy = oy;
  o.oz = oz;
  o.obstacle_range = obstacle_range;
  o.raytrace_range = raytrace_range;
  o.cloud.resize(n);
  for (uint32_t i = 0; i < n; ++i) {
    o.cloud[i].x = xyz[3 * i];
    o.cloud[i].y = xyz[3 * i + 1];
    o.cloud[i].z = xyz[3 * i + 2];
  }
  if (marking) lc->marking.push_back(o);
  if (clearing) lc->clearing.push_back(o);
}
void orc_lc_clear_observations(void* h) {
  auto* lc = static_cast<LayeredCostmapOracle*>(h);
  lc->marking.clear();
  lc->clearing.clear();
}
// CostmapLayer::resetBoundingBox on the obstacle / voxel layer (what Costmap2DROS::resetBoundingBox calls per layer)
void orc_lc_reset_bounding_box(void* h, double min_x, double min_y, double max_x, double max_y) {
  static_cast<LayeredCostmapOracle*>(h)->olayer.resetBoundingBox(min_x, min_y, max_x, max_y);
}
void orc_lc_update_map(void* h, double rx, double ry, double ryaw) { static_cast<LayeredCostmapOracle*>(h)->updateMap(rx, ry, ryaw); }
void orc_lc_get_master(void* h, uint8_t* out) {
  auto* lc = static_cast<LayeredCostmapOracle*>(h);
  memcpy(out, lc->master.cells.data(), lc->master.cells.size());
}
void orc_lc_set_master(void* h, const uint8_t* in) {
  auto* lc = static_cast<LayeredCostmapOracle*>(h);
  memcpy(lc->master.cells.data(), in, lc->master.cells.size());
}
void orc_lc_get_layer(void* h, int which, uint8_t* out) {  // 1 static, 2 obstacle
  auto* lc = static_cast<LayeredCostmapOracle*>(h);
  const Grid2D& g = which == 1 ? lc->slayer.grid : lc->olayer.grid;
  memcpy(out, g.cells.data(), g.cells.size());
}
void orc_lc_set_layer(void* h, int which, const uint8_t* in) {
  auto* lc = static_cast<LayeredCostmapOracle*>(h);
  Grid2D& g = which == 1 ? lc->slayer.grid : lc->olayer.grid;
  memcpy(g.cells.data(), in, g.cells.size());
}
void orc_lc_get_voxels(void* h, uint32_t* out) {
  auto* lc = static_cast<LayeredCostmapOracle*>(h);
  memcpy(out, lc->olayer.vg.data.data(), lc->olayer.vg.data.size() * 4);
}
void orc_lc_set_voxels(void* h, const uint32_t* in) {
  auto* lc = static_cast<LayeredCostmapOracle*>(h);
  memcpy(lc->olayer.vg.data.data(), in, lc->olayer.vg.data.size() * 4);
}
void orc_lc_get_bounds(void* h, int32_t* box) {
  auto* lc = static_cast<LayeredCostmapOracle*>(h);
  box[0] = lc->bx0;
  box[1] = lc->bxn;
  box[2] = lc->by0;
  box[3] = lc->byn;
}
void orc_lc_size(void* h, uint32_t* sx, uint32_t* sy) {
  auto* lc = static_cast<LayeredCostmapOracle*>(h);
  *sx = lc->master.size_x;
  *sy = lc->master.size_y;
}

// ------------------------------------------------------------------ standalone costmap pieces
// InflationLayer::updateCosts on a raw grid; exact=0 reference PQ walk, exact=1 windowed exact EDT
void orc_inflate(uint8_t* grid, uint32_t sx, uint32_t sy, double res, double radius, double scaling, double inscribed, int min_i,
                 int min_j, int max_i, int max_j, int exact) {
  Grid2D g;
  g.resize(sx, sy, res, 0, 0);
  memcpy(g.cells.data(), grid, size_t(sx) * sy);
  InflationOracle inf;
  inf.configure(g, radius, scaling, inscribed);
  if (exact)
    inf.updateCostsExact(g, min_i, min_j, max_i, max_j);
  else
    inf.updateCosts(g, min_i, min_j, max_i, max_j);
  memcpy(grid, g.cells.data(), size_t(sx) * sy);
}
// cached_costs_ table, (R+2)^2 bytes; returns R
uint32_t orc_cost_lut(double res, double radius, double scaling, double inscribed, uint8_t* costs, double* dists, uint32_t cap) {
  Grid2D g;
  g.resolution = res;
  InflationOracle inf;
  inf.configure(g, radius, scaling, inscribed);
  uint32_t n = inf.stride();
  if (size_t(n) * n <= cap) {
    if (costs) memcpy(costs, inf.cached_costs.data(), size_t(n) * n);
    if (dists) memcpy(dists, inf.cached_distances.data(), size_t(n) * n * sizeof(double));
  }
  return inf.cell_inflation_radius;
}
uint8_t orc_compute_cost(double res, double scaling, double inscribed, double distance_cells) {
  InflationOracle inf;
  inf.resolution = res;
  inf.weight = scaling;
  inf.inscribed_radius = inscribed;
  return inf.computeCost(distance_cells);
}
int orc_raytrace_cells(uint32_t sx, uint32_t sy, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t max_length,
                       uint32_t* out, int cap) {
  Grid2D g;
  g.size_x = sx;
  g.size_y = sy;
  int n = 0;
  g.raytraceLine(
      [&](uint32_t off) {
        if (n < cap) out[n] = off;
        ++n;
      },
      x0, y0, x1, y1, max_length);
  return n;
}
int orc_polygon_fill(uint8_t* grid, uint32_t sx, uint32_t sy, double res, double ox, double oy, const double* xy, uint32_t n,
                     uint8_t value) {
  Grid2D g;
  g.resize(sx, sy, res, ox, oy);
  memcpy(g.cells.data(), grid, size_t(sx) * sy);
  bool ok = g.setConvexPolygonCost(toPts(xy, n), value);
  memcpy(grid, g.cells.data(), size_t(sx) * sy);
  return ok ? 1 : 0;
}
void orc_merge(const uint8_t* layer, uint8_t* master, uint32_t sx, uint32_t sy, int mode, int min_i, int min_j, int max_i,
               int max_j) {
  Grid2D l, m;
  l.resize(sx, sy, 1, 0, 0);
  m.resize(sx, sy, 1, 0, 0);
  memcpy(l.cells.data(), layer, size_t(sx) * sy);
  memcpy(m.cells.data(), master, size_t(sx) * sy);
  if (mode == 0)
    updateWithOverwrite(l, m, min_i, min_j, max_i, max_j);
  else if (mode == 1)
    updateWithMax(l, m, min_i, min_j, max_i, max_j);
  else
    updateWithTrueOverwrite(l, m, min_i, min_j, max_i, max_j);
  memcpy(master, m.cells.data(), size_t(sx) * sy);
}
void orc_min_max_distances(const double* xy, uint32_t n, double* mn, double* mx) { calculateMinAndMaxDistances(toPts(xy, n), *mn, *mx); }

// voxel grid (voxel_grid_tests.cpp drives these)
void* orc_vg_create(uint32_t sx, uint32_t sy, uint32_t sz) {
  auto* v = new VoxelGridOracle();
  v->resize(sx, sy, sz);
  return v;
}
void orc_vg_destroy(void* h) { delete static_cast<VoxelGridOracle*>(h); }
void orc_vg_mark_line(void* h, double x0, double y0, double z0, double x1, double y1, double z1) {
  static_cast<VoxelGridOracle*>(h)->markVoxelLine(x0, y0, z0, x1, y1, z1);
}
void orc_vg_clear_line(void* h, double x0, double y0, double z0, double x1, double y1, double z1) {
  static_cast<VoxelGridOracle*>(h)->clearVoxelLine(x0, y0, z0, x1, y1, z1);
}
void orc_vg_mark_voxel(void* h, uint32_t x, uint32_t y, uint32_t z) { static_cast<VoxelGridOracle*>(h)->markVoxel(x, y, z); }
int orc_vg_get_voxel(void* h, uint32_t x, uint32_t y, uint32_t z) { return static_cast<VoxelGridOracle*>(h)->getVoxel(x, y, z); }
void orc_vg_data(void* h, uint32_t* out) {
  auto* v = static_cast<VoxelGridOracle*>(h);
  memcpy(out, v->data.data(), v->data.size() * 4);
}

// ------------------------------------------------------------------ planner pieces
int orc_velocity_samples(double mn, double mx, int n, double* out, int cap) {
  std::vector<double> s = velocitySamples(mn, mx, n);
  for (size_t i = 0; i < s.size() && (int)i < cap; ++i) out[i] = s[i];
  return (int)s.size();
}
int orc_line_cells(int x0, int y0, int x1, int y1, int32_t* out_xy, int cap) {
  int n = 0;
  lineCells(x0, y0, x1, y1, [&](int x, int y) {
    if (n < cap) {
      out_xy[2 * n] = x;
      out_xy[2 * n + 1] = y;
    }
    ++n;
    return true;
  });
  return n;
}
double orc_footprint_cost(const uint8_t* grid, uint32_t sx, uint32_t sy, double res, double ox, double oy, double x, double y,
                          double th, const double* fp_xy, uint32_t nfp, int allow_unknown) {
  Grid2D g;
  g.resize(sx, sy, res, ox, oy);
  memcpy(g.cells.data(), grid, size_t(sx) * sy);
  CostmapModelOracle m;
  m.cm = &g;
  m.allow_unknown = allow_unknown != 0;
  return m.footprintCost(x, y, th, toPts(fp_xy, nfp));
}
// ObstacleCostFunction::footprintCost (with the -6/-7 mapping and the centre-cell max)
double orc_obstacle_step_cost(const uint8_t* grid, uint32_t sx, uint32_t sy, double res, double ox, double oy, double x, double y,
                              double th, const double* fp_xy, uint32_t nfp, int allow_unknown) {
  Grid2D g;
  g.resize(sx, sy, res, ox, oy);
  memcpy(g.cells.data(), grid, size_t(sx) * sy);
  ObstacleCritic c;
  c.cm = &g;
  c.model.cm = &g;
  c.model.allow_unknown = allow_unknown != 0;
  c.footprint_spec = toPts(fp_xy, nfp);
  return c.footprintCost(x, y, th);
}
int orc_adjust_plan(const double* xy, uint32_t n, double res, double* out_xy, int cap) {
  std::vector<Pt2> out;
  MapGridOracle::adjustPlanResolution(toPts(xy, n), out, res);
  for (size_t i = 0; i < out.size() && (int)i < cap; ++i) {
    out_xy[2 * i] = out[i].x;
    out_xy[2 * i + 1] = out[i].y;
  }
  return (int)out.size();
}
// MapGridCostFunction::prepare on a raw grid.  mode 0 = setTargetCells (path), 1 = setLocalGoal
void orc_map_grid(const uint8_t* grid, uint32_t sx, uint32_t sy, double res, double ox, double oy, const double* plan_xy,
                  uint32_t n, int mode, int allow_unknown, double* dist_out) {
  Grid2D g;
  g.resize(sx, sy, res, ox, oy);
  memcpy(g.cells.data(), grid, size_t(sx) * sy);
  MapGridCritic c;
  c.cm = &g;
  c.is_local_goal_function = mode == 1;
  c.map.allow_unknown = allow_unknown != 0;
  c.target_poses = toPts(plan_xy, n);
  c.prepare();
  memcpy(dist_out, c.map.dist.data(), c.map.dist.size() * sizeof(double));
}
// MapGrid::computeTargetDistance from explicit seed cells (map_grid_test.cpp / utest.cpp style)
void orc_map_grid_seeded(const uint8_t* grid, uint32_t sx, uint32_t sy, const uint32_t* seeds, uint32_t nseeds, int allow_unknown,
                         double* dist_out) {
  Grid2D g;
  g.resize(sx, sy, 1, 0, 0);
  memcpy(g.cells.data(), grid, size_t(sx) * sy);
  MapGridOracle m;
  m.allow_unknown = allow_unknown != 0;
  m.sizeCheck(sx, sy);
  m.resetPathDist();
  std::queue<size_t> q;
  for (uint32_t i = 0; i < nseeds; ++i) {
    m.dist[seeds[i]] = 0.0;
    m.mark[seeds[i]] = 1;
    q.push(seeds[i]);
  }
  m.computeTargetDistance(q, g);
  memcpy(dist_out, m.dist.data(), m.dist.size() * sizeof(double));
}
int orc_samples(const navgpu_dwa_config* c, const float* pos, const float* vel, const float* goal, float* out_xyz, int cap) {
  DwaConfig cfg = toCfg(*c);
  TrajectoryGenerator g;
  V3f p, v, gl;
  for (int i = 0; i < 3; ++i) {
    p[i] = pos[i];
    v[i] = vel[i];
    gl[i] = goal[i];
  }
  g.initialise(p, v, gl, cfg);
  for (size_t i = 0; i < g.samples.size() && (int)i < cap; ++i)
    for (int k = 0; k < 3; ++k) out_xyz[3 * i + k] = g.samples[i][k];
  return (int)g.samples.size();
}
// generateTrajectory for one sample: returns n points, or -1 when the generator rejects it
int orc_generate_trajectory(const navgpu_dwa_config* c, const float* pos, const float* vel, const float* sample, double* xyth,
                            int cap, double* xv_yv_thv_dt) {
  DwaConfig cfg = toCfg(*c);
  TrajectoryGenerator g;
  g.cfg = &cfg;
  V3f p, v, s;
  for (int i = 0; i < 3; ++i) {
    p[i] = pos[i];
    v[i] = vel[i];
    s[i] = sample[i];
  }
  Trajectory t;
  bool ok = g.generateTrajectory(p, v, s, t);
  if (xv_yv_thv_dt) {
    xv_yv_thv_dt[0] = t.xv;
    xv_yv_thv_dt[1] = t.yv;
    xv_yv_thv_dt[2] = t.thetav;
    xv_yv_thv_dt[3] = t.time_delta;
  }
  if (!ok) return -1;
  for (size_t i = 0; i < t.x.size() && (int)i < cap; ++i) {
    xyth[3 * i] = t.x[i];
    xyth[3 * i + 1] = t.y[i];
    xyth[3 * i + 2] = t.th[i];
  }
  return (int)t.x.size();
}

// ------------------------------------------------------------------ full DWA planner
void* orc_dwa_create(uint32_t sx, uint32_t sy, double res, double ox, double oy) {
  auto* h = new PlannerHandle();
  h->grid.resize(sx, sy, res, ox, oy);
  h->planner.bind(&h->grid);
  return h;
}
void orc_dwa_destroy(void* h) { delete static_cast<PlannerHandle*>(h); }
void orc_dwa_set_costmap(void* h, const uint8_t* cells) {
  auto* p = static_cast<PlannerHandle*>(h);
  memcpy(p->grid.cells.data(), cells, p->grid.cells.size());
}
void orc_dwa_configure(void* h, const navgpu_dwa_config* c) { static_cast<PlannerHandle*>(h)->planner.reconfigure(toCfg(*c)); }
// MapGridCritic options DWAPlanner never sets itself: critic 0 path, 1 goal, 2 goal_front, 3 alignment
void orc_dwa_set_map_grid_options(void* h, int critic, int aggregation, double yshift) {
  auto& p = static_cast<PlannerHandle*>(h)->planner;
  MapGridCritic* c[4] = {&p.path, &p.goal, &p.goal_front, &p.alignment};
  c[critic]->aggregation = aggregation;
  c[critic]->yshift = yshift;
}
void orc_dwa_set_plan(void* h) { static_cast<PlannerHandle*>(h)->planner.setPlan(); }
// updatePlanAndLocalCosts + findBestPath.  sample_* arrays optional (cap slots).
int orc_dwa_cycle(void* h, const float* pos, const float* vel, const double* plan_xy, uint32_t n_plan, const double* fp_xy,
                  uint32_t nfp, navgpu_plan_result* out, double* traj_xyth, int traj_cap, double* sample_cost_ref,
                  double* sample_cost_full, int32_t* sample_status, int sample_cap) {
  auto* p = static_cast<PlannerHandle*>(h);
  V3f ps, vl;
  for (int i = 0; i < 3; ++i) {
    ps[i] = pos[i];
    vl[i] = vel[i];
  }
  p->planner.updatePlanAndLocalCosts(ps, toPts(plan_xy, n_plan));
  double drive[3];
  const Trajectory& r = p->planner.findBestPath(ps, vl, toPts(fp_xy, nfp), drive);
  const auto& rec = p->planner.records;
  if (out) {
    out->best_index = p->planner.best_index;
    out->n_samples = (int)rec.size();
    int scored = 0, valid = 0;
    for (const auto& s : rec) {
      scored += s.status == 1;
      valid += (s.status == 1 && s.cost_full >= 0);
    }
    out->n_scored = scored;
    out->n_valid = valid;
    out->n_points = (int)r.x.size();
    out->oscillation_flags = p->planner.oscillation.packFlags();
    out->xv = r.xv;
    out->yv = r.yv;
    out->thetav = r.thetav;
    out->reserved = 0;
    out->cost = r.cost;
    for (int i = 0; i < 3; ++i) out->drive[i] = drive[i];
  }
  for (size_t i = 0; i < r.x.size() && (int)i < traj_cap; ++i) {
    traj_xyth[3 * i] = r.x[i];
    traj_xyth[3 * i + 1] = r.y[i];
    traj_xyth[3 * i + 2] = r.th[i];
  }
  for (size_t i = 0; i < rec.size() && (int)i < sample_cap; ++i) {
    if (sample_cost_ref) sample_cost_ref[i] = rec[i].cost_ref;
    if (sample_cost_full) sample_cost_full[i] = rec[i].cost_full;
    if (sample_status) sample_status[i] = rec[i].status;
  }
  return (int)rec.size();
}
// DWAPlanner::updatePlanAndLocalCosts alone (what DWAPlannerROS::computeVelocityCommands always calls, :274)
void orc_dwa_update_plan(void* h, const float* pos, const double* plan_xy, uint32_t n_plan) {
  auto* p = static_cast<PlannerHandle*>(h);
  V3f ps;
  for (int i = 0; i < 3; ++i) ps[i] = pos[i];
  p->planner.updatePlanAndLocalCosts(ps, toPts(plan_xy, n_plan));
}
int orc_dwa_check_trajectory(void* h, const float* pos, const float* vel, const float* sample) {
  auto* p = static_cast<PlannerHandle*>(h);
  V3f ps, vl, s;
  for (int i = 0; i < 3; ++i) {
    ps[i] = pos[i];
    vl[i] = vel[i];
    s[i] = sample[i];
  }
  return p->planner.checkTrajectory(ps, vl, s) ? 1 : 0;
}
// which: 0 path, 1 goal, 2 goal_front, 3 alignment
void orc_dwa_get_grid(void* h, int which, double* out) {
  auto* p = static_cast<PlannerHandle*>(h);
  MapGridCritic* c[4] = {&p->planner.path, &p->planner.goal, &p->planner.goal_front, &p->planner.alignment};
  memcpy(out, c[which]->map.dist.data(), c[which]->map.dist.size() * sizeof(double));
}
// the velocity samples SimpleTrajectoryGenerator::initialise laid out for the last cycle (slot order)
int orc_dwa_get_samples(void* h, float* out_xyz, int cap) {
  auto* p = static_cast<PlannerHandle*>(h);
  const auto& sm = p->planner.gen.samples;
  for (size_t i = 0; i < sm.size() && (int)i < cap; ++i)
    for (int k = 0; k < 3; ++k) out_xyz[3 * i + k] = sm[i][k];
  return (int)sm.size();
}
double orc_dwa_alignment_scale(void* h) { return static_cast<PlannerHandle*>(h)->planner.alignment.scale; }
void orc_dwa_get_oscillation(void* h, uint32_t* flags, float* prev_xyz) {
  auto* p = static_cast<PlannerHandle*>(h);
  *flags = p->planner.oscillation.packFlags();
  for (int i = 0; i < 3; ++i) prev_xyz[i] = p->planner.oscillation.prev_stationary_pos[i];
}
void orc_dwa_set_oscillation(void* h, uint32_t flags, const float* prev_xyz) {
  auto* p = static_cast<PlannerHandle*>(h);
  p->planner.oscillation.unpackFlags(flags);
  for (int i = 0; i < 3; ++i) p->planner.oscillation.prev_stationary_pos[i] = prev_xyz[i];
}

// ------------------------------------------------------------------ cpu_baseline timing helpers (bench.py only)
// Runs `cycles` planner cycles on `n_threads` std::threads, one planner instance per thread, and
// returns seconds of wall time.  Same inputs every cycle (a bounded sample of the fleet workload).
double orc_bench_dwa(uint32_t sx, uint32_t sy, double res, const uint8_t* cells /* n_inst grids */, uint32_t n_inst,
                     const navgpu_dwa_config* c, const float* pos /* n_inst x3 */, const float* vel, const double* plan_xy,
                     uint32_t n_plan, const double* origins_xy, const double* fp_xy, uint32_t nfp, uint32_t cycles,
                     uint32_t n_threads, uint64_t* trajectories_scored) {
  std::vector<PlannerHandle*> hs(n_inst);
  for (uint32_t i = 0; i < n_inst; ++i) {
    hs[i] = static_cast<PlannerHandle*>(orc_dwa_create(sx, sy, res, origins_xy[2 * i], origins_xy[2 * i + 1]));
    orc_dwa_set_costmap(hs[i], cells + size_t(i) * sx * sy);
    orc_dwa_configure(hs[i], c);
  }
  std::vector<uint64_t> counts(n_threads, 0);
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (uint32_t t = 0; t < n_threads; ++t)
    th.emplace_back([&, t]() {
      for (uint32_t cyc = 0; cyc < cycles; ++cyc)
        for (uint32_t i = t; i < n_inst; i += n_threads) {
          navgpu_plan_result r;
          orc_dwa_cycle(hs[i], pos + 3 * i, vel + 3 * i, plan_xy + size_t(2) * n_plan * i, n_plan, fp_xy, nfp, &r, nullptr, 0,
                        nullptr, nullptr, nullptr, 0);
          counts[t] += r.n_scored;
        }
    });
  for (auto& x : th) x.join();
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  uint64_t total = 0;
  for (uint64_t x : counts) total += x;
  if (trajectories_scored) *trajectories_scored = total;
  for (auto* p : hs) orc_dwa_destroy(p);
  return dt;
}
// Full-window InflationLayer::updateCosts (reference PQ walk) on n_maps grids, returns seconds.
double orc_bench_inflate(const uint8_t* cells, uint32_t n_maps, uint32_t sx, uint32_t sy, double res, double radius, double scaling,
                         double inscribed, uint32_t reps, uint32_t n_threads) {
  std::vector<std::vector<uint8_t>> work(n_maps);
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (uint32_t t = 0; t < n_threads; ++t)
    th.emplace_back([&, t]() {
      for (uint32_t r = 0; r < reps; ++r)
        for (uint32_t i = t; i < n_maps; i += n_threads) {
          work[i].assign(cells + size_t(i) * sx * sy, cells + size_t(i + 1) * sx * sy);
          orc_inflate(work[i].data(), sx, sy, res, radius, scaling, inscribed, 0, 0, sx, sy, 0);
        }
    });
  for (auto& x : th) x.join();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}


// ---------------------------------------------------------------------------------------------
// legacy TrajectoryPlanner (trajectory_planner_oracle.hpp)
// ---------------------------------------------------------------------------------------------
struct TpHandle {
  Grid2D grid;
  TrajectoryPlannerOracle tp;
};
static TpConfig toTpCfg(const navgpu_tp_config& c) {
  TpConfig o;
  o.acc_lim_x = c.acc_lim_x;
  o.acc_lim_y = c.acc_lim_y;
  o.acc_lim_theta = c.acc_lim_theta;
  o.sim_time = c.sim_time;
  o.sim_granularity = c.sim_granularity;
  o.angular_sim_granularity = c.angular_sim_granularity;
  o.vx_samples = c.vx_samples <= 0 ? 1 : c.vx_samples;  // trajectory_planner.cpp:98-107
  o.vtheta_samples = c.vtheta_samples <= 0 ? 1 : c.vtheta_samples;
  o.pdist_scale = c.pdist_scale;
  o.gdist_scale = c.gdist_scale;
  o.occdist_scale = c.occdist_scale;
  o.heading_lookahead = c.heading_lookahead;
  o.oscillation_reset_dist = c.oscillation_reset_dist;
  o.escape_reset_dist = c.escape_reset_dist;
  o.escape_reset_theta = c.escape_reset_theta;
  o.holonomic_robot = c.holonomic_robot;
  o.max_vel_x = c.max_vel_x;
  o.min_vel_x = c.min_vel_x;
  o.max_vel_th = c.max_vel_th;
  o.min_vel_th = c.min_vel_th;
  o.min_in_place_vel_th = c.min_in_place_vel_th;
  o.backup_vel = c.backup_vel;
  o.dwa = c.dwa;
  o.sim_period = c.sim_period;
  o.n_y_vels = c.n_y_vels;
  for (int i = 0; i < 8; ++i) o.y_vels[i] = c.y_vels[i];
  o.allow_unknown = c.allow_unknown;
  o.heading_scoring = c.heading_scoring;
  o.simple_attractor = c.simple_attractor;
  o.heading_scoring_timestep = c.heading_scoring_timestep;
  return o;
}
void* orc_tp_create(uint32_t sx, uint32_t sy, double res, double ox, double oy, const uint8_t* cells, const navgpu_tp_config* c,
                    const double* fp_xy, uint32_t nfp) {
  auto* h = new TpHandle();
  h->grid.resize(sx, sy, res, ox, oy);
  std::copy(cells, cells + size_t(sx) * sy, h->grid.cells.begin());
  h->tp.bind(&h->grid, toTpCfg(*c), toPts(fp_xy, nfp));
  return h;
}
void orc_tp_destroy(void* h) { delete static_cast<TpHandle*>(h); }
void orc_tp_set_costmap(void* h, const uint8_t* cells) {
  auto* p = static_cast<TpHandle*>(h);
  std::copy(cells, cells + p->grid.cells.size(), p->grid.cells.begin());
}
void orc_tp_update_plan(void* h, const double* plan_xy, uint32_t n, int compute_dists) {
  static_cast<TpHandle*>(h)->tp.updatePlan(toPts(plan_xy, n), compute_dists != 0);
}
int orc_tp_find_best_path(void* h, const float* pos, const float* vel, navgpu_tp_result* out, double* traj_xyth, int traj_cap,
                          navgpu_tp_sample* samples, int sample_cap) {
  auto* p = static_cast<TpHandle*>(h);
  V3f ps, vl;
  for (int i = 0; i < 3; ++i) {
    ps[i] = pos[i];
    vl[i] = vel[i];
  }
  double drive[3];
  Trajectory best = p->tp.findBestPath(ps, vl, drive);
  const auto& rec = p->tp.records;
  if (out) {
    *out = navgpu_tp_result();
    out->xv = best.xv;
    out->yv = best.yv;
    out->thetav = best.thetav;
    out->cost = best.cost;
    for (int i = 0; i < 3; ++i) out->drive[i] = drive[i];
    out->n_points = (int)best.x.size();
    out->n_samples = (int)rec.size();
    out->best_sample = -1;
  }
  for (size_t i = 0; i < best.x.size() && (int)i < traj_cap; ++i) {
    traj_xyth[3 * i] = best.x[i];
    traj_xyth[3 * i + 1] = best.y[i];
    traj_xyth[3 * i + 2] = best.th[i];
  }
  for (size_t i = 0; i < rec.size() && (int)i < sample_cap; ++i)
    samples[i] = navgpu_tp_sample{rec[i].vx, rec[i].vy, rec[i].vth, rec[i].cost, rec[i].n_points, 0};
  return (int)rec.size();
}
double orc_tp_score_trajectory(void* h, const double* pose, const double* vel, const double* vs) {
  return static_cast<TpHandle*>(h)->tp.scoreTrajectory(pose[0], pose[1], pose[2], vel[0], vel[1], vel[2], vs[0], vs[1], vs[2]);
}
// raw generateTrajectory with explicit acceleration limits and impossible_cost (utest.cpp:75-102)
double orc_tp_generate(void* h, const double* pose, const double* vel, const double* vs, const double* acc, double impossible_cost) {
  Trajectory t;
  static_cast<TpHandle*>(h)->tp.generateTrajectory(pose[0], pose[1], pose[2], vel[0], vel[1], vel[2], vs[0], vs[1], vs[2], acc[0], acc[1],
                                                   acc[2], impossible_cost, t);
  return t.cost;
}
void orc_tp_get_grid(void* h, int which, double* out) {
  auto* p = static_cast<TpHandle*>(h);
  const MapGridOracle& g = which == 0 ? p->tp.path_map : p->tp.goal_map;
  std::copy(g.dist.begin(), g.dist.end(), out);
}
void orc_tp_get_state(void* h, navgpu_tp_state* s) {
  const TrajectoryPlannerOracle& t = static_cast<TpHandle*>(h)->tp;
  s->flags = (t.stuck_left ? NAVGPU_TP_STUCK_LEFT : 0) | (t.stuck_right ? NAVGPU_TP_STUCK_RIGHT : 0) |
             (t.rotating_left ? NAVGPU_TP_ROTATING_LEFT : 0) | (t.rotating_right ? NAVGPU_TP_ROTATING_RIGHT : 0) |
             (t.stuck_left_strafe ? NAVGPU_TP_STUCK_LEFT_STRAFE : 0) | (t.stuck_right_strafe ? NAVGPU_TP_STUCK_RIGHT_STRAFE : 0) |
             (t.strafe_left ? NAVGPU_TP_STRAFE_LEFT : 0) | (t.strafe_right ? NAVGPU_TP_STRAFE_RIGHT : 0) |
             (t.escaping ? NAVGPU_TP_ESCAPING : 0);
  s->reserved = 0;
  s->prev_x = t.prev_x;
  s->prev_y = t.prev_y;
  s->escape_x = t.escape_x;
  s->escape_y = t.escape_y;
  s->escape_theta = t.escape_theta;
}
void orc_tp_set_state(void* h, const navgpu_tp_state* s) {
  TrajectoryPlannerOracle& t = static_cast<TpHandle*>(h)->tp;
  t.stuck_left = s->flags & NAVGPU_TP_STUCK_LEFT;
  t.stuck_right = s->flags & NAVGPU_TP_STUCK_RIGHT;
  t.rotating_left = s->flags & NAVGPU_TP_ROTATING_LEFT;
  t.rotating_right = s->flags & NAVGPU_TP_ROTATING_RIGHT;
  t.stuck_left_strafe = s->flags & NAVGPU_TP_STUCK_LEFT_STRAFE;
  t.stuck_right_strafe = s->flags & NAVGPU_TP_STUCK_RIGHT_STRAFE;
  t.strafe_left = s->flags & NAVGPU_TP_STRAFE_LEFT;
  t.strafe_right = s->flags & NAVGPU_TP_STRAFE_RIGHT;
  t.escaping = s->flags & NAVGPU_TP_ESCAPING;
  t.prev_x = s->prev_x;
  t.prev_y = s->prev_y;
  t.escape_x = s->escape_x;
  t.escape_y = s->escape_y;
  t.escape_theta = s->escape_theta;
}
// FootprintHelper::getFootprintCells on a map of the given geometry; returns the number of cells
int orc_tp_footprint_cells(uint32_t sx, uint32_t sy, double res, double ox, double oy, const float* pos, const double* fp_xy, uint32_t nfp,
                           int fill, int32_t* out_xy, int cap) {
  Grid2D g;
  g.resize(sx, sy, res, ox, oy);
  V3f ps;
  for (int i = 0; i < 3; ++i) ps[i] = pos[i];
  std::vector<FpCell> c = getFootprintCells(ps, toPts(fp_xy, nfp), g, fill != 0);
  for (size_t i = 0; i < c.size() && (int)i < cap; ++i) {
    out_xy[2 * i] = c[i].x;
    out_xy[2 * i + 1] = c[i].y;
  }
  return (int)c.size();
}
// ------------------------------------------------------------------ navfn::NavFn
// cost_mode: 0 = cmap IS costarr (path_calc_test.cpp:52 memcpy), 1 = setCostmap(cmap, isROS = true, allow_unknown),
// 2 = setCostmap(cmap, isROS = false).  Returns the path length (0 = none); potarr_out (ns floats) and path_xy optional.
int orc_navfn_plan(const uint8_t* cmap, int nx, int ny, int cost_mode, int allow_unknown, const int* goal, const int* start, int astar,
                   int at_start, float* potarr_out, float* path_xy, int path_cap, int* cycles_used) {
  NavFnOracle nav(nx, ny);
  if (cost_mode == 0)
    memcpy(nav.costarr.data(), cmap, (size_t)nx * ny);
  else
    nav.setCostmap(cmap, cost_mode == 1, allow_unknown != 0);
  nav.goal[0] = goal[0];
  nav.goal[1] = goal[1];
  nav.start[0] = start[0];
  nav.start[1] = start[1];
  nav.setupNavFn();
  int cyc = 0, len;
  if (astar) {
    nav.propagate<true>(std::max(nx * ny / 20, nx + ny), true, &cyc);
    len = nav.calcPath(nx * 4);
  } else {
    nav.propagate<false>(std::max(nx * ny / 20, nx + ny), at_start != 0, &cyc);
    len = nav.calcPath(nx * ny / 2);
  }
  if (cycles_used) *cycles_used = cyc;
  if (potarr_out) memcpy(potarr_out, nav.potarr.data(), sizeof(float) * (size_t)nx * ny);
  for (int i = 0; i < len && i < path_cap && path_xy; ++i) {
    path_xy[2 * i] = nav.pathx[i];
    path_xy[2 * i + 1] = nav.pathy[i];
  }
  return len;
}
// The fixed point of NavFn::updateCell's rule (NavFnOracle::propagateFixedPoint) + calcPath on it: the checker of the HIP
// path's tiled wavefront mode.  Same arguments as orc_navfn_plan (Dijkstra only).
int orc_navfn_fixed_point(const uint8_t* cmap, int nx, int ny, int cost_mode, int allow_unknown, const int* goal, const int* start,
                          float* potarr_out, float* path_xy, int path_cap) {
  NavFnOracle nav(nx, ny);
  if (cost_mode == 0)
    memcpy(nav.costarr.data(), cmap, (size_t)nx * ny);
  else
    nav.setCostmap(cmap, cost_mode == 1, allow_unknown != 0);
  nav.goal[0] = goal[0];
  nav.goal[1] = goal[1];
  nav.start[0] = start[0];
  nav.start[1] = start[1];
  nav.setupNavFn();
  nav.propagateFixedPoint();
  const int len = nav.calcPath(nx * ny / 2);
  if (potarr_out) memcpy(potarr_out, nav.potarr.data(), sizeof(float) * (size_t)nx * ny);
  for (int i = 0; i < len && i < path_cap && path_xy; ++i) {
    path_xy[2 * i] = nav.pathx[i];
    path_xy[2 * i + 1] = nav.pathy[i];
  }
  return len;
}
// ------------------------------------------------------------------ global_planner (second half)
// params = {use_dijkstra, use_quadratic, use_grid_path, old_navfn_behavior, allow_unknown, lethal_cost, neutral_cost, outline_map,
//           fixed_point (1: the Dijkstra rule's fixed point instead of the reference-order expansion - the wavefront mode's checker)}
int orc_global_planner_plan(const uint8_t* cmap, int nx, int ny, const int* params, float cost_factor, const double* start_xy, const double* goal_xy,
                            const int* goal_cell, float* potential_out, float* path_xy, int path_cap, int* found_legal, int* cycles_used) {
  GlobalPlannerParams p;
  p.use_dijkstra = params[0];
  p.use_quadratic = params[1];
  p.use_grid_path = params[2];
  p.old_navfn_behavior = params[3];
  p.allow_unknown = params[4];
  p.lethal_cost = params[5];
  p.neutral_cost = params[6];
  p.outline_map = params[7];
  p.cost_factor = cost_factor;
  GlobalPlannerOracle gp(nx, ny, p);
  bool legal = false;
  const bool ok = gp.plan(cmap, start_xy[0], start_xy[1], goal_xy[0], goal_xy[1], goal_cell[0], goal_cell[1], &legal, /*fixed_point=*/params[8] != 0);
  if (found_legal) *found_legal = legal;
  if (cycles_used) *cycles_used = gp.cycles_used;
  if (potential_out) memcpy(potential_out, gp.potential.data(), sizeof(float) * (size_t)nx * ny);
  const int len = ok ? (int)gp.path.size() : 0;
  for (int i = 0; i < len && i < path_cap && path_xy; ++i) {
    path_xy[2 * i] = gp.path[i].first;
    path_xy[2 * i + 1] = gp.path[i].second;
  }
  return len;
}
// the two pieces of the global_planner oracle that oracle/_ref/libref_gp.so can check directly
void orc_gp_calculate_potential(int quadratic, const float* potential, int nx, int ny, const uint8_t* cost, const int* cells, const float* prev,
                                int count, float* out) {
  GlobalPlannerParams p;
  p.use_quadratic = quadratic;
  GlobalPlannerOracle gp(nx, ny, p);
  memcpy(gp.potential.data(), potential, sizeof(float) * (size_t)nx * ny);
  for (int i = 0; i < count; ++i) out[i] = gp.calculatePotential(cost[i], cells[i], prev[i]);
}
int orc_gp_grid_path(const float* potential, int nx, int ny, double start_x, double start_y, double end_x, double end_y, float* path_xy, int cap) {
  GlobalPlannerParams p;
  GlobalPlannerOracle gp(nx, ny, p);
  memcpy(gp.potential.data(), potential, sizeof(float) * (size_t)nx * ny);
  if (!gp.gridPath(start_x, start_y, end_x, end_y)) return 0;
  const int n = (int)gp.path.size();
  for (int i = 0; i < n && i < cap; ++i) {
    path_xy[2 * i] = gp.path[i].first;
    path_xy[2 * i + 1] = gp.path[i].second;
  }
  return n;
}
}  // extern "C"
