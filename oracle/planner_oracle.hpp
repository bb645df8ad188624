// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of the local-planner half of the hot path (base_local_planner +
// dwa_local_planner).  See costmap_oracle.hpp for the usage rules and the pinning statement.
//
// Floating-point contract restated from the reference (SURVEY §7 hard part 2): rollout state is
// three floats (Eigen::Vector3f), every step is evaluated in double and rounded back to float
// (simple_trajectory_generator.cpp:253-260); unqualified cos/sin/hypot on float arguments resolve
// to the double C functions (Kinetic toolchain, <cmath> only) — "parity pinned to double libm".
// allow_unknown is an explicit input (the reference reads an uninitialised member,
// costmap_model.cpp:47 / map_grid.cpp:106 — SURVEY §7 hard part 3).
#pragma once
#include <queue>

#include "costmap_oracle.hpp"

namespace oracle {

// base_local_planner/include/base_local_planner/velocity_iterator.h:49-74
inline std::vector<double> velocitySamples(double min, double max, int num_samples) {
  std::vector<double> s;
  if (min == max) {
    s.push_back(min);
  } else {
    num_samples = std::max(2, num_samples);
    double step_size = (max - min) / double(std::max(1, (num_samples - 1)));
    double current, next = min;
    for (int j = 0; j < num_samples - 1; ++j) {
      current = next;
      next += step_size;
      s.push_back(current);
      if ((current < 0) && (next > 0)) s.push_back(0.0);
    }
    s.push_back(max);
  }
  return s;
}

// base_local_planner/include/base_local_planner/line_iterator.h:38-139, as a cell visitor
template <class F>
inline void lineCells(int x0, int y0, int x1, int y1, F&& visit) {
  int deltax = abs(x1 - x0), deltay = abs(y1 - y0);
  int x = x0, y = y0;
  int xinc1, xinc2, yinc1, yinc2, den, num, numadd, numpixels;
  xinc1 = xinc2 = (x1 >= x0) ? 1 : -1;
  yinc1 = yinc2 = (y1 >= y0) ? 1 : -1;
  if (deltax >= deltay) {
    xinc1 = 0;
    yinc2 = 0;
    den = deltax;
    num = deltax / 2;
    numadd = deltay;
    numpixels = deltax;
  } else {
    xinc2 = 0;
    yinc1 = 0;
    den = deltay;
    num = deltay / 2;
    numadd = deltax;
    numpixels = deltay;
  }
  for (int curpixel = 0; curpixel <= numpixels; ++curpixel) {
    if (!visit(x, y)) return;
    num += numadd;
    if (num >= den) {
      num -= den;
      x += xinc1;
      y += yinc1;
    }
    x += xinc2;
    y += yinc2;
  }
}

// base_local_planner/include/base_local_planner/local_planner_limits.h + cfg/DWAPlanner.cfg
struct DwaConfig {
  double max_trans_vel = 0.55, min_trans_vel = 0.1;
  double max_vel_x = 0.55, min_vel_x = 0.0, max_vel_y = 0.1, min_vel_y = -0.1;
  double max_rot_vel = 1.0, min_rot_vel = 0.4;
  double acc_lim_x = 2.5, acc_lim_y = 2.5, acc_lim_theta = 3.2;
  double sim_time = 1.7, sim_granularity = 0.025, angular_sim_granularity = 0.1, sim_period = 0.05;
  int vx_samples = 3, vy_samples = 10, vth_samples = 20;
  int use_dwa = 1, discretize_by_time = 0, sum_scores = 0;
  double path_distance_bias = 32.0, goal_distance_bias = 24.0, occdist_scale = 0.01;
  double forward_point_distance = 0.325, cheat_factor = 1.0;
  double oscillation_reset_dist = 0.05, oscillation_reset_angle = 0.2;
  int allow_unknown = 1;
  // Which function the unqualified cos(pos[2]) / sin(pos[2]) of computeNewPositions (simple_trajectory_generator.cpp:253-260) names
  // for its FLOAT argument is a property of the build, not of the source: 0 = ::cos(double) (only the C declarations in scope - the
  // fork's Kinetic / GCC 5 toolchain, SURVEY 7 hard part 2), 1 = the float overload (libstdc++ >= 6 with <math.h> in scope puts
  // std::cos(float) into the global namespace): then vel[0] * cos(pos[2]) is a float product.
  int rollout_trig = 0;
};

// base_local_planner/include/base_local_planner/trajectory.h:44-116
struct Trajectory {
  double xv = 0, yv = 0, thetav = 0, cost = -1.0, time_delta = 0;
  std::vector<double> x, y, th;
  void reset() {
    x.clear();
    y.clear();
    th.clear();
  }
};

struct V3f {
  float v[3] = {0, 0, 0};
  float& operator[](int i) { return v[i]; }
  const float& operator[](int i) const { return v[i]; }
};

// ---------------------------------------------------------------------------------------------
// SimpleTrajectoryGenerator: base_local_planner/src/simple_trajectory_generator.cpp:60-276
// ---------------------------------------------------------------------------------------------
struct TrajectoryGenerator {
  const DwaConfig* cfg = nullptr;
  V3f pos, vel;
  std::vector<V3f> samples;
  size_t next_index = 0;

  void initialise(const V3f& p, const V3f& v, const V3f& goal, const DwaConfig& c) {
    cfg = &c;
    double max_vel_th = c.max_rot_vel, min_vel_th = -1.0 * max_vel_th;
    V3f acc_lim;
    acc_lim[0] = c.acc_lim_x;
    acc_lim[1] = c.acc_lim_y;
    acc_lim[2] = c.acc_lim_theta;
    pos = p;
    vel = v;
    next_index = 0;
    samples.clear();
    double min_vel_x = c.min_vel_x, max_vel_x = c.max_vel_x, min_vel_y = c.min_vel_y, max_vel_y = c.max_vel_y;
    V3f vsamples;
    vsamples[0] = c.vx_samples;
    vsamples[1] = c.vy_samples;
    vsamples[2] = c.vth_samples;
    if (vsamples[0] * vsamples[1] * vsamples[2] > 0) {
      V3f max_vel, min_vel;
      if (!c.use_dwa) {
        double dist = hypot(goal[0] - p[0], goal[1] - p[1]);
        max_vel_x = std::max(std::min(max_vel_x, dist / c.sim_time), min_vel_x);
        max_vel_y = std::max(std::min(max_vel_y, dist / c.sim_time), min_vel_y);
        max_vel[0] = std::min(max_vel_x, v[0] + acc_lim[0] * c.sim_time);
        max_vel[1] = std::min(max_vel_y, v[1] + acc_lim[1] * c.sim_time);
        max_vel[2] = std::min(max_vel_th, v[2] + acc_lim[2] * c.sim_time);
        min_vel[0] = std::max(min_vel_x, v[0] - acc_lim[0] * c.sim_time);
        min_vel[1] = std::max(min_vel_y, v[1] - acc_lim[1] * c.sim_time);
        min_vel[2] = std::max(min_vel_th, v[2] - acc_lim[2] * c.sim_time);
      } else {
        max_vel[0] = std::min(max_vel_x, v[0] + acc_lim[0] * c.sim_period);
        max_vel[1] = std::min(max_vel_y, v[1] + acc_lim[1] * c.sim_period);
        max_vel[2] = std::min(max_vel_th, v[2] + acc_lim[2] * c.sim_period);
        min_vel[0] = std::max(min_vel_x, v[0] - acc_lim[0] * c.sim_period);
        min_vel[1] = std::max(min_vel_y, v[1] - acc_lim[1] * c.sim_period);
        min_vel[2] = std::max(min_vel_th, v[2] - acc_lim[2] * c.sim_period);
      }
      std::vector<double> xs = velocitySamples(min_vel[0], max_vel[0], vsamples[0]);
      std::vector<double> ys = velocitySamples(min_vel[1], max_vel[1], vsamples[1]);
      std::vector<double> ts = velocitySamples(min_vel[2], max_vel[2], vsamples[2]);
      V3f s;
      for (double xv : xs) {
        s[0] = xv;
        for (double yv : ys) {
          s[1] = yv;
          for (double tv : ts) {
            s[2] = tv;
            samples.push_back(s);
          }
        }
      }
    }
  }

  static V3f computeNewPositions(const V3f& pos, const V3f& vel, double dt, int rollout_trig = 0) {  // :253-260
    V3f n;
    if (rollout_trig) {  // cos(float) -> float: the first product is a float one; M_PI_2 + pos[2] is a double either way
      n[0] = pos[0] + (vel[0] * cosf(pos[2]) + vel[1] * cos(M_PI_2 + pos[2])) * dt;
      n[1] = pos[1] + (vel[0] * sinf(pos[2]) + vel[1] * sin(M_PI_2 + pos[2])) * dt;
    } else {
      n[0] = pos[0] + (vel[0] * cos((double)pos[2]) + vel[1] * cos(M_PI_2 + pos[2])) * dt;
      n[1] = pos[1] + (vel[0] * sin((double)pos[2]) + vel[1] * sin(M_PI_2 + pos[2])) * dt;
    }
    n[2] = pos[2] + vel[2] * dt;
    return n;
  }
  static V3f computeNewVelocities(const V3f& target, const V3f& vel, const V3f& acclimits, double dt) {  // :265-276
    V3f n;
    for (int i = 0; i < 3; ++i) {
      if (vel[i] < target[i])
        n[i] = std::min(double(target[i]), vel[i] + acclimits[i] * dt);
      else
        n[i] = std::max(double(target[i]), vel[i] - acclimits[i] * dt);
    }
    return n;
  }

  // :180-251.  Returns false for rejected samples (cost stays -1, no points).
  bool generateTrajectory(V3f p, V3f v, const V3f& sample, Trajectory& traj) const {
    const DwaConfig& c = *cfg;
    double vmag = hypot((double)sample[0], (double)sample[1]);
    double eps = 1e-4;
    traj.cost = -1.0;
    traj.reset();
    if ((c.min_trans_vel >= 0 && vmag + eps < c.min_trans_vel) && (c.min_rot_vel >= 0 && fabs((double)sample[2]) + eps < c.min_rot_vel))
      return false;
    if (c.max_trans_vel >= 0 && vmag - eps > c.max_trans_vel) return false;
    int num_steps;
    if (c.discretize_by_time) {
      num_steps = ceil(c.sim_time / c.sim_granularity);
    } else {
      double sim_time_distance = vmag * c.sim_time;
      double sim_time_angle = fabs((double)sample[2]) * c.sim_time;
      num_steps = ceil(std::max(sim_time_distance / c.sim_granularity, sim_time_angle / c.angular_sim_granularity));
    }
    double dt = c.sim_time / num_steps;
    traj.time_delta = dt;
    V3f acc;
    acc[0] = c.acc_lim_x;
    acc[1] = c.acc_lim_y;
    acc[2] = c.acc_lim_theta;
    V3f loop_vel;
    bool continued_acceleration = !c.use_dwa;
    if (continued_acceleration) {
      loop_vel = computeNewVelocities(sample, v, acc, dt);
      traj.xv = loop_vel[0];
      traj.yv = loop_vel[1];
      traj.thetav = loop_vel[2];
    } else {
      loop_vel = sample;
      traj.xv = sample[0];
      traj.yv = sample[1];
      traj.thetav = sample[2];
    }
    for (int i = 0; i < num_steps; ++i) {
      traj.x.push_back(p[0]);
      traj.y.push_back(p[1]);
      traj.th.push_back(p[2]);
      if (continued_acceleration) loop_vel = computeNewVelocities(sample, loop_vel, acc, dt);
      p = computeNewPositions(p, loop_vel, dt, c.rollout_trig);
    }
    return num_steps > 0;
  }
  bool hasMore() const { return next_index < samples.size(); }
  bool next(Trajectory& t) {  // nextTrajectory :160-174
    bool r = false;
    if (hasMore()) r = generateTrajectory(pos, vel, samples[next_index], t);
    next_index++;
    return r;
  }
};

// ---------------------------------------------------------------------------------------------
// CostmapModel: base_local_planner/src/costmap_model.cpp:50-142 ; WorldModel::footprintCost(x,y,th,spec)
// base_local_planner/include/base_local_planner/world_model.h:65-86
// ---------------------------------------------------------------------------------------------
struct CostmapModelOracle {
  const Grid2D* cm = nullptr;
  bool allow_unknown = true;
  double pointCost(int x, int y) const {
    uint8_t cost = cm->cost(x, y);
    if (cost == LETHAL_OBSTACLE || (cost == NO_INFORMATION && !allow_unknown)) return -1;
    return cost;
  }
  double lineCost(int x0, int x1, int y0, int y1) const {
    double line_cost = 0.0;
    bool bad = false;
    lineCells(x0, y0, x1, y1, [&](int x, int y) {
      double pc = pointCost(x, y);
      if (pc < 0) {
        bad = true;
        return false;
      }
      if (line_cost < pc) line_cost = pc;
      return true;
    });
    return bad ? -1 : line_cost;
  }
  double footprintCost(const Pt2& position, const std::vector<Pt2>& fp) const {
    uint32_t cell_x, cell_y;
    if (!cm->worldToMap(position.x, position.y, cell_x, cell_y)) return -1.0;
    if (fp.size() < 3) {
      uint8_t cost = cm->cost(cell_x, cell_y);
      if (cost == LETHAL_OBSTACLE || cost == INSCRIBED_INFLATED_OBSTACLE || (cost == NO_INFORMATION && !allow_unknown)) return -1.0;
      return cost;
    }
    uint32_t x0, x1, y0, y1;
    double line_cost = 0.0, footprint_cost = 0.0;
    for (size_t i = 0; i + 1 < fp.size(); ++i) {
      if (!cm->worldToMap(fp[i].x, fp[i].y, x0, y0)) return -1.0;
      if (!cm->worldToMap(fp[i + 1].x, fp[i + 1].y, x1, y1)) return -1.0;
      line_cost = lineCost(x0, x1, y0, y1);
      footprint_cost = std::max(line_cost, footprint_cost);
      if (line_cost < 0) return -1.0;
    }
    if (!cm->worldToMap(fp.back().x, fp.back().y, x0, y0)) return -1.0;
    if (!cm->worldToMap(fp.front().x, fp.front().y, x1, y1)) return -1.0;
    line_cost = lineCost(x0, x1, y0, y1);
    footprint_cost = std::max(line_cost, footprint_cost);
    if (line_cost < 0) return -1.0;
    return footprint_cost;
  }
  double footprintCost(double x, double y, double theta, const std::vector<Pt2>& spec) const {
    double cos_th = cos(theta), sin_th = sin(theta);
    std::vector<Pt2> oriented;
    for (const Pt2& p : spec) {
      Pt2 q;
      q.x = x + (p.x * cos_th - p.y * sin_th);
      q.y = y + (p.x * sin_th + p.y * cos_th);
      oriented.push_back(q);
    }
    Pt2 pos;
    pos.x = x;
    pos.y = y;
    return footprintCost(pos, oriented);
  }
};

// base_local_planner/src/obstacle_cost_function.cpp:74-142
struct ObstacleCritic {
  const Grid2D* cm = nullptr;
  CostmapModelOracle model;
  std::vector<Pt2> footprint_spec;
  bool sum_scores = false;
  double scale = 1.0;
  double footprintCost(double x, double y, double th) const {
    double f = model.footprintCost(x, y, th, footprint_spec);
    if (f < 0) return -6.0;
    uint32_t cx, cy;
    if (!cm->worldToMap(x, y, cx, cy)) return -7.0;
    return std::max(std::max(0.0, f), double(cm->cost(cx, cy)));
  }
  double score(const Trajectory& t) const {
    double cost = 0;
    if (footprint_spec.size() == 0) return -9;
    for (size_t i = 0; i < t.x.size(); ++i) {
      double f = footprintCost(t.x[i], t.y[i], t.th[i]);
      if (f < 0) return f;
      if (sum_scores)
        cost += f;
      else
        cost = f;
    }
    return cost;
  }
};

// ---------------------------------------------------------------------------------------------
// MapGrid: base_local_planner/src/map_grid.cpp:103-310, map_grid.h:136-146, map_cell.h:44-64
// target_dist is kept as double exactly like MapCell::target_dist.
// ---------------------------------------------------------------------------------------------
struct MapGridOracle {
  uint32_t size_x = 0, size_y = 0;
  std::vector<double> dist;
  std::vector<uint8_t> mark;
  std::vector<uint8_t> within_robot;  // MapCell::within_robot; only TrajectoryPlanner::findBestPath sets it (path_map_)
  double goal_x = 0, goal_y = 0;
  bool allow_unknown = true;
  double obstacleCosts() const { return (double)dist.size(); }
  double unreachableCellCosts() const { return (double)(dist.size() + 1); }
  void sizeCheck(uint32_t sx, uint32_t sy) {
    if (dist.size() != size_t(sx) * sy) {
      dist.resize(size_t(sx) * sy);
      mark.resize(size_t(sx) * sy);
      within_robot.assign(size_t(sx) * sy, 0);
    }
    size_x = sx;
    size_y = sy;
  }
  void resetPathDist() {
    for (size_t i = 0; i < dist.size(); ++i) {
      dist[i] = unreachableCellCosts();
      mark[i] = 0;
    }
    within_robot.assign(dist.size(), 0);  // map_grid.cpp:126-132
  }
  static void adjustPlanResolution(const std::vector<Pt2>& in, std::vector<Pt2>& out, double resolution) {  // :135-171
    if (in.size() == 0) return;
    double last_x = in[0].x, last_y = in[0].y;
    out.push_back(in[0]);
    double min_sq_resolution = resolution * resolution * 4;
    for (size_t i = 1; i < in.size(); ++i) {
      double loop_x = in[i].x, loop_y = in[i].y;
      double sqdist = (loop_x - last_x) * (loop_x - last_x) + (loop_y - last_y) * (loop_y - last_y);
      if (sqdist > min_sq_resolution) {
        int steps = ((sqrt(sqdist) - sqrt(min_sq_resolution)) / resolution) - 1;
        double deltax = (loop_x - last_x) / steps;
        double deltay = (loop_y - last_y) / steps;
        for (int j = 1; j < steps; ++j) {
          Pt2 p;
          p.x = last_x + j * deltax;
          p.y = last_y + j * deltay;
          out.push_back(p);
        }
      }
      out.push_back(in[i]);
      last_x = loop_x;
      last_y = loop_y;
    }
  }
  bool updatePathCell(size_t cur, size_t chk, const Grid2D& cm) {  // :103-122
    uint8_t cost = cm.cells[chk];
    if (!(chk < within_robot.size() && within_robot[chk]) && (cost == LETHAL_OBSTACLE || cost == INSCRIBED_INFLATED_OBSTACLE || (cost == NO_INFORMATION && !allow_unknown))) {
      dist[chk] = obstacleCosts();
      return false;
    }
    double nd = dist[cur] + 1;
    if (nd < dist[chk]) dist[chk] = nd;
    return true;
  }
  void computeTargetDistance(std::queue<size_t>& q, const Grid2D& cm) {  // :262-310
    uint32_t last_col = size_x - 1, last_row = size_y - 1;
    while (!q.empty()) {
      size_t cur = q.front();
      q.pop();
      uint32_t cx = cur % size_x, cy = cur / size_x;
      auto visit = [&](size_t chk) {
        if (!mark[chk]) {
          mark[chk] = 1;
          if (updatePathCell(cur, chk, cm)) q.push(chk);
        }
      };
      if (cx > 0) visit(cur - 1);
      if (cx < last_col) visit(cur + 1);
      if (cy > 0) visit(cur - size_x);
      if (cy < last_row) visit(cur + size_x);
    }
  }
  void setTargetCells(const Grid2D& cm, const std::vector<Pt2>& plan) {  // :174-213
    sizeCheck(cm.size_x, cm.size_y);
    bool started_path = false;
    std::queue<size_t> q;
    std::vector<Pt2> adj;
    adjustPlanResolution(plan, adj, cm.resolution);
    for (size_t i = 0; i < adj.size(); ++i) {
      uint32_t mx, my;
      if (cm.worldToMap(adj[i].x, adj[i].y, mx, my) && cm.cost(mx, my) != NO_INFORMATION) {
        size_t idx = cm.index(mx, my);
        dist[idx] = 0.0;
        mark[idx] = 1;
        q.push(idx);
        started_path = true;
      } else if (started_path) {
        break;
      }
    }
    if (!started_path) return;
    computeTargetDistance(q, cm);
  }
  void setLocalGoal(const Grid2D& cm, const std::vector<Pt2>& plan) {  // :216-258
    sizeCheck(cm.size_x, cm.size_y);
    int local_goal_x = -1, local_goal_y = -1;
    bool started_path = false;
    std::vector<Pt2> adj;
    adjustPlanResolution(plan, adj, cm.resolution);
    for (size_t i = 0; i < adj.size(); ++i) {
      uint32_t mx, my;
      if (cm.worldToMap(adj[i].x, adj[i].y, mx, my) && cm.cost(mx, my) != NO_INFORMATION) {
        local_goal_x = mx;
        local_goal_y = my;
        started_path = true;
      } else {
        if (started_path) break;
      }
    }
    if (!started_path) return;
    std::queue<size_t> q;
    if (local_goal_x >= 0 && local_goal_y >= 0) {
      size_t idx = cm.index(local_goal_x, local_goal_y);
      cm.mapToWorld(local_goal_x, local_goal_y, goal_x, goal_y);
      dist[idx] = 0.0;
      mark[idx] = 1;
      q.push(idx);
    }
    computeTargetDistance(q, cm);
  }
};

// base_local_planner/src/map_grid_cost_function.cpp:59-129
enum Aggregation { AggLast = 0, AggSum = 1, AggProduct = 2 };
struct MapGridCritic {
  const Grid2D* cm = nullptr;
  MapGridOracle map;
  std::vector<Pt2> target_poses;
  double xshift = 0, yshift = 0, scale = 1.0;
  bool is_local_goal_function = false, stop_on_failure = true;
  int aggregation = AggLast;
  bool prepare() {
    map.sizeCheck(cm->size_x, cm->size_y);
    map.resetPathDist();
    if (is_local_goal_function)
      map.setLocalGoal(*cm, target_poses);
    else
      map.setTargetCells(*cm, target_poses);
    return true;
  }
  double score(const Trajectory& t) const {
    double cost = 0.0;
    if (aggregation == AggProduct) cost = 1.0;
    for (size_t i = 0; i < t.x.size(); ++i) {
      double px = t.x[i], py = t.y[i], pth = t.th[i];
      if (xshift != 0.0) {
        px = px + xshift * cos(pth);
        py = py + xshift * sin(pth);
      }
      if (yshift != 0.0) {
        px = px + yshift * cos(pth + M_PI_2);
        py = py + yshift * sin(pth + M_PI_2);
      }
      uint32_t cx, cy;
      if (!cm->worldToMap(px, py, cx, cy)) return -4.0;
      double grid_dist = map.dist[cm->index(cx, cy)];
      if (stop_on_failure) {
        if (grid_dist == map.obstacleCosts())
          return -3.0;
        else if (grid_dist == map.unreachableCellCosts())
          return -2.0;
      }
      switch (aggregation) {
        case AggLast: cost = grid_dist; break;
        case AggSum: cost += grid_dist; break;
        case AggProduct:
          if (cost > 0) cost *= grid_dist;
          break;
      }
    }
    return cost;
  }
};

// base_local_planner/src/oscillation_cost_function.cpp:56-176
struct OscillationCritic {
  bool strafe_pos_only = false, strafe_neg_only = false, strafing_pos = false, strafing_neg = false;
  bool rot_pos_only = false, rot_neg_only = false, rotating_pos = false, rotating_neg = false;
  bool forward_pos_only = false, forward_neg_only = false, forward_pos = false, forward_neg = false;
  double oscillation_reset_dist = 0.05, oscillation_reset_angle = 0.2;
  V3f prev_stationary_pos;
  double scale = 1.0;
  void resetOscillationFlags() {
    strafe_pos_only = strafe_neg_only = strafing_pos = strafing_neg = false;
    rot_pos_only = rot_neg_only = rotating_pos = rotating_neg = false;
    forward_pos_only = forward_neg_only = forward_pos = forward_neg = false;
  }
  bool setOscillationFlags(const Trajectory& t, double min_vel_trans) {
    bool flag_set = false;
    if (t.xv < 0.0) {
      if (forward_pos) {
        forward_neg_only = true;
        flag_set = true;
      }
      forward_pos = false;
      forward_neg = true;
    }
    if (t.xv > 0.0) {
      if (forward_neg) {
        forward_pos_only = true;
        flag_set = true;
      }
      forward_neg = false;
      forward_pos = true;
    }
    if (fabs(t.xv) <= min_vel_trans) {
      if (t.yv < 0) {
        if (strafing_pos) {
          strafe_neg_only = true;
          flag_set = true;
        }
        strafing_pos = false;
        strafing_neg = true;
      }
      if (t.yv > 0) {
        if (strafing_neg) {
          strafe_pos_only = true;
          flag_set = true;
        }
        strafing_neg = false;
        strafing_pos = true;
      }
      if (t.thetav < 0) {
        if (rotating_pos) {
          rot_neg_only = true;
          flag_set = true;
        }
        rotating_pos = false;
        rotating_neg = true;
      }
      if (t.thetav > 0) {
        if (rotating_neg) {
          rot_pos_only = true;
          flag_set = true;
        }
        rotating_neg = false;
        rotating_pos = true;
      }
    }
    return flag_set;
  }
  void resetOscillationFlagsIfPossible(const V3f& pos, const V3f& prev) {
    double x_diff = pos[0] - prev[0];
    double y_diff = pos[1] - prev[1];
    double sq_dist = x_diff * x_diff + y_diff * y_diff;
    double th_diff = pos[2] - prev[2];
    if (sq_dist > oscillation_reset_dist * oscillation_reset_dist || fabs(th_diff) > oscillation_reset_angle) resetOscillationFlags();
  }
  void updateOscillationFlags(const V3f& pos, const Trajectory& traj, double min_vel_trans) {
    if (traj.cost >= 0) {
      if (setOscillationFlags(traj, min_vel_trans)) prev_stationary_pos = pos;
      if (forward_pos_only || forward_neg_only || strafe_pos_only || strafe_neg_only || rot_pos_only || rot_neg_only)
        resetOscillationFlagsIfPossible(pos, prev_stationary_pos);
    }
  }
  double score(const Trajectory& t) const {
    if ((forward_pos_only && t.xv < 0.0) || (forward_neg_only && t.xv > 0.0) || (strafe_pos_only && t.yv < 0.0) ||
        (strafe_neg_only && t.yv > 0.0) || (rot_pos_only && t.thetav < 0.0) || (rot_neg_only && t.thetav > 0.0))
      return -5.0;
    return 0.0;
  }
  uint32_t packFlags() const {
    return (strafe_pos_only << 0) | (strafe_neg_only << 1) | (strafing_pos << 2) | (strafing_neg << 3) | (rot_pos_only << 4) |
           (rot_neg_only << 5) | (rotating_pos << 6) | (rotating_neg << 7) | (forward_pos_only << 8) | (forward_neg_only << 9) |
           (forward_pos << 10) | (forward_neg << 11);
  }
  void unpackFlags(uint32_t f) {
    strafe_pos_only = f & 1;
    strafe_neg_only = f & 2;
    strafing_pos = f & 4;
    strafing_neg = f & 8;
    rot_pos_only = f & 16;
    rot_neg_only = f & 32;
    rotating_pos = f & 64;
    rotating_neg = f & 128;
    forward_pos_only = f & 256;
    forward_neg_only = f & 512;
    forward_pos = f & 1024;
    forward_neg = f & 2048;
  }
};

// ---------------------------------------------------------------------------------------------
// DWAPlanner + SimpleScoredSamplingPlanner:
//   dwa_local_planner/src/dwa_planner.cpp:52-182 (wiring/scales), :213-237 (checkTrajectory),
//   :240-286 (updatePlanAndLocalCosts), :292-371 (findBestPath)
//   base_local_planner/src/simple_scored_sampling_planner.cpp:50-142
// ---------------------------------------------------------------------------------------------
struct SampleRecord {  // what the scorer saw for one sample slot (diagnostics for parity tests)
  int32_t status = 0;      // 0 = rejected by the generator, 1 = scored
  double cost_ref = -1.0;  // reference flow (with early-out against the incumbent)
  double cost_full = -1.0; // every critic summed, no early-out
  int32_t n_points = 0;
};

struct DwaPlannerOracle {
  const Grid2D* cm = nullptr;
  DwaConfig cfg;
  TrajectoryGenerator gen;
  OscillationCritic oscillation;
  ObstacleCritic obstacle;
  MapGridCritic path, goal, goal_front, alignment;
  std::vector<Pt2> global_plan;
  Trajectory result;
  std::vector<SampleRecord> records;
  int best_index = -1;

  void bind(const Grid2D* costmap) {
    cm = costmap;
    obstacle.cm = cm;
    obstacle.model.cm = cm;
    path.cm = goal.cm = goal_front.cm = alignment.cm = cm;
    goal.is_local_goal_function = true;
    goal_front.is_local_goal_function = true;
    goal_front.stop_on_failure = false;
    alignment.stop_on_failure = false;
    oscillation.resetOscillationFlags();
  }
  void reconfigure(const DwaConfig& c) {  // dwa_planner.cpp:52-116
    cfg = c;
    double resolution = cm->resolution;
    path.scale = resolution * c.path_distance_bias * 0.5;
    alignment.scale = resolution * c.path_distance_bias * 0.5;
    goal.scale = resolution * c.goal_distance_bias * 0.5;
    goal_front.scale = resolution * c.goal_distance_bias * 0.5;
    obstacle.scale = resolution * c.occdist_scale;
    oscillation.oscillation_reset_dist = c.oscillation_reset_dist;
    oscillation.oscillation_reset_angle = c.oscillation_reset_angle;
    goal_front.xshift = c.forward_point_distance;
    alignment.xshift = c.forward_point_distance;
    obstacle.sum_scores = c.sum_scores;
    obstacle.model.allow_unknown = c.allow_unknown;
    path.map.allow_unknown = goal.map.allow_unknown = goal_front.map.allow_unknown = alignment.map.allow_unknown = c.allow_unknown;
    if (cfg.vx_samples <= 0) cfg.vx_samples = 1;
    if (cfg.vy_samples <= 0) cfg.vy_samples = 1;
    if (cfg.vth_samples <= 0) cfg.vth_samples = 1;
  }
  void setPlan() { oscillation.resetOscillationFlags(); }  // dwa_planner.cpp:204-207

  void updatePlanAndLocalCosts(const V3f& pos, const std::vector<Pt2>& new_plan) {
    global_plan = new_plan;
    path.target_poses = global_plan;
    goal.target_poses = global_plan;
    Pt2 goal_pose = global_plan.back();
    double sq_dist = (pos[0] - goal_pose.x) * (pos[0] - goal_pose.x) + (pos[1] - goal_pose.y) * (pos[1] - goal_pose.y);
    std::vector<Pt2> front = global_plan;
    double angle_to_goal = atan2(goal_pose.y - pos[1], goal_pose.x - pos[0]);
    front.back().x = front.back().x + cfg.forward_point_distance * cos(angle_to_goal);
    front.back().y = front.back().y + cfg.forward_point_distance * sin(angle_to_goal);
    goal_front.target_poses = front;
    if (sq_dist > cfg.forward_point_distance * cfg.forward_point_distance * cfg.cheat_factor) {
      alignment.scale = cm->resolution * cfg.path_distance_bias * 0.5;
      alignment.target_poses = global_plan;
    } else {
      alignment.scale = 0.0;
    }
  }

  // critic order dwa_planner.cpp:167-173
  template <class F>
  void forEachCritic(F&& f) {
    f(0, oscillation.scale, [&](const Trajectory& t) { return oscillation.score(t); });
    f(1, obstacle.scale, [&](const Trajectory& t) { return obstacle.score(t); });
    f(2, goal_front.scale, [&](const Trajectory& t) { return goal_front.score(t); });
    f(3, alignment.scale, [&](const Trajectory& t) { return alignment.score(t); });
    f(4, path.scale, [&](const Trajectory& t) { return path.score(t); });
    f(5, goal.scale, [&](const Trajectory& t) { return goal.score(t); });
  }

  // simple_scored_sampling_planner.cpp:50-79.  early_out=false gives the order-independent full sum.
  double scoreTrajectory(const Trajectory& traj, double best_traj_cost, bool early_out = true) {
    double traj_cost = 0;
    bool stop = false;
    forEachCritic([&](int, double scale, auto&& score) {
      if (stop) return;
      if (scale == 0) return;
      double cost = score(traj);
      if (cost < 0) {
        traj_cost = cost;
        stop = true;
        return;
      }
      if (cost != 0) cost *= scale;
      traj_cost += cost;
      if (early_out && best_traj_cost > 0) {
        if (traj_cost > best_traj_cost) stop = true;
      }
    });
    return traj_cost;
  }

  bool prepareCritics() {  // :87-93 (all six prepare(); only the map-grid ones do work)
    goal_front.prepare();
    alignment.prepare();
    path.prepare();
    goal.prepare();
    return true;
  }

  // findBestTrajectory :81-142 with the single DWA generator, plus per-slot records
  bool findBestTrajectory(Trajectory& out, bool do_prepare = true) {
    Trajectory loop_traj, best_traj;
    double best_cost = -1;
    best_index = -1;
    if (do_prepare && !prepareCritics()) return false;
    records.assign(gen.samples.size(), SampleRecord());
    while (gen.hasMore()) {
      size_t slot = gen.next_index;
      bool ok = gen.next(loop_traj);
      if (!ok) continue;
      double c = scoreTrajectory(loop_traj, best_cost, true);
      records[slot].status = 1;
      records[slot].cost_ref = c;
      records[slot].cost_full = scoreTrajectory(loop_traj, -1, false);
      records[slot].n_points = (int)loop_traj.x.size();
      loop_traj.cost = c;
      if (c >= 0) {
        if (best_cost < 0 || c < best_cost) {
          best_cost = c;
          best_traj = loop_traj;
          best_index = (int)slot;
        }
      }
    }
    if (best_cost >= 0) {
      out.xv = best_traj.xv;
      out.yv = best_traj.yv;
      out.thetav = best_traj.thetav;
      out.cost = best_cost;
      out.x = best_traj.x;
      out.y = best_traj.y;
      out.th = best_traj.th;
    }
    return best_cost >= 0;
  }

  // findBestPath :292-371; drive velocities are (xv, yv, thetav) or zeros when cost < 0
  const Trajectory& findBestPath(const V3f& pos, const V3f& vel, const std::vector<Pt2>& footprint_spec, double drive[3],
                                 bool do_prepare = true) {
    obstacle.footprint_spec = footprint_spec;
    Pt2 gp = global_plan.back();
    V3f goal_v;
    goal_v[0] = gp.x;
    goal_v[1] = gp.y;
    goal_v[2] = 0;  // goal yaw only matters for !use_dwa distance, which ignores it
    gen.initialise(pos, vel, goal_v, cfg);
    result = Trajectory();
    result.cost = -7;
    findBestTrajectory(result, do_prepare);
    oscillation.updateOscillationFlags(pos, result, cfg.min_trans_vel);
    if (result.cost < 0) {
      drive[0] = drive[1] = drive[2] = 0;
    } else {
      drive[0] = result.xv;
      drive[1] = result.yv;
      drive[2] = result.thetav;
    }
    return result;
  }

  bool checkTrajectory(const V3f& pos, const V3f& vel, const V3f& vel_samples) {  // :213-237
    oscillation.resetOscillationFlags();
    Trajectory traj;
    Pt2 gp = global_plan.back();
    V3f goal_v;
    goal_v[0] = gp.x;
    goal_v[1] = gp.y;
    gen.initialise(pos, vel, goal_v, cfg);
    gen.generateTrajectory(pos, vel, vel_samples, traj);
    double cost = scoreTrajectory(traj, -1, true);
    return cost >= 0;
  }
};

}  // namespace oracle
