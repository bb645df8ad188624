// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of the legacy base_local_planner::TrajectoryPlanner:
// base_local_planner/src/trajectory_planner.cpp (generateTrajectory :214-370, lineCost/pointCost :388-472,
// updatePlan :474-500, checkTrajectory/scoreTrajectory :502-531, createTrajectories :537-906,
// findBestPath :908-984), include/base_local_planner/trajectory_planner.h:332-374 (computeNew*),
// src/footprint_helper.cpp:51-258 (getLineCells, getFillCells, getFootprintCells).
// Pinned by base_local_planner/test/utest.cpp:75-102 (footprintObstacles: two generateTrajectory
// known answers), test/footprint_helper_test.cpp (outline / fill cells) and the MapGrid fixtures; the
// sequential selection logic of createTrajectories has no reference test — restated line by line.
// heading_scoring_ / simple_attractor_ (both default false) included: headingDiff :372-386 with the planner's own
// lineCost / pointCost (:388-472; pointCost also fails on INSCRIBED cells, unlike CostmapModel's).
#pragma once
#include <cfloat>

#include "planner_oracle.hpp"

namespace oracle {

struct TpConfig {  // BaseLocalPlanner.cfg defaults
  double acc_lim_x = 2.5, acc_lim_y = 2.5, acc_lim_theta = 3.2;
  double sim_time = 1.7, sim_granularity = 0.025, angular_sim_granularity = 0.025;
  int vx_samples = 20, vtheta_samples = 20;
  double pdist_scale = 0.6, gdist_scale = 0.8, occdist_scale = 0.01;
  double heading_lookahead = 0.325, oscillation_reset_dist = 0.05, escape_reset_dist = 0.10, escape_reset_theta = M_PI_2;
  int holonomic_robot = 1;
  double max_vel_x = 0.55, min_vel_x = 0.0, max_vel_th = 1.0, min_vel_th = -1.0, min_in_place_vel_th = 0.4;
  double backup_vel = -0.1;
  int dwa = 0;
  double sim_period = 0.05;
  int n_y_vels = 4;
  double y_vels[8] = {-0.3, -0.1, 0.1, 0.3, 0, 0, 0, 0};
  int allow_unknown = 1;
  int heading_scoring = 0, simple_attractor = 0;
  double heading_scoring_timestep = 0.1;
};

struct FpCell {  // Position2DInt
  int x, y;
};

// footprint_helper.cpp:51-124
inline void getLineCells(int x0, int x1, int y0, int y1, std::vector<FpCell>& pts) {
  lineCells(x0, y0, x1, y1, [&](int x, int y) {
    pts.push_back(FpCell{x, y});
    return true;
  });
}
// footprint_helper.cpp:127-181 (bubble sort by x, then fill each column between its extreme cells)
inline void getFillCells(std::vector<FpCell>& footprint) {
  unsigned int i = 0;
  while (i < footprint.size() - 1) {
    if (footprint[i].x > footprint[i + 1].x) {
      std::swap(footprint[i], footprint[i + 1]);
      if (i > 0) --i;
    } else {
      ++i;
    }
  }
  i = 0;
  FpCell min_pt, max_pt;
  unsigned int min_x = footprint[0].x;
  unsigned int max_x = footprint[footprint.size() - 1].x;
  for (unsigned int x = min_x; x <= max_x; ++x) {
    if (i >= footprint.size() - 1) break;
    if (footprint[i].y < footprint[i + 1].y) {
      min_pt = footprint[i];
      max_pt = footprint[i + 1];
    } else {
      min_pt = footprint[i + 1];
      max_pt = footprint[i];
    }
    i += 2;
    while (i < footprint.size() && (unsigned int)footprint[i].x == x) {
      if (footprint[i].y < min_pt.y)
        min_pt = footprint[i];
      else if (footprint[i].y > max_pt.y)
        max_pt = footprint[i];
      ++i;
    }
    for (unsigned int y = min_pt.y; y < (unsigned int)max_pt.y; ++y) footprint.push_back(FpCell{(int)x, (int)y});
  }
}
// footprint_helper.cpp:186-258
inline std::vector<FpCell> getFootprintCells(const V3f& pos, const std::vector<Pt2>& spec, const Grid2D& cm, bool fill) {
  double x_i = pos[0], y_i = pos[1], theta_i = pos[2];
  std::vector<FpCell> cells;
  if (spec.size() <= 1) {
    uint32_t mx, my;
    if (cm.worldToMap(x_i, y_i, mx, my)) cells.push_back(FpCell{(int)mx, (int)my});
    return cells;
  }
  double cos_th = cos(theta_i), sin_th = sin(theta_i);
  uint32_t x0, y0, x1, y1;
  size_t last = spec.size() - 1;
  for (size_t i = 0; i < last; ++i) {
    double nx = x_i + (spec[i].x * cos_th - spec[i].y * sin_th), ny = y_i + (spec[i].x * sin_th + spec[i].y * cos_th);
    if (!cm.worldToMap(nx, ny, x0, y0)) return cells;
    nx = x_i + (spec[i + 1].x * cos_th - spec[i + 1].y * sin_th);
    ny = y_i + (spec[i + 1].x * sin_th + spec[i + 1].y * cos_th);
    if (!cm.worldToMap(nx, ny, x1, y1)) return cells;
    getLineCells(x0, x1, y0, y1, cells);
  }
  double nx = x_i + (spec[last].x * cos_th - spec[last].y * sin_th), ny = y_i + (spec[last].x * sin_th + spec[last].y * cos_th);
  if (!cm.worldToMap(nx, ny, x0, y0)) return cells;
  nx = x_i + (spec[0].x * cos_th - spec[0].y * sin_th);
  ny = y_i + (spec[0].x * sin_th + spec[0].y * cos_th);
  if (!cm.worldToMap(nx, ny, x1, y1)) return cells;
  getLineCells(x0, x1, y0, y1, cells);
  if (fill) getFillCells(cells);
  return cells;
}

inline double tpNormalizeAngle(double a) {  // angles::normalize_angle (fmod form; package not in the reference tree)
  double r = fmod(fmod(a, 2.0 * M_PI) + 2.0 * M_PI, 2.0 * M_PI);
  if (r > M_PI) r -= 2.0 * M_PI;
  return r;
}

struct TpSampleRecord {  // every generateTrajectory call of one createTrajectories, in call order
  double vx, vy, vth, cost;
  int n_points;
};

struct TrajectoryPlannerOracle {
  const Grid2D* cm = nullptr;
  CostmapModelOracle wm;
  MapGridOracle path_map, goal_map;
  TpConfig c;
  std::vector<Pt2> footprint_spec, global_plan;
  double final_goal_x = 0, final_goal_y = 0;
  bool final_goal_position_valid = false;
  // persistent oscillation / escape state
  bool stuck_left = false, stuck_right = false, rotating_left = false, rotating_right = false;
  bool stuck_left_strafe = false, stuck_right_strafe = false, strafe_left = false, strafe_right = false;
  bool escaping = false;
  double prev_x = 0, prev_y = 0, escape_x = 0, escape_y = 0, escape_theta = 0;
  Trajectory traj_one, traj_two;
  std::vector<TpSampleRecord> records;

  void bind(const Grid2D* costmap, const TpConfig& cfg, const std::vector<Pt2>& fp) {
    cm = costmap;
    c = cfg;
    footprint_spec = fp;
    wm.cm = costmap;
    wm.allow_unknown = cfg.allow_unknown != 0;
    path_map.allow_unknown = goal_map.allow_unknown = cfg.allow_unknown != 0;
    path_map.sizeCheck(cm->size_x, cm->size_y);
    goal_map.sizeCheck(cm->size_x, cm->size_y);
    path_map.resetPathDist();
    goal_map.resetPathDist();
  }
  // trajectory_planner.h:332-374
  static double computeNewXPosition(double xi, double vx, double vy, double theta, double dt) {
    return xi + (vx * cos(theta) + vy * cos(M_PI_2 + theta)) * dt;
  }
  static double computeNewYPosition(double yi, double vx, double vy, double theta, double dt) {
    return yi + (vx * sin(theta) + vy * sin(M_PI_2 + theta)) * dt;
  }
  static double computeNewThetaPosition(double thetai, double vth, double dt) { return thetai + vth * dt; }
  static double computeNewVelocity(double vg, double vi, double a_max, double dt) {
    if ((vg - vi) >= 0) return std::min(vg, vi + a_max * dt);
    return std::max(vg, vi - a_max * dt);
  }
  void updatePlan(const std::vector<Pt2>& new_plan, bool compute_dists) {  // :474-500
    global_plan = new_plan;
    if (!global_plan.empty()) {
      final_goal_x = global_plan.back().x;
      final_goal_y = global_plan.back().y;
      final_goal_position_valid = true;
    } else {
      final_goal_position_valid = false;
    }
    if (compute_dists) {
      path_map.resetPathDist();
      goal_map.resetPathDist();
      path_map.setTargetCells(*cm, global_plan);
      goal_map.setLocalGoal(*cm, global_plan);
    }
  }
  double footprintCost(double x, double y, double th) const { return wm.footprintCost(x, y, th, footprint_spec); }  // :986-990

  void generateTrajectory(double x, double y, double theta, double vx, double vy, double vtheta, double vx_samp, double vy_samp,
                          double vtheta_samp, double acc_x, double acc_y, double acc_theta, double impossible_cost,
                          Trajectory& traj) {  // :214-370
    double x_i = x, y_i = y, theta_i = theta;
    double vx_i = vx, vy_i = vy, vtheta_i = vtheta;
    double vmag = hypot(vx_samp, vy_samp);
    int num_steps;
    if (!c.heading_scoring)
      num_steps = int(std::max((vmag * c.sim_time) / c.sim_granularity, fabs(vtheta_samp) / c.angular_sim_granularity) + 0.5);
    else
      num_steps = int(c.sim_time / c.sim_granularity + 0.5);
    if (num_steps == 0) num_steps = 1;
    double dt = c.sim_time / num_steps;
    double time = 0.0, heading_diff = 0.0;
    traj.reset();
    traj.xv = vx_samp;
    traj.yv = vy_samp;
    traj.thetav = vtheta_samp;
    traj.cost = -1.0;
    double path_dist = 0.0, goal_dist = 0.0, occ_cost = 0.0;
    for (int i = 0; i < num_steps; ++i) {
      uint32_t cell_x, cell_y;
      if (!cm->worldToMap(x_i, y_i, cell_x, cell_y)) {
        traj.cost = -1.0;
        return;
      }
      double footprint_cost = footprintCost(x_i, y_i, theta_i);
      if (footprint_cost < 0) {
        traj.cost = -1.0;
        return;
      }
      occ_cost = std::max(std::max(occ_cost, footprint_cost), double(cm->cost(cell_x, cell_y)));
      if (c.simple_attractor) {  // :310-315
        const Pt2& g = global_plan[global_plan.size() - 1];
        goal_dist = (x_i - g.x) * (x_i - g.x) + (y_i - g.y) * (y_i - g.y);
      } else {
        bool update_path_and_goal_distances = true;
        if (c.heading_scoring) {  // :321-327: path and goal distance of ONE point of the trajectory, plus its heading difference
          if (time >= c.heading_scoring_timestep && time < c.heading_scoring_timestep + dt)
            heading_diff = headingDiff(cell_x, cell_y, x_i, y_i, theta_i);
          else
            update_path_and_goal_distances = false;
        }
        if (update_path_and_goal_distances) {
          path_dist = path_map.dist[cm->index(cell_x, cell_y)];
          goal_dist = goal_map.dist[cm->index(cell_x, cell_y)];
          if (impossible_cost <= goal_dist || impossible_cost <= path_dist) {
            traj.cost = -2.0;
            return;
          }
        }
      }
      traj.x.push_back(x_i);
      traj.y.push_back(y_i);
      traj.th.push_back(theta_i);
      vx_i = computeNewVelocity(vx_samp, vx_i, acc_x, dt);
      vy_i = computeNewVelocity(vy_samp, vy_i, acc_y, dt);
      vtheta_i = computeNewVelocity(vtheta_samp, vtheta_i, acc_theta, dt);
      x_i = computeNewXPosition(x_i, vx_i, vy_i, theta_i, dt);
      y_i = computeNewYPosition(y_i, vx_i, vy_i, theta_i, dt);
      theta_i = computeNewThetaPosition(theta_i, vtheta_i, dt);
      time += dt;
    }
    if (!c.heading_scoring)
      traj.cost = c.pdist_scale * path_dist + goal_dist * c.gdist_scale + c.occdist_scale * occ_cost;
    else
      traj.cost = c.occdist_scale * occ_cost + c.pdist_scale * path_dist + 0.3 * heading_diff + goal_dist * c.gdist_scale;
  }
  // TrajectoryPlanner::pointCost / lineCost (:388-472): the planner's own ray walk; INSCRIBED fails too
  double tpPointCost(int x, int y) const {
    const uint8_t cost = cm->cost(x, y);
    if (cost == LETHAL_OBSTACLE || cost == INSCRIBED_INFLATED_OBSTACLE || (cost == NO_INFORMATION && !c.allow_unknown)) return -1;
    return cost;
  }
  double tpLineCost(int x0, int x1, int y0, int y1) const {
    double line_cost = 0.0;
    bool bad = false;
    lineCells(x0, y0, x1, y1, [&](int x, int y) {
      const double pc = tpPointCost(x, y);
      if (pc < 0) {
        bad = true;
        return false;
      }
      if (line_cost < pc) line_cost = pc;
      return true;
    });
    return bad ? -1 : line_cost;
  }
  // :372-386: the farthest plan pose with a clear line of sight from the robot's cell
  double headingDiff(int cell_x, int cell_y, double x, double y, double heading) const {
    for (int i = (int)global_plan.size() - 1; i >= 0; --i) {
      uint32_t gx_c, gy_c;
      if (cm->worldToMap(global_plan[i].x, global_plan[i].y, gx_c, gy_c)) {
        if (tpLineCost(cell_x, (int)gx_c, cell_y, (int)gy_c) >= 0) {
          double gx, gy;
          cm->mapToWorld(gx_c, gy_c, gx, gy);
          return fabs(tpNormalizeAngle(atan2(gy - y, gx - x) - heading));  // angles::shortest_angular_distance(heading, atan2(..))
        }
      }
    }
    return DBL_MAX;
  }
  double scoreTrajectory(double x, double y, double theta, double vx, double vy, double vtheta, double vx_samp, double vy_samp,
                         double vtheta_samp) {  // :518-531
    Trajectory t;
    generateTrajectory(x, y, theta, vx, vy, vtheta, vx_samp, vy_samp, vtheta_samp, c.acc_lim_x, c.acc_lim_y, c.acc_lim_theta,
                       path_map.obstacleCosts(), t);
    return t.cost;
  }
  bool checkTrajectory(double x, double y, double theta, double vx, double vy, double vtheta, double vx_samp, double vy_samp,
                       double vtheta_samp) {
    return scoreTrajectory(x, y, theta, vx, vy, vtheta, vx_samp, vy_samp, vtheta_samp) >= 0;
  }

  void resetOscillationIfMoved(double x, double y) {
    double dist = hypot(x - prev_x, y - prev_y);
    if (dist > c.oscillation_reset_dist) {
      rotating_left = rotating_right = strafe_left = strafe_right = false;
      stuck_left = stuck_right = stuck_left_strafe = stuck_right_strafe = false;
    }
  }
  void resetEscapeIfMoved(double x, double y, double theta) {
    double dist = hypot(x - escape_x, y - escape_y);
    if (dist > c.escape_reset_dist || fabs(tpNormalizeAngle(theta - escape_theta)) > c.escape_reset_theta) escaping = false;
  }

  Trajectory createTrajectories(double x, double y, double theta, double vx, double vy, double vtheta, double acc_x, double acc_y,
                                double acc_theta) {  // :537-906
    records.clear();
    double max_vel_x = c.max_vel_x, max_vel_theta, min_vel_x, min_vel_theta;
    if (final_goal_position_valid) {
      double final_goal_dist = hypot(final_goal_x - x, final_goal_y - y);
      max_vel_x = std::min(max_vel_x, final_goal_dist / c.sim_time);
    }
    if (c.dwa) {
      max_vel_x = std::max(std::min(max_vel_x, vx + acc_x * c.sim_period), c.min_vel_x);
      min_vel_x = std::max(c.min_vel_x, vx - acc_x * c.sim_period);
      max_vel_theta = std::min(c.max_vel_th, vtheta + acc_theta * c.sim_period);
      min_vel_theta = std::max(c.min_vel_th, vtheta - acc_theta * c.sim_period);
    } else {
      max_vel_x = std::max(std::min(max_vel_x, vx + acc_x * c.sim_time), c.min_vel_x);
      min_vel_x = std::max(c.min_vel_x, vx - acc_x * c.sim_time);
      max_vel_theta = std::min(c.max_vel_th, vtheta + acc_theta * c.sim_time);
      min_vel_theta = std::max(c.min_vel_th, vtheta - acc_theta * c.sim_time);
    }
    double dvx = (max_vel_x - min_vel_x) / (c.vx_samples - 1);
    double dvtheta = (max_vel_theta - min_vel_theta) / (c.vtheta_samples - 1);
    double vx_samp = min_vel_x, vtheta_samp = min_vel_theta, vy_samp = 0.0;
    Trajectory* best_traj = &traj_one;
    best_traj->cost = -1.0;
    Trajectory* comp_traj = &traj_two;
    comp_traj->cost = -1.0;
    double impossible_cost = path_map.obstacleCosts();
    auto gen = [&](double a, double b, double w) {
      generateTrajectory(x, y, theta, vx, vy, vtheta, a, b, w, acc_x, acc_y, acc_theta, impossible_cost, *comp_traj);
      records.push_back(TpSampleRecord{a, b, w, comp_traj->cost, (int)comp_traj->x.size()});
    };
    auto takeIfBetter = [&]() {
      if (comp_traj->cost >= 0 && (comp_traj->cost < best_traj->cost || best_traj->cost < 0)) std::swap(best_traj, comp_traj);
    };
    if (!escaping) {
      for (int i = 0; i < c.vx_samples; ++i) {
        vtheta_samp = 0;
        gen(vx_samp, vy_samp, vtheta_samp);
        takeIfBetter();
        vtheta_samp = min_vel_theta;
        for (int j = 0; j < c.vtheta_samples - 1; ++j) {
          gen(vx_samp, vy_samp, vtheta_samp);
          takeIfBetter();
          vtheta_samp += dvtheta;
        }
        vx_samp += dvx;
      }
      if (c.holonomic_robot) {
        vx_samp = 0.1;
        vy_samp = 0.1;
        vtheta_samp = 0.0;
        gen(vx_samp, vy_samp, vtheta_samp);
        takeIfBetter();
        vx_samp = 0.1;
        vy_samp = -0.1;
        vtheta_samp = 0.0;
        gen(vx_samp, vy_samp, vtheta_samp);
        takeIfBetter();
      }
    }
    vtheta_samp = min_vel_theta;
    vx_samp = 0.0;
    vy_samp = 0.0;
    double heading_dist = DBL_MAX;
    auto aheadDist = [&](double& out) -> bool {  // goal_map_ at the lookahead point of comp_traj's endpoint
      double x_r = comp_traj->x.back(), y_r = comp_traj->y.back(), th_r = comp_traj->th.back();  // getEndpoint
      x_r += c.heading_lookahead * cos(th_r);
      y_r += c.heading_lookahead * sin(th_r);
      uint32_t cell_x, cell_y;
      if (!cm->worldToMap(x_r, y_r, cell_x, cell_y)) return false;
      out = goal_map.dist[cm->index(cell_x, cell_y)];
      return true;
    };
    for (int i = 0; i < c.vtheta_samples; ++i) {
      double vtheta_samp_limited = vtheta_samp > 0 ? std::max(vtheta_samp, c.min_in_place_vel_th) : std::min(vtheta_samp, -1.0 * c.min_in_place_vel_th);
      gen(vx_samp, vy_samp, vtheta_samp_limited);
      if (comp_traj->cost >= 0 && (comp_traj->cost <= best_traj->cost || best_traj->cost < 0 || best_traj->yv != 0.0) &&
          (vtheta_samp > dvtheta || vtheta_samp < -1 * dvtheta)) {
        double ahead_gdist;
        if (aheadDist(ahead_gdist)) {
          if (ahead_gdist < heading_dist) {
            if (vtheta_samp < 0 && !stuck_left) {
              std::swap(best_traj, comp_traj);
              heading_dist = ahead_gdist;
            } else if (vtheta_samp > 0 && !stuck_right) {
              std::swap(best_traj, comp_traj);
              heading_dist = ahead_gdist;
            }
          }
        }
      }
      vtheta_samp += dvtheta;
    }
    if (best_traj->cost >= 0) {  // :722-768
      if (!(best_traj->xv > 0)) {
        if (best_traj->thetav < 0) {
          if (rotating_right) stuck_right = true;
          rotating_right = true;
        } else if (best_traj->thetav > 0) {
          if (rotating_left) stuck_left = true;
          rotating_left = true;
        } else if (best_traj->yv > 0) {
          if (strafe_right) stuck_right_strafe = true;
          strafe_right = true;
        } else if (best_traj->yv < 0) {
          if (strafe_left) stuck_left_strafe = true;
          strafe_left = true;
        }
        prev_x = x;
        prev_y = y;
      }
      resetOscillationIfMoved(x, y);
      resetEscapeIfMoved(x, y, theta);
      return *best_traj;
    }
    if (c.holonomic_robot) {  // :771-817
      vtheta_samp = min_vel_theta;
      vx_samp = 0.0;
      for (int i = 0; i < c.n_y_vels; ++i) {
        vtheta_samp = 0;
        vy_samp = c.y_vels[i];
        gen(vx_samp, vy_samp, vtheta_samp);
        if (comp_traj->cost >= 0 && (comp_traj->cost <= best_traj->cost || best_traj->cost < 0)) {
          double ahead_gdist;
          if (aheadDist(ahead_gdist)) {
            if (ahead_gdist < heading_dist) {
              if (vy_samp > 0 && !stuck_left_strafe) {
                std::swap(best_traj, comp_traj);
                heading_dist = ahead_gdist;
              } else if (vy_samp < 0 && !stuck_right_strafe) {
                std::swap(best_traj, comp_traj);
                heading_dist = ahead_gdist;
              }
            }
          }
        }
      }
    }
    if (best_traj->cost >= 0) {  // :820-868 (note: the flags set here differ from the block above)
      if (!(best_traj->xv > 0)) {
        if (best_traj->thetav < 0) {
          if (rotating_right) stuck_right = true;
          rotating_left = true;
        } else if (best_traj->thetav > 0) {
          if (rotating_left) stuck_left = true;
          rotating_right = true;
        } else if (best_traj->yv > 0) {
          if (strafe_right) stuck_right_strafe = true;
          strafe_left = true;
        } else if (best_traj->yv < 0) {
          if (strafe_left) stuck_left_strafe = true;
          strafe_right = true;
        }
        prev_x = x;
        prev_y = y;
      }
      resetOscillationIfMoved(x, y);
      resetEscapeIfMoved(x, y, theta);
      return *best_traj;
    }
    // :871-905 back up slowly, even when the footprint check fails
    vtheta_samp = 0.0;
    vx_samp = c.backup_vel;
    vy_samp = 0.0;
    gen(vx_samp, vy_samp, vtheta_samp);
    std::swap(best_traj, comp_traj);
    resetOscillationIfMoved(x, y);
    if (!escaping && best_traj->cost > -2.0) {
      escape_x = x;
      escape_y = y;
      escape_theta = theta;
      escaping = true;
    }
    resetEscapeIfMoved(x, y, theta);
    if (best_traj->cost == -1.0) best_traj->cost = 1.0;
    return *best_traj;
  }

  Trajectory findBestPath(const V3f& pos, const V3f& vel, double drive[3]) {  // :908-984
    path_map.resetPathDist();
    goal_map.resetPathDist();
    std::vector<FpCell> fpc = getFootprintCells(pos, footprint_spec, *cm, true);
    for (const FpCell& q : fpc) path_map.within_robot[cm->index(q.x, q.y)] = 1;
    path_map.setTargetCells(*cm, global_plan);
    goal_map.setLocalGoal(*cm, global_plan);
    Trajectory best = createTrajectories(pos[0], pos[1], pos[2], vel[0], vel[1], vel[2], c.acc_lim_x, c.acc_lim_y, c.acc_lim_theta);
    if (best.cost < 0) {
      drive[0] = drive[1] = drive[2] = 0.0;
    } else {
      drive[0] = best.xv;
      drive[1] = best.yv;
      drive[2] = best.thetav;
    }
    return best;
  }
};

}  // namespace oracle
