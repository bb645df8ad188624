// Host-side mirror of DWAPlannerROS's control cycle (include/navgpu.h, "DWAPlannerROS control cycle"): local-plan
// window, latched stop-and-rotate controller, dispatch to navgpu_planner_* — plain host arithmetic in fp64, the
// obstacle check and the DWA cycle run on the GPU through the C-ABI of navgpu_host.cpp.
#include "navgpu_fleet.h"

extern "C" {

// ------------------------------------------------------------------------------------------------ DWAPlannerROS mirror
// angles::normalize_angle_positive / normalize_angle / shortest_angular_distance (ros/angles, fmod form)
static double normalizeAnglePositive(double a) { return fmod(fmod(a, 2.0 * M_PI) + 2.0 * M_PI, 2.0 * M_PI); }
static double normalizeAngle(double a) {
  double r = normalizeAnglePositive(a);
  if (r > M_PI) r -= 2.0 * M_PI;
  return r;
}
double navgpu_shortest_angular_distance(double from, double to) { return normalizeAngle(to - from); }
static double signOf(double x) { return x < 0.0 ? -1.0 : 1.0; }  // base_local_planner sign()

int navgpu_local_plan_window(const double* plan, uint32_t n, const double pose[3], const double* T, double dist_threshold,
                             int32_t prune, double* out, uint32_t capacity, uint32_t* n_out, uint32_t* n_erased) {
  if (!plan || !pose || !out || !n_out || !n_erased) return NAVGPU_ERR_INVALID;
  *n_out = 0;
  *n_erased = 0;
  if (n == 0) return NAVGPU_ERR_INVALID;  // "Received plan with zero length" (goal_functions.cpp:98-101)
  // the robot in the frame of the plan (tf.transformPose, :113-114)
  double rx = pose[0], ry = pose[1];
  double c = 1.0, sn = 0.0;
  if (T) {
    c = cos(T[2]);
    sn = sin(T[2]);
    const double dx = pose[0] - T[0], dy = pose[1] - T[1];
    rx = c * dx + sn * dy;
    ry = -sn * dx + c * dy;
  }
  const double sq_thr = dist_threshold * dist_threshold;
  uint32_t i = 0;
  double sq_dist = 0;
  while (i < n) {  // :126-134: up to the first pose within reach
    const double xd = rx - plan[3 * i], yd = ry - plan[3 * i + 1];
    sq_dist = xd * xd + yd * yd;
    if (sq_dist <= sq_thr) break;
    ++i;
  }
  uint32_t m = 0;
  while (i < n && sq_dist <= sq_thr) {  // :140-154: the pose that leaves the reach is still taken
    if (m >= capacity) return NAVGPU_ERR_CAPACITY;
    const double px = plan[3 * i], py = plan[3 * i + 1], pth = plan[3 * i + 2];
    if (T) {
      out[3 * m] = c * px - sn * py + T[0];
      out[3 * m + 1] = sn * px + c * py + T[1];
      out[3 * m + 2] = pth + T[2];
    } else {
      out[3 * m] = px;
      out[3 * m + 1] = py;
      out[3 * m + 2] = pth;
    }
    ++m;
    const double xd = rx - px, yd = ry - py;
    sq_dist = xd * xd + yd * yd;
    ++i;
  }
  uint32_t erased = 0;
  if (prune) {  // prunePlan (:69-86): drop leading poses until one is closer than 1 m
    while (erased < m) {
      const double xd = pose[0] - out[3 * erased], yd = pose[1] - out[3 * erased + 1];
      if (xd * xd + yd * yd < 1) break;
      ++erased;
    }
    if (erased) memmove(out, out + 3 * (size_t)erased, sizeof(double) * 3 * (m - erased));
  }
  *n_out = m - erased;
  *n_erased = erased;
  return NAVGPU_OK;
}

int navgpu_local_planner_configure(navgpu_fleet* f, const navgpu_local_limits* lim) {
  if (!f || !lim) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  f->lp_limits = *lim;
  f->lp_configured = true;
  if (f->lp.size() != f->desc.n_instances) f->lp.assign(f->desc.n_instances, navgpu_fleet::LocalPlannerState());
  return NAVGPU_OK;
}

int navgpu_local_planner_set_plan(navgpu_fleet* f, uint32_t instance, const double* plan, uint32_t n, const double* T) {
  if (!f || instance >= f->desc.n_instances || (n && !plan)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  if (!f->lp_configured || !f->planner_configured) return NAVGPU_ERR_STATE;
  navgpu_fleet::LocalPlannerState& st = f->lp[instance];
  st.xy_tolerance_latch = false;  // latchedStopRotateController_.resetLatching() (dwa_planner_ros.cpp:136)
  st.plan.assign(plan, plan + 3 * (size_t)n);
  st.have_plan = true;
  st.has_T = T != nullptr;
  if (T) memcpy(st.T, T, sizeof(st.T));
  return navgpu_planner_set_plan(f, instance, 1);  // DWAPlanner::setPlan: resetOscillationFlags
}

int navgpu_local_planner_get_plan(navgpu_fleet* f, uint32_t instance, double* xyyaw, uint32_t capacity) {
  if (!f || instance >= f->desc.n_instances || !f->lp_configured) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  const std::vector<double>& pl = f->lp[instance].plan;
  const uint32_t n = (uint32_t)(pl.size() / 3);
  if (xyyaw) {
    if (capacity < n) return NAVGPU_ERR_CAPACITY;
    memcpy(xyyaw, pl.data(), sizeof(double) * pl.size());
  }
  return (int)n;
}

// getGoalPose (goal_functions.cpp:175-214): the last pose of the stored plan in the global frame
static bool goalPose(const navgpu_fleet::LocalPlannerState& st, double goal[3]) {
  if (st.plan.empty()) return false;
  const double* g = &st.plan[st.plan.size() - 3];
  if (st.has_T) {
    const double c = cos(st.T[2]), sn = sin(st.T[2]);
    goal[0] = c * g[0] - sn * g[1] + st.T[0];
    goal[1] = sn * g[0] + c * g[1] + st.T[1];
    goal[2] = g[2] + st.T[2];
  } else {
    goal[0] = g[0];
    goal[1] = g[1];
    goal[2] = g[2];
  }
  return true;
}
static bool stoppedOdom(const double v[3], double rot_stopped, double trans_stopped) {  // goal_functions.cpp:248-253
  return fabs(v[2]) <= rot_stopped && fabs(v[0]) <= trans_stopped && fabs(v[1]) <= trans_stopped;
}

int navgpu_local_planner_is_goal_reached(navgpu_fleet* f, uint32_t first, uint32_t count, const navgpu_robot_input* in, int32_t* reached) {
  if (!f || !in || !reached || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  if (!f->lp_configured) return NAVGPU_ERR_STATE;
  const navgpu_local_limits& lim = f->lp_limits;
  for (uint32_t k = 0; k < count; ++k) {
    navgpu_fleet::LocalPlannerState& st = f->lp[first + k];
    reached[k] = 0;
    double goal[3];
    if (!in[k].have_pose || !goalPose(st, goal)) continue;
    // LatchedStopRotateController::isGoalReached (:66-109)
    const double dist = hypot(goal[0] - in[k].pose[0], goal[1] - in[k].pose[1]);
    if ((lim.latch_xy_goal_tolerance && st.xy_tolerance_latch) || dist <= lim.xy_goal_tolerance) {
      if (lim.latch_xy_goal_tolerance && !st.xy_tolerance_latch) st.xy_tolerance_latch = true;
      const double angle = navgpu_shortest_angular_distance(in[k].pose[2], goal[2]);
      if (fabs(angle) <= lim.yaw_goal_tolerance && stoppedOdom(in[k].odom_vel, lim.rot_stopped_vel, lim.trans_stopped_vel)) reached[k] = 1;
    }
  }
  return NAVGPU_OK;
}

int navgpu_local_planner_compute_velocity_commands(navgpu_fleet* f, uint32_t first, uint32_t count, const navgpu_robot_input* in,
                                                   navgpu_cmd_result* out) {
  if (!f || !in || !out || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  if (!f->lp_configured || !f->planner_configured) return NAVGPU_ERR_STATE;
  const navgpu_local_limits& lim = f->lp_limits;
  const uint32_t max_plan = f->pl.max_plan;
  const double dist_threshold = std::max(f->cm.nx * f->cm.res / 2.0, f->cm.ny * f->cm.res / 2.0);
  std::vector<double> local((size_t)count * max_plan * 3), packed;
  std::vector<navgpu_robot_state> states(count);
  std::vector<uint8_t> valid(count, 0), dwa(count, 0);
  packed.reserve((size_t)count * max_plan * 2);
  // --- getLocalPlan per robot (computeVelocityCommands :254-271)
  for (uint32_t k = 0; k < count; ++k) {
    navgpu_fleet::LocalPlannerState& st = f->lp[first + k];
    navgpu_cmd_result& o = out[k];
    o = navgpu_cmd_result();
    if (!in[k].have_pose) continue;                    // "Could not get robot pose"
    if (!st.have_plan || st.plan.empty()) continue;    // transformGlobalPlan: "Received plan with zero length"
    uint32_t n_loc = 0, n_er = 0;
    int rc = navgpu_local_plan_window(st.plan.data(), (uint32_t)(st.plan.size() / 3), in[k].pose, st.has_T ? st.T : nullptr,
                                      dist_threshold, lim.prune_plan, &local[(size_t)k * max_plan * 3], max_plan, &n_loc, &n_er);
    if (rc == NAVGPU_ERR_CAPACITY) return rc;
    if (rc != NAVGPU_OK) continue;
    if (n_er) st.plan.erase(st.plan.begin(), st.plan.begin() + 3 * (size_t)std::min<size_t>(n_er, st.plan.size() / 3));
    o.local_plan_points = (int32_t)n_loc;
    if (n_loc == 0) continue;                          // "Received an empty transformed plan."
    valid[k] = 1;
    navgpu_robot_state& rs = states[k];
    for (int a = 0; a < 3; ++a) {
      rs.pos[a] = (float)in[k].pose[a];                // Eigen::Vector3f pos / vel (dwa_planner.cpp:303-304)
      rs.vel[a] = (float)in[k].odom_vel[a];
    }
    rs.plan_first = (uint32_t)(packed.size() / 2);
    rs.plan_count = n_loc;
    for (uint32_t q = 0; q < n_loc; ++q) {
      packed.push_back(local[((size_t)k * max_plan + q) * 3]);
      packed.push_back(local[((size_t)k * max_plan + q) * 3 + 1]);
    }
  }
  // --- updatePlanAndLocalCosts for every robot that has a local plan (:274), in contiguous runs
  for (uint32_t k = 0; k < count;) {
    if (!valid[k]) {
      ++k;
      continue;
    }
    uint32_t e = k;
    while (e < count && valid[e]) ++e;
    int rc = navgpu_planner_stage(f, first + k, e - k, &states[k], packed.data(), (uint32_t)(packed.size() / 2));
    if (rc != NAVGPU_OK) return rc;
    k = e;
  }
  // --- dispatch: isPositionReached (latched_stop_rotate_controller.cpp:37-58)
  for (uint32_t k = 0; k < count; ++k) {
    if (!valid[k]) continue;
    navgpu_fleet::LocalPlannerState& st = f->lp[first + k];
    double goal[3];
    bool reached = false;
    if (goalPose(st, goal)) {
      const double dist = hypot(goal[0] - in[k].pose[0], goal[1] - in[k].pose[1]);
      if ((lim.latch_xy_goal_tolerance && st.xy_tolerance_latch) || dist <= lim.xy_goal_tolerance) {
        st.xy_tolerance_latch = true;
        reached = true;
      }
    }
    if (!reached) {
      dwa[k] = 1;
      out[k].branch = NAVGPU_BRANCH_DWA;
      continue;
    }
    // computeVelocityCommandsStopRotate (:211-273)
    navgpu_cmd_result& o = out[k];
    if (!goalPose(st, goal)) continue;  // "Could not get goal pose"
    if (lim.latch_xy_goal_tolerance && !st.xy_tolerance_latch) st.xy_tolerance_latch = true;
    const double yaw = in[k].pose[2], vel_yaw = in[k].odom_vel[2];
    const double angle = navgpu_shortest_angular_distance(yaw, goal[2]);
    if (fabs(angle) <= lim.yaw_goal_tolerance) {
      o.cmd_vel[0] = o.cmd_vel[1] = o.cmd_vel[2] = 0.0;
      st.rotating_to_goal = false;
      o.ok = 1;
      o.branch = NAVGPU_BRANCH_AT_GOAL;
      continue;
    }
    const double acc[3] = {lim.acc_lim_x, lim.acc_lim_y, lim.acc_lim_theta};
    float vs[3];
    int32_t okc = 0;
    if (!st.rotating_to_goal && !stoppedOdom(in[k].odom_vel, lim.rot_stopped_vel, lim.trans_stopped_vel)) {
      // stopWithAccLimits (:111-146); Eigen::Vector3f narrows the samples to float
      const double vx = signOf(in[k].odom_vel[0]) * std::max(0.0, fabs(in[k].odom_vel[0]) - acc[0] * lim.sim_period);
      const double vy = signOf(in[k].odom_vel[1]) * std::max(0.0, fabs(in[k].odom_vel[1]) - acc[1] * lim.sim_period);
      const double vth = signOf(vel_yaw) * std::max(0.0, fabs(vel_yaw) - acc[2] * lim.sim_period);
      vs[0] = (float)vx;
      vs[1] = (float)vy;
      vs[2] = (float)vth;
      int rc = navgpu_planner_check_trajectory(f, first + k, vs, &okc);
      if (rc != NAVGPU_OK) return rc;
      o.branch = NAVGPU_BRANCH_STOP;
      if (okc) {
        o.cmd_vel[0] = vx;
        o.cmd_vel[1] = vy;
        o.cmd_vel[2] = vth;
        o.ok = 1;
      }  // else: zeros, "Error when stopping." -> false
    } else {
      // rotateToGoal (:148-209)
      st.rotating_to_goal = true;
      const double ang_diff = angle;
      double v = std::min(lim.max_rot_vel, std::max(lim.min_rot_vel, fabs(ang_diff)));
      const double max_acc_vel = fabs(vel_yaw) + acc[2] * lim.sim_period;
      const double min_acc_vel = fabs(vel_yaw) - acc[2] * lim.sim_period;
      v = std::min(std::max(fabs(v), min_acc_vel), max_acc_vel);
      const double max_speed_to_stop = sqrt(2 * acc[2] * fabs(ang_diff));
      v = std::min(max_speed_to_stop, fabs(v));
      v = std::min(lim.max_rot_vel, std::max(lim.min_rot_vel, v));
      if (ang_diff < 0) v = -v;
      vs[0] = 0.f;
      vs[1] = 0.f;
      vs[2] = (float)v;
      int rc = navgpu_planner_check_trajectory(f, first + k, vs, &okc);
      if (rc != NAVGPU_OK) return rc;
      o.branch = NAVGPU_BRANCH_ROTATE;
      if (okc) {
        o.cmd_vel[2] = v;
        o.ok = 1;
      }  // else: "Rotation cmd in collision" -> zeros, false
    }
  }
  // --- dwaComputeVelocityCommands (:176-247) for the others, in contiguous runs
  std::vector<navgpu_plan_result> res(count);
  for (uint32_t k = 0; k < count;) {
    if (!dwa[k]) {
      ++k;
      continue;
    }
    uint32_t e = k;
    while (e < count && dwa[e]) ++e;
    int rc = navgpu_planner_cycle(f, first + k, e - k);
    if (rc != NAVGPU_OK) return rc;
    rc = navgpu_planner_results(f, first + k, e - k, &res[k]);
    if (rc != NAVGPU_OK) return rc;
    for (uint32_t q = k; q < e; ++q) {
      navgpu_cmd_result& o = out[q];
      o.cmd_vel[0] = res[q].drive[0];
      o.cmd_vel[1] = res[q].drive[1];
      o.cmd_vel[2] = res[q].drive[2];
      o.ok = res[q].cost >= 0 ? 1 : 0;  // path.cost_ < 0: "failed to find a valid plan"
      o.trajectory_points = o.ok ? res[q].n_points : 0;
    }
    k = e;
  }
  return NAVGPU_OK;
}

}  // extern "C"
