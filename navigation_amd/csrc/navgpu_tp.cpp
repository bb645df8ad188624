// Host side of the legacy base_local_planner::TrajectoryPlanner (include/navgpu.h, navgpu_tp_*): footprint cells
// under the robot, velocity window and sample enumeration, launch of the wavefronts and of k_tp_rollout, and the
// replay of createTrajectories' sequential, stateful selection over the per-sample results.
#include "navgpu_fleet.h"

extern "C" {

// ------------------------------------------------------------------------------------------------ legacy TrajectoryPlanner
namespace {
struct TpCell {
  int x, y;
};
// Costmap2D::worldToMap on the host (costmap_2d.cpp:208-220)
bool hostWorldToMap(double ox, double oy, double res, uint32_t nx, uint32_t ny, double wx, double wy, uint32_t& mx, uint32_t& my) {
  if (wx < ox || wy < oy) return false;
  const double fx = (wx - ox) / res, fy = (wy - oy) / res;
  if (!(fx < 2147483648.0) || !(fy < 2147483648.0)) return false;
  mx = (uint32_t)(int)fx;
  my = (uint32_t)(int)fy;
  return mx < nx && my < ny;
}
// FootprintHelper::getLineCells (footprint_helper.cpp:51-124)
void tpLineCells(int x0, int x1, int y0, int y1, std::vector<TpCell>& pts) {
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
  for (int curpixel = 0; curpixel <= numpixels; curpixel++) {
    pts.push_back(TpCell{x, y});
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
// FootprintHelper::getFillCells (:127-181)
void tpFillCells(std::vector<TpCell>& fp) {
  unsigned int i = 0;
  while (i < fp.size() - 1) {
    if (fp[i].x > fp[i + 1].x) {
      std::swap(fp[i], fp[i + 1]);
      if (i > 0) --i;
    } else {
      ++i;
    }
  }
  i = 0;
  TpCell min_pt, max_pt;
  const unsigned int min_x = fp[0].x, max_x = fp[fp.size() - 1].x;
  for (unsigned int x = min_x; x <= max_x; ++x) {
    if (i >= fp.size() - 1) break;
    if (fp[i].y < fp[i + 1].y) {
      min_pt = fp[i];
      max_pt = fp[i + 1];
    } else {
      min_pt = fp[i + 1];
      max_pt = fp[i];
    }
    i += 2;
    while (i < fp.size() && (unsigned int)fp[i].x == x) {
      if (fp[i].y < min_pt.y)
        min_pt = fp[i];
      else if (fp[i].y > max_pt.y)
        max_pt = fp[i];
      ++i;
    }
    for (unsigned int y = min_pt.y; y < (unsigned int)max_pt.y; ++y) fp.push_back(TpCell{(int)x, (int)y});
  }
}
// FootprintHelper::getFootprintCells(pos, spec, costmap, fill = true) (:186-258)
void tpFootprintCells(const float pos[3], const double* spec, uint32_t nfp, double ox, double oy, double res, uint32_t nx, uint32_t ny,
                      std::vector<TpCell>& cells) {
  const double x_i = pos[0], y_i = pos[1], theta_i = pos[2];
  cells.clear();
  if (nfp <= 1) {
    uint32_t mx, my;
    if (hostWorldToMap(ox, oy, res, nx, ny, x_i, y_i, mx, my)) cells.push_back(TpCell{(int)mx, (int)my});
    return;
  }
  const double cos_th = cos(theta_i), sin_th = sin(theta_i);
  uint32_t x0, y0, x1, y1;
  const uint32_t last = nfp - 1;
  auto vertex = [&](uint32_t i, uint32_t& mx, uint32_t& my) {
    const double wx = x_i + (spec[2 * i] * cos_th - spec[2 * i + 1] * sin_th);
    const double wy = y_i + (spec[2 * i] * sin_th + spec[2 * i + 1] * cos_th);
    return hostWorldToMap(ox, oy, res, nx, ny, wx, wy, mx, my);
  };
  for (uint32_t i = 0; i < last; ++i) {
    if (!vertex(i, x0, y0)) return;
    if (!vertex(i + 1, x1, y1)) return;
    tpLineCells((int)x0, (int)x1, (int)y0, (int)y1, cells);
  }
  if (!vertex(last, x0, y0)) return;
  if (!vertex(0, x1, y1)) return;
  tpLineCells((int)x0, (int)x1, (int)y0, (int)y1, cells);
  tpFillCells(cells);
}
// createTrajectories' sample enumeration (:537-665,777-780,871-874): every generateTrajectory call the
// reference could make this cycle, with the stage it belongs to
struct TpPlanned {
  double vx, vy, vth;
  double vth_unlimited;  // stage C: the loop variable before the min_in_place clamp
  int stage;             // 0 forward grid, 1 holonomic pair, 2 in-place rotation, 3 y velocities, 4 backing up
};
}  // namespace

static uint32_t tpMaxSamples(const navgpu_tp_config& c) {
  return (uint32_t)(c.vx_samples * c.vtheta_samples + 2 + c.vtheta_samples + c.n_y_vels + 1);
}

int navgpu_tp_configure(navgpu_fleet* f, const navgpu_tp_config* cfg_in) {
  if (!f || !cfg_in) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  navgpu_tp_config c = *cfg_in;
  if (c.n_y_vels < 0 || c.n_y_vels > 8 || !(c.sim_time > 0) || !(c.sim_granularity > 0) || !(c.angular_sim_granularity > 0))
    return NAVGPU_ERR_INVALID;
  if (c.vx_samples <= 0) c.vx_samples = 1;          // trajectory_planner.cpp:98-107
  if (c.vtheta_samples <= 0) c.vtheta_samples = 1;
  // step capacity: num_steps = int(max(vmag * sim_time / sim_granularity, |vtheta| / angular_sim_granularity) + 0.5)
  double ymax = 0.1;
  for (int i = 0; i < c.n_y_vels; ++i) ymax = std::max(ymax, fabs(c.y_vels[i]));
  const double vxmax = std::max(std::max(fabs(c.max_vel_x), fabs(c.min_vel_x)), std::max(fabs(c.backup_vel), 0.1));
  const double wmax = std::max(std::max(fabs(c.max_vel_th), fabs(c.min_vel_th)), fabs(c.min_in_place_vel_th));
  const double steps = (c.heading_scoring ? c.sim_time / c.sim_granularity  // (heading scoring: int(sim_time / sim_granularity + 0.5) steps, :240-244)
                                          : std::max(hypot(vxmax, ymax) * c.sim_time / c.sim_granularity, wmax / c.angular_sim_granularity)) + 1.5;
  if (steps > (double)f->pl.max_sim_steps) {
    g_last_error = "navgpu_tp_configure: trajectories need more points than max_sim_steps";
    return NAVGPU_ERR_CAPACITY;
  }
  HIP_TRY(waitStream(f->stream));
  TpDev& tp = f->tp;
  const uint32_t n = f->desc.n_instances;
  const uint32_t ms = tpMaxSamples(c);
  if (!f->tp_configured || ms > tp.max_samples) {
    f->release(tp.samples);
    f->release(tp.out);
    tp.samples = nullptr;
    tp.out = nullptr;
    int rc = f->alloc(&tp.samples, (size_t)n * ms * 3);
    if (rc) return rc;
    rc = f->alloc(&tp.out, (size_t)n * ms);
    if (rc) return rc;
    tp.max_samples = ms;
  }
  if (!f->tp_configured) {
    const uint32_t W = (f->cm.nx + 31) / 32;
    int rc = f->alloc(&tp.n_samples, n);
    if (!rc) rc = f->alloc(&tp.start, (size_t)n * 6);
    if (!rc) rc = f->alloc(&tp.winner, n);
    if (!rc) rc = f->alloc(&tp.points, (size_t)n * f->pl.max_sim_steps * 3);
    if (!rc) rc = f->alloc(&tp.within_count, n);
    if (!rc) rc = f->alloc(&tp.within_bits, (size_t)n * f->cm.ny * W);
    if (rc) return rc;
    f->tph.assign(n, navgpu_fleet::TpHost());
  }
  tp.cfg = c;
  f->tp_h_samples.assign((size_t)n * tp.max_samples * 3, 0.0);
  f->tp_h_out.assign((size_t)n * tp.max_samples, TpOut());
  f->tp_h_start.assign((size_t)n * 6, 0.0);
  f->tp_h_nsamples.assign(n, 0);
  f->tp_h_within_count.assign(n, 0);
  f->tp_h_winner.assign(n, -1);
  f->tp_configured = true;
  HIP_TRY(waitStream(f->stream));
  return NAVGPU_OK;
}

// the wavefront launch of the legacy planner: two grids, path_map_ optionally with within_robot bits
static int tpLaunchGrids(navgpu_fleet* f, uint32_t first, uint32_t count, bool with_within) {
  PlannerDev pl = f->pl;
  pl.bfs_grids = 2;
  pl.within = with_within ? f->tp.within_bits : nullptr;
  pl.cfg.allow_unknown = f->tp.cfg.allow_unknown;
  pl.bfs_bounded = 0;  // the legacy planner's look-ups are not confined to a box around the robot
  for (uint32_t i = first; i < first + count; ++i) f->grid_partial[i] = 0;
  PROFILED(f, NAVGPU_K_BFS, launch_bfs(pl, first, count, f->stream));
  return NAVGPU_OK;
}
static int tpUploadPlans(navgpu_fleet* f, uint32_t first, uint32_t count) {
  PlannerDev& pl = f->pl;
  for (uint32_t i = first; i < first + count; ++i) {
    const std::vector<double>& p = f->tph[i].plan;
    const uint32_t np = (uint32_t)(p.size() / 2);
    if (np > pl.max_plan) return NAVGPU_ERR_CAPACITY;
    if (np) memcpy(&f->hp_plan[(size_t)i * pl.max_plan * 2], p.data(), sizeof(double) * p.size());
    f->hp_plan_cnt[i] = np;
  }
  HIP_TRY(hipMemcpyAsync(pl.plan + (size_t)first * pl.max_plan * 2, f->hp_plan + (size_t)first * pl.max_plan * 2,
                         sizeof(double) * 2 * (size_t)count * pl.max_plan, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(pl.plan_count + first, f->hp_plan_cnt + first, sizeof(uint32_t) * count, hipMemcpyHostToDevice, f->stream));
  return NAVGPU_OK;
}

int navgpu_tp_update_plan(navgpu_fleet* f, uint32_t instance, const double* plan_xy, uint32_t n, int32_t compute_dists) {
  if (!f || instance >= f->desc.n_instances || (n && !plan_xy)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  if (!f->tp_configured) return NAVGPU_ERR_STATE;
  if (n > f->pl.max_plan) return NAVGPU_ERR_CAPACITY;
  navgpu_fleet::TpHost& h = f->tph[instance];
  h.plan.assign(plan_xy, plan_xy + 2 * (size_t)n);
  if (n) {  // :480-487
    h.final_goal_x = plan_xy[2 * (size_t)(n - 1)];
    h.final_goal_y = plan_xy[2 * (size_t)(n - 1) + 1];
    h.final_goal_position_valid = true;
  } else {
    h.final_goal_position_valid = false;
  }
  if (compute_dists) {  // :489-499 (resetPathDist clears within_robot)
    int rc = tpUploadPlans(f, instance, 1);
    if (rc) return rc;
    rc = tpLaunchGrids(f, instance, 1, false);
    if (rc) return rc;
    HIP_TRY(waitStream(f->stream));
    return checkLaunch();
  }
  return NAVGPU_OK;
}

static inline bool tpFlag(const navgpu_tp_state& s, uint32_t bit) { return (s.flags & bit) != 0; }
static inline void tpSet(navgpu_tp_state& s, uint32_t bit, bool v) { s.flags = v ? (s.flags | bit) : (s.flags & ~bit); }

int navgpu_tp_find_best_path(navgpu_fleet* f, uint32_t first, uint32_t count, const navgpu_robot_state* states, navgpu_tp_result* results) {
  if (!f || !states || !results || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  if (!f->tp_configured) return NAVGPU_ERR_STATE;
  TpDev& tp = f->tp;
  const navgpu_tp_config& c = tp.cfg;
  const CostmapDev& cm = f->cm;
  const uint32_t ms = tp.max_samples;
  std::vector<std::vector<TpPlanned>> planned(count);
  std::vector<double> dvth(count, 0.0);
  std::vector<std::vector<TpCell>> cells(count);
  size_t max_cells = 0;
  // ---- host: footprint cells under the robot, velocity window, sample enumeration
  for (uint32_t k = 0; k < count; ++k) {
    const uint32_t inst = first + k;
    navgpu_fleet::TpHost& h = f->tph[inst];
    if (c.simple_attractor && h.plan.empty()) {  // the reference indexes global_plan_[size() - 1] (:311-314)
      g_last_error = "navgpu_tp_find_best_path: simple_attractor needs a plan";
      return NAVGPU_ERR_STATE;
    }
    const float* pos = states[k].pos;
    const float* vel = states[k].vel;
    tpFootprintCells(pos, &f->h_fp_spec[(size_t)inst * kMaxFootprint * 2], f->h_fp_n[inst], f->h_origin[2 * inst],
                     f->h_origin[2 * inst + 1], cm.res, cm.nx, cm.ny, cells[k]);
    max_cells = std::max(max_cells, cells[k].size());
    const double x = pos[0], y = pos[1], vx = vel[0], vtheta = vel[2];
    const double acc_x = c.acc_lim_x, acc_theta = c.acc_lim_theta;
    double* st = &f->tp_h_start[(size_t)inst * 6];
    st[0] = x;
    st[1] = y;
    st[2] = pos[2];
    st[3] = vx;
    st[4] = vel[1];
    st[5] = vtheta;
    // :540-566
    double max_vel_x = c.max_vel_x, max_vel_theta, min_vel_x, min_vel_theta;
    if (h.final_goal_position_valid) {
      const double final_goal_dist = hypot(h.final_goal_x - x, h.final_goal_y - y);
      max_vel_x = std::min(max_vel_x, final_goal_dist / c.sim_time);
    }
    const double horizon = c.dwa ? c.sim_period : c.sim_time;
    max_vel_x = std::max(std::min(max_vel_x, vx + acc_x * horizon), c.min_vel_x);
    min_vel_x = std::max(c.min_vel_x, vx - acc_x * horizon);
    max_vel_theta = std::min(c.max_vel_th, vtheta + acc_theta * horizon);
    min_vel_theta = std::max(c.min_vel_th, vtheta - acc_theta * horizon);
    const double dvx = (max_vel_x - min_vel_x) / (c.vx_samples - 1);
    const double dvtheta = (max_vel_theta - min_vel_theta) / (c.vtheta_samples - 1);
    dvth[k] = dvtheta;
    std::vector<TpPlanned>& P = planned[k];
    double vx_samp = min_vel_x, vtheta_samp = min_vel_theta, vy_samp = 0.0;
    if (!tpFlag(h.st, NAVGPU_TP_ESCAPING)) {
      for (int i = 0; i < c.vx_samples; ++i) {  // :584-611
        vtheta_samp = 0;
        P.push_back(TpPlanned{vx_samp, vy_samp, vtheta_samp, 0, 0});
        vtheta_samp = min_vel_theta;
        for (int j = 0; j < c.vtheta_samples - 1; ++j) {
          P.push_back(TpPlanned{vx_samp, vy_samp, vtheta_samp, 0, 0});
          vtheta_samp += dvtheta;
        }
        vx_samp += dvx;
      }
      if (c.holonomic_robot) {  // :614-644
        P.push_back(TpPlanned{0.1, 0.1, 0.0, 0, 1});
        P.push_back(TpPlanned{0.1, -0.1, 0.0, 0, 1});
      }
    }
    vtheta_samp = min_vel_theta;  // :648-720
    for (int i = 0; i < c.vtheta_samples; ++i) {
      const double lim = vtheta_samp > 0 ? std::max(vtheta_samp, c.min_in_place_vel_th) : std::min(vtheta_samp, -1.0 * c.min_in_place_vel_th);
      P.push_back(TpPlanned{0.0, 0.0, lim, vtheta_samp, 2});
      vtheta_samp += dvtheta;
    }
    if (c.holonomic_robot)  // :771-817
      for (int i = 0; i < c.n_y_vels; ++i) P.push_back(TpPlanned{0.0, c.y_vels[i], 0.0, 0, 3});
    P.push_back(TpPlanned{c.backup_vel, 0.0, 0.0, 0, 4});  // :871-876
    if (P.size() > ms) return NAVGPU_ERR_CAPACITY;
    f->tp_h_nsamples[inst] = (uint32_t)P.size();
    for (size_t q = 0; q < P.size(); ++q) {
      double* d = &f->tp_h_samples[((size_t)inst * ms + q) * 3];
      d[0] = P[q].vx;
      d[1] = P[q].vy;
      d[2] = P[q].vth;
    }
  }
  // ---- within_robot cells (capacity grows with the footprint)
  if (max_cells > tp.max_within || !tp.within_cells) {
    HIP_TRY(waitStream(f->stream));
    f->release(tp.within_cells);
    tp.within_cells = nullptr;
    tp.max_within = (uint32_t)std::max<size_t>(max_cells * 2, 256);
    int rc = f->alloc(&tp.within_cells, (size_t)f->desc.n_instances * tp.max_within);
    if (rc) return rc;
    f->tp_h_within.assign((size_t)f->desc.n_instances * tp.max_within, 0);
  }
  for (uint32_t k = 0; k < count; ++k) {
    const uint32_t inst = first + k;
    f->tp_h_within_count[inst] = (uint32_t)cells[k].size();
    for (size_t q = 0; q < cells[k].size(); ++q)
      f->tp_h_within[(size_t)inst * tp.max_within + q] = (uint32_t)cells[k][q].y * cm.nx + (uint32_t)cells[k][q].x;
  }
  // ---- H2D + wavefronts + rollout of every planned sample
  int rc = tpUploadPlans(f, first, count);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(tp.within_cells + (size_t)first * tp.max_within, &f->tp_h_within[(size_t)first * tp.max_within],
                         sizeof(uint32_t) * (size_t)count * tp.max_within, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(tp.within_count + first, &f->tp_h_within_count[first], sizeof(uint32_t) * count, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(tp.samples + (size_t)first * ms * 3, &f->tp_h_samples[(size_t)first * ms * 3], sizeof(double) * 3 * (size_t)count * ms,
                         hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(tp.n_samples + first, &f->tp_h_nsamples[first], sizeof(uint32_t) * count, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(tp.start + (size_t)first * 6, &f->tp_h_start[(size_t)first * 6], sizeof(double) * 6 * count, hipMemcpyHostToDevice, f->stream));
  launch_tp_within(f->pl, tp, first, count, f->stream);
  rc = tpLaunchGrids(f, first, count, true);
  if (rc) return rc;
  PROFILED(f, NAVGPU_K_SCORE, launch_tp_rollout(f->pl, tp, first, count, 0, f->stream));
  HIP_TRY(hipMemcpyAsync(&f->tp_h_out[(size_t)first * ms], tp.out + (size_t)first * ms, sizeof(TpOut) * (size_t)count * ms, hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(waitStream(f->stream));
  rc = checkLaunch();
  if (rc) return rc;
  // ---- host: the reference's sequential selection (createTrajectories :568-905) over the per-sample results
  for (uint32_t k = 0; k < count; ++k) {
    const uint32_t inst = first + k;
    navgpu_fleet::TpHost& h = f->tph[inst];
    navgpu_tp_state& S = h.st;
    const std::vector<TpPlanned>& P = planned[k];
    const TpOut* O = &f->tp_h_out[(size_t)inst * ms];
    const double x = states[k].pos[0], y = states[k].pos[1], theta = states[k].pos[2];
    const double dvtheta = dvth[k];
    h.made.clear();
    int best = -1;        // index into P of best_traj (-1: the initial best_traj with cost -1)
    int best_made = -1;   // its position among the calls actually made
    double best_cost = -1.0, best_xv = 0, best_yv = 0, best_thv = 0;
    auto made = [&](size_t q) {
      h.made.push_back(navgpu_tp_sample{P[q].vx, P[q].vy, P[q].vth, O[q].cost, O[q].n_points, 0});
    };
    auto take = [&](size_t q) {  // always called right after made(q)
      best = (int)q;
      best_made = (int)h.made.size() - 1;
      best_cost = O[q].cost;
      best_xv = P[q].vx;
      best_yv = P[q].vy;
      best_thv = P[q].vth;
    };
    size_t q = 0;
    for (; q < P.size() && P[q].stage <= 1; ++q) {  // forward grid and the two holonomic samples: strict improvement
      made(q);
      if (O[q].cost >= 0 && (O[q].cost < best_cost || best_cost < 0)) take(q);
    }
    double heading_dist = DBL_MAX;
    for (; q < P.size() && P[q].stage == 2; ++q) {  // in-place rotations :654-719
      made(q);
      const double vtheta_samp = P[q].vth_unlimited;
      if (O[q].cost >= 0 && (O[q].cost <= best_cost || best_cost < 0 || best_yv != 0.0) &&
          (vtheta_samp > dvtheta || vtheta_samp < -1 * dvtheta)) {
        if (O[q].ahead_ok) {
          const double ahead_gdist = O[q].ahead;
          if (ahead_gdist < heading_dist) {
            if (vtheta_samp < 0 && !tpFlag(S, NAVGPU_TP_STUCK_LEFT)) {
              take(q);
              heading_dist = ahead_gdist;
            } else if (vtheta_samp > 0 && !tpFlag(S, NAVGPU_TP_STUCK_RIGHT)) {
              take(q);
              heading_dist = ahead_gdist;
            }
          }
        }
      }
    }
    auto resetOscillationIfMoved = [&]() {
      const double dist = hypot(x - S.prev_x, y - S.prev_y);
      if (dist > c.oscillation_reset_dist)
        S.flags &= ~(NAVGPU_TP_ROTATING_LEFT | NAVGPU_TP_ROTATING_RIGHT | NAVGPU_TP_STRAFE_LEFT | NAVGPU_TP_STRAFE_RIGHT |
                     NAVGPU_TP_STUCK_LEFT | NAVGPU_TP_STUCK_RIGHT | NAVGPU_TP_STUCK_LEFT_STRAFE | NAVGPU_TP_STUCK_RIGHT_STRAFE);
    };
    auto resetEscapeIfMoved = [&]() {
      const double dist = hypot(x - S.escape_x, y - S.escape_y);
      if (dist > c.escape_reset_dist || fabs(navgpu_shortest_angular_distance(S.escape_theta, theta)) > c.escape_reset_theta)
        tpSet(S, NAVGPU_TP_ESCAPING, false);
    };
    bool finished = false;
    if (best_cost >= 0) {  // :722-768
      if (!(best_xv > 0)) {
        if (best_thv < 0) {
          if (tpFlag(S, NAVGPU_TP_ROTATING_RIGHT)) tpSet(S, NAVGPU_TP_STUCK_RIGHT, true);
          tpSet(S, NAVGPU_TP_ROTATING_RIGHT, true);
        } else if (best_thv > 0) {
          if (tpFlag(S, NAVGPU_TP_ROTATING_LEFT)) tpSet(S, NAVGPU_TP_STUCK_LEFT, true);
          tpSet(S, NAVGPU_TP_ROTATING_LEFT, true);
        } else if (best_yv > 0) {
          if (tpFlag(S, NAVGPU_TP_STRAFE_RIGHT)) tpSet(S, NAVGPU_TP_STUCK_RIGHT_STRAFE, true);
          tpSet(S, NAVGPU_TP_STRAFE_RIGHT, true);
        } else if (best_yv < 0) {
          if (tpFlag(S, NAVGPU_TP_STRAFE_LEFT)) tpSet(S, NAVGPU_TP_STUCK_LEFT_STRAFE, true);
          tpSet(S, NAVGPU_TP_STRAFE_LEFT, true);
        }
        S.prev_x = x;
        S.prev_y = y;
      }
      resetOscillationIfMoved();
      resetEscapeIfMoved();
      finished = true;
    }
    if (!finished) {
      for (; q < P.size() && P[q].stage == 3; ++q) {  // sideways :771-817
        made(q);
        const double vy_samp = P[q].vy;
        if (O[q].cost >= 0 && (O[q].cost <= best_cost || best_cost < 0)) {
          if (O[q].ahead_ok) {
            const double ahead_gdist = O[q].ahead;
            if (ahead_gdist < heading_dist) {
              if (vy_samp > 0 && !tpFlag(S, NAVGPU_TP_STUCK_LEFT_STRAFE)) {
                take(q);
                heading_dist = ahead_gdist;
              } else if (vy_samp < 0 && !tpFlag(S, NAVGPU_TP_STUCK_RIGHT_STRAFE)) {
                take(q);
                heading_dist = ahead_gdist;
              }
            }
          }
        }
      }
      if (best_cost >= 0) {  // :820-868 — the flags set here are not those of the block above
        if (!(best_xv > 0)) {
          if (best_thv < 0) {
            if (tpFlag(S, NAVGPU_TP_ROTATING_RIGHT)) tpSet(S, NAVGPU_TP_STUCK_RIGHT, true);
            tpSet(S, NAVGPU_TP_ROTATING_LEFT, true);
          } else if (best_thv > 0) {
            if (tpFlag(S, NAVGPU_TP_ROTATING_LEFT)) tpSet(S, NAVGPU_TP_STUCK_LEFT, true);
            tpSet(S, NAVGPU_TP_ROTATING_RIGHT, true);
          } else if (best_yv > 0) {
            if (tpFlag(S, NAVGPU_TP_STRAFE_RIGHT)) tpSet(S, NAVGPU_TP_STUCK_RIGHT_STRAFE, true);
            tpSet(S, NAVGPU_TP_STRAFE_LEFT, true);
          } else if (best_yv < 0) {
            if (tpFlag(S, NAVGPU_TP_STRAFE_LEFT)) tpSet(S, NAVGPU_TP_STUCK_LEFT_STRAFE, true);
            tpSet(S, NAVGPU_TP_STRAFE_RIGHT, true);
          }
          S.prev_x = x;
          S.prev_y = y;
        }
        resetOscillationIfMoved();
        resetEscapeIfMoved();
        finished = true;
      }
    }
    if (!finished) {  // :871-905 back up slowly, whatever the footprint check says
      while (q < P.size() && P[q].stage != 4) ++q;
      made(q);
      take(q);
      resetOscillationIfMoved();
      if (!tpFlag(S, NAVGPU_TP_ESCAPING) && best_cost > -2.0) {
        S.escape_x = x;
        S.escape_y = y;
        S.escape_theta = theta;
        tpSet(S, NAVGPU_TP_ESCAPING, true);
      }
      resetEscapeIfMoved();
      if (best_cost == -1.0) best_cost = 1.0;
    }
    navgpu_tp_result& r = results[k];
    r = navgpu_tp_result();
    r.n_samples = (int32_t)h.made.size();
    r.cost = best_cost;
    if (best >= 0) {
      r.xv = best_xv;
      r.yv = best_yv;
      r.thetav = best_thv;
      r.n_points = O[best].n_points;
      r.best_sample = best_made;
    } else {  // the initial best_traj: Trajectory() with cost -1 and no points (cannot happen: backing up always takes)
      r.best_sample = -1;
    }
    h.n_points = r.n_points;
    if (r.cost >= 0) {  // findBestPath :969-978
      r.drive[0] = r.xv;
      r.drive[1] = r.yv;
      r.drive[2] = r.thetav;
    }
    f->tp_h_winner[inst] = best;
  }
  // ---- second pass: the winner's points (published as the local plan)
  HIP_TRY(hipMemcpyAsync(tp.winner + first, &f->tp_h_winner[first], sizeof(int32_t) * count, hipMemcpyHostToDevice, f->stream));
  launch_tp_rollout(f->pl, tp, first, count, 1, f->stream);
  HIP_TRY(waitStream(f->stream));
  return checkLaunch();
}

int navgpu_tp_trajectory(navgpu_fleet* f, uint32_t instance, double* xyth, uint32_t cap) {
  if (!f || !xyth || instance >= f->desc.n_instances) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  if (!f->tp_configured) return NAVGPU_ERR_STATE;
  const int n = f->tph[instance].n_points;
  if (n > (int)cap) return NAVGPU_ERR_CAPACITY;
  if (n > 0) {
    HIP_TRY(hipMemcpyAsync(xyth, f->tp.points + (size_t)instance * f->pl.max_sim_steps * 3, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, f->stream));
    HIP_TRY(waitStream(f->stream));
  }
  return n;
}

int navgpu_tp_samples(navgpu_fleet* f, uint32_t instance, navgpu_tp_sample* samples, uint32_t cap) {
  if (!f || instance >= f->desc.n_instances) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  if (!f->tp_configured) return NAVGPU_ERR_STATE;
  const std::vector<navgpu_tp_sample>& m = f->tph[instance].made;
  if (samples) {
    if (m.size() > cap) return NAVGPU_ERR_CAPACITY;
    if (!m.empty()) memcpy(samples, m.data(), sizeof(navgpu_tp_sample) * m.size());
  }
  return (int)m.size();
}

int navgpu_tp_score_trajectory(navgpu_fleet* f, uint32_t instance, const double pose[3], const double vel[3], const double vs[3], double* cost) {
  if (!f || !pose || !vel || !vs || !cost || instance >= f->desc.n_instances) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  if (!f->tp_configured) return NAVGPU_ERR_STATE;
  TpDev& tp = f->tp;
  if (tp.cfg.simple_attractor && f->tph[instance].plan.empty()) return NAVGPU_ERR_STATE;
  const uint32_t ms = tp.max_samples;
  const double start[6] = {pose[0], pose[1], pose[2], vel[0], vel[1], vel[2]};
  const uint32_t one = 1;
  HIP_TRY(hipMemcpyAsync(tp.start + (size_t)instance * 6, start, sizeof(start), hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(tp.samples + (size_t)instance * ms * 3, vs, sizeof(double) * 3, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(tp.n_samples + instance, &one, sizeof(one), hipMemcpyHostToDevice, f->stream));
  HIP_TRY(waitStream(f->stream));  // the sources above live on this stack frame
  launch_tp_rollout(f->pl, tp, instance, 1, 0, f->stream);
  TpOut o;
  HIP_TRY(hipMemcpyAsync(&o, tp.out + (size_t)instance * ms, sizeof(o), hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(waitStream(f->stream));
  *cost = o.cost;
  return checkLaunch();
}

int navgpu_tp_get_state(navgpu_fleet* f, uint32_t first, uint32_t count, navgpu_tp_state* states) {
  if (!f || !states || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  if (!f->tp_configured) return NAVGPU_ERR_STATE;
  for (uint32_t k = 0; k < count; ++k) states[k] = f->tph[first + k].st;
  return NAVGPU_OK;
}
int navgpu_tp_set_state(navgpu_fleet* f, uint32_t first, uint32_t count, const navgpu_tp_state* states) {
  if (!f || !states || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  if (!f->tp_configured) return NAVGPU_ERR_STATE;
  for (uint32_t k = 0; k < count; ++k) f->tph[first + k].st = states[k];
  return NAVGPU_OK;
}

}  // extern "C"
