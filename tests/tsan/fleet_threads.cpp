// TEST INFRASTRUCTURE ONLY: ThreadSanitizer harness around libnavgpu's host side (see hip_host_stub.cpp).
// Thread A runs control cycles on a 3-robot fleet (stage -> costmap update -> planner cycle -> results), thread B
// reconfigures the planner between two sample counts (re-allocating the per-sample tables), thread C reconfigures the
// inflation layer, the obstacle layer and the footprint, thread D reads back state (bounds, wavefront boxes, levels).
// No application lock: the library's per-fleet mutex is all that orders them.  TSAN aborts the run on any data race.
// A cycle's result must come from ONE configuration (best_index and n_scored agree): a configure that landed inside
// a cycle's launch sequence would show as a mixed pair.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/navgpu.h"

#define CK(x) do { int rc_ = (x); if (rc_ != NAVGPU_OK) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, navgpu_last_error()); exit(2); } } while (0)

static navgpu_dwa_config cfgWith(int vx) {
  navgpu_dwa_config c;
  memset(&c, 0, sizeof(c));
  c.max_trans_vel = 0.55; c.min_trans_vel = 0.1; c.max_vel_x = 0.55; c.min_vel_x = 0.0; c.max_vel_y = 0.1; c.min_vel_y = -0.1;
  c.max_rot_vel = 1.0; c.min_rot_vel = 0.4; c.acc_lim_x = 2.5; c.acc_lim_y = 2.5; c.acc_lim_theta = 3.2;
  c.sim_time = 1.7; c.sim_granularity = 0.1; c.angular_sim_granularity = 0.1; c.sim_period = 0.05;
  c.path_distance_bias = 32; c.goal_distance_bias = 24; c.occdist_scale = 0.01; c.forward_point_distance = 0.325;
  c.oscillation_reset_dist = 0.05; c.oscillation_reset_angle = 0.2; c.vx_samples = vx; c.vy_samples = 5; c.vth_samples = 8;
  c.use_dwa = 1; c.discretize_by_time = 1; c.cheat_factor = 1.0; c.allow_unknown = 1;
  return c;
}

int main(int argc, char** argv) {
  const int cycles = argc > 1 ? atoi(argv[1]) : 300;
  const uint32_t n = 3;
  navgpu_fleet_desc d;
  memset(&d, 0, sizeof(d));
  d.n_instances = n; d.size_x = 64; d.size_y = 64; d.resolution = 0.05;
  d.layers = NAVGPU_LAYER_OBSTACLE | NAVGPU_LAYER_INFLATION;
  d.max_points = 64; d.max_observations = 2; d.max_plan = 32; d.max_footprint = 8; d.max_sim_steps = 32;
  navgpu_fleet* f = nullptr;
  CK(navgpu_fleet_create(&d, &f));
  const double fp[8] = {-0.2, -0.2, -0.2, 0.2, 0.2, 0.2, 0.2, -0.2};
  CK(navgpu_set_footprint(f, 0, n, fp, 4));
  navgpu_inflation_params ip;
  memset(&ip, 0, sizeof(ip));
  ip.enabled = 1; ip.inflation_radius = 0.55; ip.cost_scaling_factor = 10.0; ip.inscribed_radius = 0.2;
  CK(navgpu_inflation_configure(f, &ip));
  navgpu_dwa_config ca = cfgWith(6), cb = cfgWith(11);
  CK(navgpu_planner_configure(f, &ca));
  // what launch_select's stand-in reports for the two configurations
  const int32_t ms_a = (6 + 1) * (5 + 1) * (8 + 1), ms_b = (11 + 1) * (5 + 1) * (8 + 1);

  std::atomic<bool> stop{false};
  std::atomic<long> n_cfg{0}, n_aux{0}, n_read{0};
  std::thread tb([&] {
    bool which = false;
    while (!stop.load()) {
      which = !which;
      CK(navgpu_planner_configure(f, which ? &cb : &ca));
      ++n_cfg;
      std::this_thread::yield();
    }
  });
  std::thread tc([&] {
    int k = 0;
    while (!stop.load()) {
      navgpu_inflation_params q = ip;
      q.inflation_radius = (k & 1) ? 0.55 : 0.40;
      CK(navgpu_inflation_configure(f, &q));
      navgpu_obstacle_params op;
      memset(&op, 0, sizeof(op));
      op.enabled = 1; op.footprint_clearing_enabled = 1; op.combination_method = 1; op.max_obstacle_height = 2.0 + (k & 3);
      CK(navgpu_obstacle_configure(f, &op));
      CK(navgpu_set_footprint(f, 0, n, fp, 4));
      ++k;
      ++n_aux;
      std::this_thread::yield();
    }
  });
  std::thread td([&] {
    std::vector<int32_t> boxes(4 * n);
    std::vector<uint32_t> lv(3 * n);
    while (!stop.load()) {
      CK(navgpu_costmap_bounds(f, 0, n, boxes.data()));
      CK(navgpu_planner_wavefront_levels(f, 0, n, lv.data()));
      int rc = navgpu_planner_wavefront_boxes(f, 0, n, boxes.data());
      if (rc != NAVGPU_OK) exit(3);
      ++n_read;
      std::this_thread::yield();
    }
  });

  std::vector<double> poses(3 * n), plan(2 * 16 * n);
  std::vector<navgpu_robot_state> st(n);
  std::vector<navgpu_observation> obs(n);
  std::vector<float> pts(3 * 8 * n);
  std::vector<navgpu_plan_result> res(n);
  long mixed = 0;
  for (int k = 0; k < cycles; ++k) {
    for (uint32_t i = 0; i < n; ++i) {
      poses[3 * i] = 1.6 + 0.001 * k; poses[3 * i + 1] = 1.6; poses[3 * i + 2] = 0.1 * i;
      for (int j = 0; j < 16; ++j) { plan[(i * 16 + j) * 2] = 1.6 + 0.04 * j; plan[(i * 16 + j) * 2 + 1] = 1.6; }
      memset(&st[i], 0, sizeof(st[i]));
      st[i].pos[0] = (float)poses[3 * i]; st[i].pos[1] = 1.6f; st[i].pos[2] = 0.1f * i; st[i].vel[0] = 0.2f;
      st[i].plan_first = i * 16; st[i].plan_count = 16;
      memset(&obs[i], 0, sizeof(obs[i]));
      obs[i].instance = i; obs[i].first_point = 8 * i; obs[i].n_points = 8; obs[i].flags = NAVGPU_OBS_MARKING | NAVGPU_OBS_CLEARING;
      obs[i].origin_x = 1.6; obs[i].origin_y = 1.6; obs[i].origin_z = 0.3; obs[i].obstacle_range = 2.5; obs[i].raytrace_range = 3.0;
      for (int j = 0; j < 8; ++j) { pts[(8 * i + j) * 3] = 2.0f; pts[(8 * i + j) * 3 + 1] = 1.0f + 0.1f * j; pts[(8 * i + j) * 3 + 2] = 0.3f; }
    }
    CK(navgpu_costmap_stage(f, 0, n, poses.data(), obs.data(), n, pts.data(), 8 * n));
    CK(navgpu_costmap_update(f, 0, n));
    CK(navgpu_planner_stage(f, 0, n, st.data(), plan.data(), 16 * n));
    CK(navgpu_planner_cycle(f, 0, n));
    CK(navgpu_planner_results(f, 0, n, res.data()));
    for (uint32_t i = 0; i < n; ++i) {
      const bool a = res[i].best_index == ms_a && res[i].n_scored == 6, b = res[i].best_index == ms_b && res[i].n_scored == 11;
      if (!(a || b)) ++mixed;
    }
    if ((k & 7) == 0) {
      float p[9] = {1.6f, 1.6f, 0, 1.6f, 1.6f, 0.1f, 1.6f, 1.6f, 0.2f}, v[9] = {0.2f, 0, 0, 0.2f, 0, 0, 0.2f, 0, 0};
      CK(navgpu_planner_stage_poses(f, 0, n, p, v));
    }
  }
  stop.store(true);
  tb.join(); tc.join(); td.join();
  CK(navgpu_fleet_destroy(f));
  printf("cycles %d planner_configures %ld layer_configures %ld readers %ld mixed_results %ld\n", cycles, n_cfg.load(), n_aux.load(), n_read.load(), mixed);
  return mixed ? 4 : 0;
}
