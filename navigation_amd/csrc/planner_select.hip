// k_select (first-strict-minimum selection, winner trajectory, oscillation flag update), k_cell_costs, k_stage_poses,
// k_sincos (gfx950).
#include "planner_common.h"

namespace navgpu {

// ------------------------------------------------------------------------------------------------
// k_cell_costs: DWAPlanner::getCellCosts (dwa_planner.cpp:185-202) for every cell — what MapGridVisualizer's
// cost cloud shows.  out[cell] = {path_cost, goal_cost, occ_cost, total_cost} as floats, total = NaN where the
// reference returns false (path cost obstacle / unreachable, or an inscribed-or-worse cell).
// ------------------------------------------------------------------------------------------------
__global__ void k_cell_costs(PlannerDev pl, uint32_t inst, float4* out) {
  const uint32_t cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= pl.cells) return;
  const float path_cost = (float)(double)pl.path[(size_t)inst * pl.cells + cell];
  const float goal_cost = (float)(double)pl.goal[(size_t)inst * pl.cells + cell];
  const float occ_cost = (float)pl.master[(size_t)inst * pl.cells_padded + cell];
  float total = __builtin_nanf("");
  if (!(path_cost == (double)pl.cells || path_cost == (double)(pl.cells + 1) || occ_cost >= (float)kInscribed)) {
    const double resolution = pl.res;
    total = (float)(pl.cfg.path_distance_bias * resolution * path_cost + pl.cfg.goal_distance_bias * resolution * goal_cost +
                    pl.cfg.occdist_scale * occ_cost);
  }
  out[cell] = make_float4(path_cost, goal_cost, occ_cost, total);
}
void launch_cell_costs(const PlannerDev& pl, uint32_t inst, float4* out, hipStream_t s) {
  hipLaunchKernelGGL(k_cell_costs, dim3((pl.cells + 255) / 256), dim3(256), 0, s, pl, inst, out);
}

// ------------------------------------------------------------------------------------------------
// k_select: the tail of SimpleScoredSamplingPlanner::findBestTrajectory (:111-135) and of
// DWAPlanner::findBestPath (dwa_planner.cpp:316,357-368): pick the first strict minimum, rebuild
// the winner's points, run OscillationCostFunction::updateOscillationFlags
// (oscillation_cost_function.cpp:56-164), fill drive velocities.
// ------------------------------------------------------------------------------------------------
constexpr int kSelectSteps = 128;  // steps whose trigonometry k_select computes side by side (longer trajectories: one lane)
__global__ __launch_bounds__(64) void k_select(PlannerDev pl, uint32_t first, uint32_t n_blocks) {
  const uint32_t inst = first + blockIdx.x;
  const uint32_t tid = threadIdx.x;
  const navgpu_dwa_config& c = pl.cfg;
  double bc = 1.0e300;
  int bi = 0x7FFFFFFF;
  for (uint32_t k = tid; k < n_blocks; k += 64) {
    double oc = pl.part_cost[(size_t)inst * pl.score_blocks + k];
    int oi = pl.part_index[(size_t)inst * pl.score_blocks + k];
    if (oc < bc || (oc == bc && oi < bi)) {
      bc = oc;
      bi = oi;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    double oc = __shfl_down(bc, off);
    int oi = __shfl_down(bi, off);
    if (oc < bc || (oc == bc && oi < bi)) {
      bc = oc;
      bi = oi;
    }
  }
  bc = __shfl(bc, 0);
  bi = __shfl(bi, 0);
  // The winner's points.  A step's position needs the step before it, but its heading only the headings before it:
  // lane 0 runs the (cheap) velocity / heading recurrences, every lane then takes the sines and cosines of some steps,
  // lane 0 adds the positions up - the two dozen fp64 sincos calls in a row were most of this kernel's 15 us.
  __shared__ float s_lv[kSelectSteps][3];
  __shared__ float s_pth[kSelectSteps];
  __shared__ double s_sc[kSelectSteps][4];
  const navgpu_robot_state st = pl.state[inst];
  const int32_t* cnt = pl.axis_count + 4 * inst;
  float sel_vs[3] = {0.f, 0.f, 0.f}, sel_lv0[3] = {0.f, 0.f, 0.f};
  double sel_dt = 0.0;
  int sel_steps = 0;
  if (bi != 0x7FFFFFFF) {  // (uniform)
    const int nth = cnt[2], nyv = cnt[1];
    const int ix = bi / (nyv * nth), rem = bi - ix * (nyv * nth);
    const int iy = rem / nth, ith = rem - iy * nth;
    sel_vs[0] = pl.axis_samples[((size_t)inst * 3 + 0) * pl.max_axis + ix];
    sel_vs[1] = pl.axis_samples[((size_t)inst * 3 + 1) * pl.max_axis + iy];
    sel_vs[2] = pl.axis_samples[((size_t)inst * 3 + 2) * pl.max_axis + ith];
    const double vmag = hyp2((double)sel_vs[0], (double)sel_vs[1]);
    double ns;
    if (c.discretize_by_time)
      ns = ceil(c.sim_time / c.sim_granularity);
    else
      ns = ceil(fmax(vmag * c.sim_time / c.sim_granularity, fabs((double)sel_vs[2]) * c.sim_time / c.angular_sim_granularity));
    sel_steps = (int)ns;
    if (sel_steps > (int)pl.max_sim_steps) sel_steps = (int)pl.max_sim_steps;
    sel_dt = c.sim_time / sel_steps;
    const bool continued = !c.use_dwa;
    const float acc[3] = {(float)c.acc_lim_x, (float)c.acc_lim_y, (float)c.acc_lim_theta};
    auto newVel = [&](const float* vel_in, float* out) {
      for (int i = 0; i < 3; ++i) {
        if (vel_in[i] < sel_vs[i])
          out[i] = (float)fmin((double)sel_vs[i], vel_in[i] + acc[i] * sel_dt);
        else
          out[i] = (float)fmax((double)sel_vs[i], vel_in[i] - acc[i] * sel_dt);
      }
    };
    float lv[3] = {sel_vs[0], sel_vs[1], sel_vs[2]};
    if (continued) {
      float t0[3];
      newVel(st.vel, t0);
      lv[0] = t0[0];
      lv[1] = t0[1];
      lv[2] = t0[2];
    }
    sel_lv0[0] = lv[0];
    sel_lv0[1] = lv[1];
    sel_lv0[2] = lv[2];
    if (sel_steps <= kSelectSteps) {
      if (tid == 0) {
        float pth = st.pos[2];
        for (int step = 0; step < sel_steps; ++step) {
          s_pth[step] = pth;
          if (continued) {
            float t1[3];
            newVel(lv, t1);
            lv[0] = t1[0];
            lv[1] = t1[1];
            lv[2] = t1[2];
          }
          s_lv[step][0] = lv[0];
          s_lv[step][1] = lv[1];
          s_lv[step][2] = lv[2];
          pth = (float)(pth + lv[2] * sel_dt);
        }
      }
      __syncthreads();
      for (int step = (int)tid; step < sel_steps; step += 64) {
        const double th = s_pth[step];
        double sn, cs, sn2 = 0.0, cs2 = 0.0;
        sincos(th, &sn, &cs);
        if (s_lv[step][1] != 0.0f) sincos(M_PI_2 + th, &sn2, &cs2);
        s_sc[step][0] = cs;
        s_sc[step][1] = sn;
        s_sc[step][2] = cs2;
        s_sc[step][3] = sn2;
      }
      __syncthreads();
    }
  }
  if (tid != 0) return;
  navgpu_plan_result r;
  r.n_samples = cnt[3];
  r.n_scored = pl.counters[2 * inst];
  r.n_valid = pl.counters[2 * inst + 1];
  r.reserved = 0;
  double* tr = pl.traj + (size_t)inst * pl.max_sim_steps * 3;
  uint32_t flags = pl.osc_flags[inst];
  if (bi == 0x7FFFFFFF) {
    r.best_index = -1;
    r.n_points = 0;
    r.xv = r.yv = r.thetav = 0.f;
    r.cost = -7.0;  // result_traj_.cost_ pre-set (dwa_planner.cpp:316)
    r.drive[0] = r.drive[1] = r.drive[2] = 0.0;
  } else {
    const int num_steps = sel_steps;
    const double dt = sel_dt;
    r.xv = sel_lv0[0];
    r.yv = sel_lv0[1];
    r.thetav = sel_lv0[2];
    float px = st.pos[0], py = st.pos[1], pth = st.pos[2];
    if (num_steps <= kSelectSteps) {
      for (int step = 0; step < num_steps; ++step) {
        tr[3 * step] = px;
        tr[3 * step + 1] = py;
        tr[3 * step + 2] = s_pth[step];
        const float lx = s_lv[step][0], ly = s_lv[step][1];
        const double tx = c.rollout_trig ? (double)(lx * (float)s_sc[step][0]) : lx * s_sc[step][0];  // (navgpu_dwa_config::rollout_trig)
        const double ty = c.rollout_trig ? (double)(lx * (float)s_sc[step][1]) : lx * s_sc[step][1];
        const float nxp = (float)(px + (tx + ly * s_sc[step][2]) * dt);
        const float nyp = (float)(py + (ty + ly * s_sc[step][3]) * dt);
        px = nxp;
        py = nyp;
      }
    } else {  // (more steps than the shared tables hold: the plain sequential form)
      const bool continued = !c.use_dwa;
      const float acc[3] = {(float)c.acc_lim_x, (float)c.acc_lim_y, (float)c.acc_lim_theta};
      float lv[3] = {sel_lv0[0], sel_lv0[1], sel_lv0[2]};
      auto newVel = [&](const float* vel_in, float* out) {
        for (int i = 0; i < 3; ++i) {
          if (vel_in[i] < sel_vs[i])
            out[i] = (float)fmin((double)sel_vs[i], vel_in[i] + acc[i] * dt);
          else
            out[i] = (float)fmax((double)sel_vs[i], vel_in[i] - acc[i] * dt);
        }
      };
      for (int step = 0; step < num_steps; ++step) {
        tr[3 * step] = px;
        tr[3 * step + 1] = py;
        tr[3 * step + 2] = pth;
        if (continued) {
          float t1[3];
          newVel(lv, t1);
          lv[0] = t1[0];
          lv[1] = t1[1];
          lv[2] = t1[2];
        }
        const double th = pth;
        double sn, cs, sn2 = 0.0, cs2 = 0.0;
        sincos(th, &sn, &cs);
        if (lv[1] != 0.0f) sincos(M_PI_2 + th, &sn2, &cs2);
        const double tx = c.rollout_trig ? (double)(lv[0] * (float)cs) : lv[0] * cs, ty = c.rollout_trig ? (double)(lv[0] * (float)sn) : lv[0] * sn;
        const float nxp = (float)(px + (tx + lv[1] * cs2) * dt);
        const float nyp = (float)(py + (ty + lv[1] * sn2) * dt);
        const float ntp = (float)(pth + lv[2] * dt);
        px = nxp;
        py = nyp;
        pth = ntp;
      }
    }
    r.best_index = bi;
    r.n_points = num_steps;
    r.cost = bc;
    r.drive[0] = r.xv;
    r.drive[1] = r.yv;
    r.drive[2] = r.thetav;
    // ---- updateOscillationFlags(pos, &result_traj_, min_trans_vel)
    const double xv = r.xv, yv = r.yv, thv = r.thetav;
    bool flag_set = false;
    auto has = [&](uint32_t b) { return (flags & b) != 0; };
    auto set = [&](uint32_t b, bool v) { flags = v ? (flags | b) : (flags & ~b); };
    if (xv < 0.0) {
      if (has(NAVGPU_OSC_FORWARD_POS)) {
        set(NAVGPU_OSC_FORWARD_NEG_ONLY, true);
        flag_set = true;
      }
      set(NAVGPU_OSC_FORWARD_POS, false);
      set(NAVGPU_OSC_FORWARD_NEG, true);
    }
    if (xv > 0.0) {
      if (has(NAVGPU_OSC_FORWARD_NEG)) {
        set(NAVGPU_OSC_FORWARD_POS_ONLY, true);
        flag_set = true;
      }
      set(NAVGPU_OSC_FORWARD_NEG, false);
      set(NAVGPU_OSC_FORWARD_POS, true);
    }
    if (fabs(xv) <= c.min_trans_vel) {
      if (yv < 0) {
        if (has(NAVGPU_OSC_STRAFING_POS)) {
          set(NAVGPU_OSC_STRAFE_NEG_ONLY, true);
          flag_set = true;
        }
        set(NAVGPU_OSC_STRAFING_POS, false);
        set(NAVGPU_OSC_STRAFING_NEG, true);
      }
      if (yv > 0) {
        if (has(NAVGPU_OSC_STRAFING_NEG)) {
          set(NAVGPU_OSC_STRAFE_POS_ONLY, true);
          flag_set = true;
        }
        set(NAVGPU_OSC_STRAFING_NEG, false);
        set(NAVGPU_OSC_STRAFING_POS, true);
      }
      if (thv < 0) {
        if (has(NAVGPU_OSC_ROTATING_POS)) {
          set(NAVGPU_OSC_ROT_NEG_ONLY, true);
          flag_set = true;
        }
        set(NAVGPU_OSC_ROTATING_POS, false);
        set(NAVGPU_OSC_ROTATING_NEG, true);
      }
      if (thv > 0) {
        if (has(NAVGPU_OSC_ROTATING_NEG)) {
          set(NAVGPU_OSC_ROT_POS_ONLY, true);
          flag_set = true;
        }
        set(NAVGPU_OSC_ROTATING_NEG, false);
        set(NAVGPU_OSC_ROTATING_POS, true);
      }
    }
    float* prev = pl.osc_prev + 3 * inst;
    if (flag_set) {
      prev[0] = st.pos[0];
      prev[1] = st.pos[1];
      prev[2] = st.pos[2];
    }
    const uint32_t only = NAVGPU_OSC_FORWARD_POS_ONLY | NAVGPU_OSC_FORWARD_NEG_ONLY | NAVGPU_OSC_STRAFE_POS_ONLY |
                          NAVGPU_OSC_STRAFE_NEG_ONLY | NAVGPU_OSC_ROT_POS_ONLY | NAVGPU_OSC_ROT_NEG_ONLY;
    if (flags & only) {  // resetOscillationFlagsIfPossible (:71-82): float differences widened to double
      const double x_diff = st.pos[0] - prev[0];
      const double y_diff = st.pos[1] - prev[1];
      const double sq_dist = x_diff * x_diff + y_diff * y_diff;
      const double th_diff = st.pos[2] - prev[2];
      if (sq_dist > c.oscillation_reset_dist * c.oscillation_reset_dist || fabs(th_diff) > c.oscillation_reset_angle) flags = 0;
    }
  }
  r.oscillation_flags = flags;
  pl.osc_flags[inst] = flags;
  pl.result[inst] = r;
}
void launch_select(const PlannerDev& pl, uint32_t first, uint32_t count, uint32_t n_blocks, hipStream_t s) {
  hipLaunchKernelGGL(k_select, dim3(count), dim3(64), 0, s, pl, first, n_blocks);
}

// ------------------------------------------------------------------------------------------------
// k_stage_poses (navgpu_planner_stage_poses): the poses of a cycle travel as KERNEL ARGUMENTS, 64 robots (3.5 KB) per
// launch - the runtime copies the argument block when the launch is queued, so there is no staging buffer whose reuse
// the host would have to wait for.  (An H2D hipMemcpyAsync of this size was measured to block until the stream had
// drained, an event wait to cost 2 ms while timing events are recorded in the same process, and a host spin on a
// device-written flag to starve the queue.)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_stage_poses(PoseChunk c) {
  const uint32_t li = threadIdx.x;
  if (li >= c.count) return;
  const uint32_t i = c.first + li;
  c.state[i] = c.st[li];
  c.front_last[2 * i] = c.front[2 * li];
  c.front_last[2 * i + 1] = c.front[2 * li + 1];
  c.align_on[i] = c.align[li];
  c.bfs_reach[i] = c.reach[li];
}
void launch_stage_poses(const PoseChunk& c, hipStream_t s) { hipLaunchKernelGGL(k_stage_poses, dim3(1), dim3(64), 0, s, c); }

// sin / cos of the headings as score_body evaluates them (navgpu_device_sincos: the floating-point contract, checkable)
__global__ void k_sincos(const double* th, uint32_t n, double* sn, double* cs) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sincos(th[i], &sn[i], &cs[i]);
}
void launch_sincos(const double* th, uint32_t n, double* sn, double* cs, hipStream_t s) {
  hipLaunchKernelGGL(k_sincos, dim3((n + 255) / 256), dim3(256), 0, s, th, n, sn, cs);
}

}  // namespace navgpu
