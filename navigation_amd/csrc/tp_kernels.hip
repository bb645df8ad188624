// HIP kernels (gfx950) for the legacy base_local_planner::TrajectoryPlanner:
//   k_tp_within  : MapCell::within_robot bits of path_map_ from the footprint cells the host rasterised
//                  (FootprintHelper::getFootprintCells with fill, trajectory_planner.cpp:918-930)
//   k_tp_rollout : TrajectoryPlanner::generateTrajectory (trajectory_planner.cpp:214-370), one lane per call of
//                  createTrajectories, fp64 state as in the reference; second pass stores the winner's points
// The two MapGrid wavefronts are the k_bfs* kernels of planner_kernels.hip (bfs_grids = 2, `within` set).
// Compiled with -ffp-contract=off.
#include "navgpu_device.h"

namespace navgpu {

__global__ void k_tp_within(PlannerDev pl, TpDev tp, uint32_t first) {
  const uint32_t inst = first + blockIdx.y;
  const uint32_t W = (pl.nx + 31) >> 5;
  const uint32_t n = tp.within_count[inst];
  uint32_t* bits = tp.within_bits + (size_t)inst * pl.ny * W;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t cell = tp.within_cells[(size_t)inst * tp.max_within + i];
    const uint32_t my = cell / pl.nx, mx = cell - my * pl.nx;
    atomicOr(&bits[my * W + (mx >> 5)], 1u << (mx & 31));
  }
}

void launch_tp_within(const PlannerDev& pl, const TpDev& tp, uint32_t first, uint32_t count, hipStream_t s) {
  const uint32_t W = (pl.nx + 31) / 32;
  hipMemsetAsync(tp.within_bits + (size_t)first * pl.ny * W, 0, sizeof(uint32_t) * (size_t)count * pl.ny * W, s);
  hipLaunchKernelGGL(k_tp_within, dim3(4, count), dim3(256), 0, s, pl, tp, first);
}

// CostmapModel::pointCost / lineCost / footprintCost (costmap_model.cpp:50-142) on the global costmap
struct TpWorld {
  const uint8_t* master;
  Geom g;
  bool allow_unknown;
  __device__ double pointCost(int x, int y) const {
    const uint8_t cost = master[(uint32_t)y * g.nx + (uint32_t)x];
    if (cost == kLethal || (cost == kNoInfo && !allow_unknown)) return -1;
    return cost;
  }
  // The cells of base_local_planner::LineIterator (line_iterator.h:38-139) from (x0, y0) to (x1, y1), both ends included:
  // the longer axis advances with every cell, the shorter one whenever the running remainder - which starts at half the
  // long extent - passes it.  visit(x, y) returns false to end the walk; the return value says whether it ran to the end.
  template <class Visit>
  __device__ static bool forEachLineCell(int x0, int y0, int x1, int y1, Visit&& visit) {
    const int ex = x1 >= x0 ? x1 - x0 : x0 - x1, ey = y1 >= y0 ? y1 - y0 : y0 - y1;
    const int sx = x1 >= x0 ? 1 : -1, sy = y1 >= y0 ? 1 : -1;
    const bool along_x = ex >= ey;
    const int long_ext = along_x ? ex : ey, short_ext = along_x ? ey : ex;
    int rem = long_ext / 2, x = x0, y = y0;
    for (int k = 0; k <= long_ext; ++k) {
      if (!visit(x, y)) return false;
      rem += short_ext;
      const bool side = rem >= long_ext;
      if (side) rem -= long_ext;
      x += along_x ? sx : (side ? sx : 0);
      y += along_x ? (side ? sy : 0) : sy;
    }
    return true;
  }
  // maximum cost over the line's cells, -1 as soon as one of them fails `fails(cost)`
  template <class Fails>
  __device__ double lineMax(int x0, int y0, int x1, int y1, Fails&& fails) const {
    double worst = 0.0;
    const bool clear = forEachLineCell(x0, y0, x1, y1, [&](int x, int y) {
      const uint8_t cost = master[(uint32_t)y * g.nx + (uint32_t)x];
      if (fails(cost)) return false;
      if (worst < (double)cost) worst = (double)cost;
      return true;
    });
    return clear ? worst : -1.0;
  }
  // CostmapModel::lineCost (costmap_model.cpp:104-125): a cell fails like pointCost
  __device__ double lineCost(int x0, int x1, int y0, int y1) const {
    return lineMax(x0, y0, x1, y1, [&](uint8_t cost) { return cost == kLethal || (cost == kNoInfo && !allow_unknown); });
  }
  // TrajectoryPlanner::pointCost / lineCost (trajectory_planner.cpp:388-472): the planner's own ray walk for headingDiff;
  // unlike CostmapModel::pointCost it fails on INSCRIBED cells too
  __device__ double planLineCost(int x0, int x1, int y0, int y1) const {
    return lineMax(x0, y0, x1, y1, [&](uint8_t cost) { return cost == kLethal || cost == kInscribed || (cost == kNoInfo && !allow_unknown); });
  }
  // TrajectoryPlanner::headingDiff (:372-386): the farthest plan pose with a clear line of sight from the robot's cell
  __device__ double headingDiff(int cell_x, int cell_y, double x, double y, double heading, const double* plan, uint32_t n_plan) const {
    for (int i = (int)n_plan - 1; i >= 0; --i) {
      uint32_t gx_c, gy_c;
      if (worldToMap(g, plan[2 * i], plan[2 * i + 1], gx_c, gy_c)) {
        if (planLineCost(cell_x, (int)gx_c, cell_y, (int)gy_c) >= 0) {
          const double gx = g.ox + (gx_c + 0.5) * g.res, gy = g.oy + (gy_c + 0.5) * g.res;  // mapToWorld
          // angles::shortest_angular_distance(heading, atan2(..)) = normalize_angle(to - from), fmod form
          const double a = atan2(gy - y, gx - x) - heading;
          double r = fmod(fmod(a, 2.0 * M_PI) + 2.0 * M_PI, 2.0 * M_PI);
          if (r > M_PI) r -= 2.0 * M_PI;
          return fabs(r);
        }
      }
    }
    return 1.7976931348623157e308;  // DBL_MAX
  }
  // WorldModel::footprintCost(x, y, theta, spec) (world_model.h:65-86) + CostmapModel::footprintCost
  __device__ double footprintCost(double x, double y, double theta, const double* spec, uint32_t nfp) const {
    const double cos_th = cos(theta), sin_th = sin(theta);
    uint32_t cell_x, cell_y;
    if (!worldToMap(g, x, y, cell_x, cell_y)) return -1.0;
    if (nfp < 3) {
      const uint8_t cost = master[cell_y * g.nx + cell_x];
      if (cost == kLethal || cost == kInscribed || (cost == kNoInfo && !allow_unknown)) return -1.0;
      return cost;
    }
    double footprint_cost = 0.0;
    uint32_t fx = 0, fy = 0, px = 0, py = 0;
    for (uint32_t v = 0; v <= nfp; ++v) {
      uint32_t vx, vy;
      if (v < nfp) {
        const double sx = spec[2 * v], sy = spec[2 * v + 1];
        const double wx = x + (sx * cos_th - sy * sin_th), wy = y + (sx * sin_th + sy * cos_th);
        if (!worldToMap(g, wx, wy, vx, vy)) return -1.0;
        if (v == 0) {
          fx = vx;
          fy = vy;
          px = vx;
          py = vy;
          continue;
        }
      } else {  // closing edge: last -> first
        vx = fx;
        vy = fy;
      }
      const double line_cost = lineCost((int)px, (int)vx, (int)py, (int)vy);
      footprint_cost = fmax(line_cost, footprint_cost);
      if (line_cost < 0) return -1.0;
      px = vx;
      py = vy;
    }
    return footprint_cost;
  }
};

__device__ __forceinline__ double tpNewVelocity(double vg, double vi, double a_max, double dt) {  // trajectory_planner.h:369-374
  if ((vg - vi) >= 0) return fmin(vg, vi + a_max * dt);
  return fmax(vg, vi - a_max * dt);
}

__global__ __launch_bounds__(128) void k_tp_rollout(PlannerDev pl, TpDev tp, uint32_t first, int store_points) {
  const uint32_t inst = first + blockIdx.y;
  const navgpu_tp_config& c = tp.cfg;
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (store_points) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    s = tp.winner[inst];
    if (s < 0) return;
  } else if (s >= (int)tp.n_samples[inst]) {
    return;
  }
  TpWorld wm;
  wm.master = pl.master + (size_t)inst * pl.cells_padded;
  wm.g = Geom{pl.origin[2 * inst], pl.origin[2 * inst + 1], pl.res, pl.nx, pl.ny};
  wm.allow_unknown = c.allow_unknown != 0;
  const uint32_t* dpath = pl.path + (size_t)inst * pl.cells;
  const uint32_t* dgoal = pl.goal + (size_t)inst * pl.cells;
  const double* spec = pl.fp_spec + (size_t)inst * kMaxFootprint * 2;
  const uint32_t nfp = pl.fp_n[inst];
  const double* st = tp.start + (size_t)inst * 6;
  const double* smp = tp.samples + ((size_t)inst * tp.max_samples + s) * 3;
  const double vx_samp = smp[0], vy_samp = smp[1], vtheta_samp = smp[2];
  double* pts = tp.points + (size_t)inst * pl.max_sim_steps * 3;

  double x_i = st[0], y_i = st[1], theta_i = st[2];
  double vx_i = st[3], vy_i = st[4], vtheta_i = st[5];
  const double vmag = hypot(vx_samp, vy_samp);
  int num_steps;
  if (!c.heading_scoring)
    num_steps = int(fmax((vmag * c.sim_time) / c.sim_granularity, fabs(vtheta_samp) / c.angular_sim_granularity) + 0.5);
  else
    num_steps = int(c.sim_time / c.sim_granularity + 0.5);
  if (num_steps == 0) num_steps = 1;
  const double dt = c.sim_time / num_steps;
  double time = 0.0, heading_diff = 0.0;
  const double* plan = pl.plan + (size_t)inst * pl.max_plan * 2;
  const uint32_t n_plan = pl.plan_count[inst];
  const double impossible_cost = (double)pl.cells;  // path_map_.obstacleCosts()
  double path_dist = 0.0, goal_dist = 0.0, occ_cost = 0.0;
  double cost = -1.0;
  double ex = 0, ey = 0, eth = 0;
  int n_points = 0;
  bool finished = true;
  for (int i = 0; i < num_steps; ++i) {
    uint32_t cell_x, cell_y;
    if (!worldToMap(wm.g, x_i, y_i, cell_x, cell_y)) {
      cost = -1.0;
      finished = false;
      break;
    }
    const double footprint_cost = wm.footprintCost(x_i, y_i, theta_i, spec, nfp);
    if (footprint_cost < 0) {
      cost = -1.0;
      finished = false;
      break;
    }
    occ_cost = fmax(fmax(occ_cost, footprint_cost), double(wm.master[cell_y * pl.nx + cell_x]));
    if (c.simple_attractor) {  // :310-315 (the host refuses an empty plan)
      const double gx = plan[2 * (n_plan - 1)], gy = plan[2 * (n_plan - 1) + 1];
      goal_dist = (x_i - gx) * (x_i - gx) + (y_i - gy) * (y_i - gy);
    } else {
      bool update_path_and_goal_distances = true;
      if (c.heading_scoring) {  // :321-327
        if (time >= c.heading_scoring_timestep && time < c.heading_scoring_timestep + dt)
          heading_diff = wm.headingDiff((int)cell_x, (int)cell_y, x_i, y_i, theta_i, plan, n_plan);
        else
          update_path_and_goal_distances = false;
      }
      if (update_path_and_goal_distances) {
        path_dist = (double)dpath[cell_y * pl.nx + cell_x];
        goal_dist = (double)dgoal[cell_y * pl.nx + cell_x];
        if (impossible_cost <= goal_dist || impossible_cost <= path_dist) {
          cost = -2.0;
          finished = false;
          break;
        }
      }
    }
    ex = x_i;
    ey = y_i;
    eth = theta_i;
    if (store_points && n_points < (int)pl.max_sim_steps) {
      pts[3 * n_points] = x_i;
      pts[3 * n_points + 1] = y_i;
      pts[3 * n_points + 2] = theta_i;
    }
    ++n_points;
    vx_i = tpNewVelocity(vx_samp, vx_i, c.acc_lim_x, dt);
    vy_i = tpNewVelocity(vy_samp, vy_i, c.acc_lim_y, dt);
    vtheta_i = tpNewVelocity(vtheta_samp, vtheta_i, c.acc_lim_theta, dt);
    const double nx_ = x_i + (vx_i * cos(theta_i) + vy_i * cos(M_PI_2 + theta_i)) * dt;  // computeNewXPosition
    const double ny_ = y_i + (vx_i * sin(theta_i) + vy_i * sin(M_PI_2 + theta_i)) * dt;
    theta_i = theta_i + vtheta_i * dt;
    x_i = nx_;
    y_i = ny_;
    time += dt;
  }
  if (finished)
    cost = !c.heading_scoring ? c.pdist_scale * path_dist + goal_dist * c.gdist_scale + c.occdist_scale * occ_cost
                              : c.occdist_scale * occ_cost + c.pdist_scale * path_dist + 0.3 * heading_diff + goal_dist * c.gdist_scale;
  if (store_points) return;
  TpOut o;
  o.cost = cost;
  o.ex = ex;
  o.ey = ey;
  o.eth = eth;
  o.n_points = n_points;
  o.ahead = 0.0;
  o.ahead_ok = 0;
  if (n_points > 0) {  // createTrajectories :680-688: goal_map_ at the heading_lookahead point of the endpoint
    const double x_r = ex + c.heading_lookahead * cos(eth), y_r = ey + c.heading_lookahead * sin(eth);
    uint32_t cell_x, cell_y;
    if (worldToMap(wm.g, x_r, y_r, cell_x, cell_y)) {
      o.ahead = (double)dgoal[cell_y * pl.nx + cell_x];
      o.ahead_ok = 1;
    }
  }
  tp.out[(size_t)inst * tp.max_samples + s] = o;
}

void launch_tp_rollout(const PlannerDev& pl, const TpDev& tp, uint32_t first, uint32_t count, int store_points, hipStream_t s) {
  const uint32_t blocks = store_points ? 1u : (tp.max_samples + 127u) / 128u;
  hipLaunchKernelGGL(k_tp_rollout, dim3(blocks, count), dim3(128), 0, s, pl, tp, first, store_points);
}

}  // namespace navgpu
