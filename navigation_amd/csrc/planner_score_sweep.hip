// k_score_sweep (gfx950): the product form of the scoring launch - use_dwa with a fixed step count (discretize_by_time)
// and DWAPlanner's own MapGrid options - as a SCREENED SWEEP plus QUEUED FOOTPRINT WALKS inside one workgroup.
//   SimpleTrajectoryGenerator::generateTrajectory / computeNewPositions (simple_trajectory_generator.cpp:180-276)
//   SimpleScoredSamplingPlanner::scoreTrajectory (simple_scored_sampling_planner.cpp:50-79), critic order of
//     dwa_planner.cpp:167-173: oscillation, obstacle, goal_front, alignment, path, goal
//   ObstacleCostFunction::scoreTrajectory / footprintCost (obstacle_cost_function.cpp:74-142),
//   WorldModel::footprintCost (world_model.h:65-86), CostmapModel::footprintCost / lineCost / pointCost
//     (costmap_model.cpp:50-142), LineIterator (line_iterator.h:38-139)
//   MapGridCostFunction::scoreTrajectory (map_grid_cost_function.cpp:75-129, aggregation Last)
//   OscillationCostFunction::scoreTrajectory (oscillation_cost_function.cpp:166-176)
//
// Why two phases.  One lane rolls one sample out.  89 % of the trajectory points pass the per-cell screens of the robot's
// LDS image (planner_score.hip builds it: nothing in reach of the footprint can fail, the path / goal grids hold a distance
// here) in ~60 vector instructions; the others have to walk the footprint's outline through the costmap window, ~1 000
// instructions with a few dozen dependent LDS round trips.  Walked where they come up (rounds 1-3: k_score_tab), a wave pays
// for a walk whenever one of its lanes needs one - 11 % of the wave-steps, at 57 % lane density - and its 20 steps are serial.
// Here the sweep only DECIDES: a point that needs its footprint walked is appended to a queue in LDS (pose, owner lane, step)
// and the lane moves on.  Every kSweepBlockSteps steps the workgroup meets, all 256 lanes take queue entries - whoever
// pushed them - and walk them side by side (every lane busy, no wave waiting for another's obstacle), and report to the
// owner: a failed point (obstacle_cost_function.cpp:127-131 -> -6) ends the owner's rollout at the next block, the last
// point's cost (or every point's, with sum_scores) is added up in the owner's word.  The critics' precedence is untouched:
// scoreTrajectory returns the code of the FIRST critic in the list that fails anywhere on the trajectory, the obstacle
// critic is the first of those a walk can fail, and the other critics' failures are kept by order as before.
// A full queue stalls the lanes that could not push: they take the same step again in the next block.
#include "planner_score.h"

namespace navgpu {

constexpr int kSweepThreads = NAVGPU_SCORE_TAB_THREADS;
#ifndef NAVGPU_SWEEP_BLOCK_STEPS
#define NAVGPU_SWEEP_BLOCK_STEPS 4  // A/B over 3..5 x 512..1024 on configs[2] and configs[4]: tools/ab_variants.sh, tools/ab_configs4.sh
#endif
#ifndef NAVGPU_SWEEP_QUEUE
#define NAVGPU_SWEEP_QUEUE 768
#endif
constexpr int kSweepBlockSteps = NAVGPU_SWEEP_BLOCK_STEPS;  // trajectory points a lane sweeps between two walk phases
constexpr int kSweepQueue = NAVGPU_SWEEP_QUEUE;             // walk entries a workgroup holds (12 B each)
constexpr uint32_t kWalkFailed = 0x80000000u;               // s_obs[owner]: a walked point failed; low bits: summed costs
#ifdef NAVGPU_SWEEP_COUNTS  // experiment builds only (tools/probe_sweep_counts.py): how many points look closer, are walked, fail
__device__ unsigned long long g_sweep_counts[16];
#define SW_COUNT(slot_, n_)                                                                    \
  do {                                                                                           \
    const unsigned long long swc_ = (unsigned long long)(n_);                                    \
    if ((threadIdx.x & 63u) == 0) atomicAdd(&g_sweep_counts[slot_], swc_);                       \
  } while (0)
#else
#define SW_COUNT(slot_, n_)
#endif
#ifdef NAVGPU_SWEEP_TIMING  // experiment builds only (tools/probe_sweep_timing.py): where a wave's time goes, one workgroup in 16 reports
__device__ unsigned long long g_sweep_stats[16];
#define SW_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define SW_ACC(i, x) sw_t[i] += (x)
#else
#define SW_STAMP(v)
#define SW_ACC(i, x)
#endif

// TRIGF: navgpu_dwa_config::rollout_trig = 1 - computeNewPositions' cos(pos[2]) / sin(pos[2]) name the float functions and
// vel[0] * cos(pos[2]) is a float product (its value: the table's double, rounded to float)
template <int CHUNK, bool TRIGF>
__global__ __launch_bounds__(kSweepThreads, NAVGPU_SCORE_TAB_WAVES) void k_score_sweep(PlannerDev pl, uint32_t first) {
  constexpr int THREADS = kSweepThreads;
  extern __shared__ __align__(16) uint8_t s_dyn[];
  uint8_t* s_win = s_dyn;
  __shared__ double s_rc[THREADS / 64];
  __shared__ int s_ri[THREADS / 64];
  __shared__ int s_cnt[2];
  // the walk queue: pose (x, y as floats), tag = owner lane | step << 10 | table row << 17 | cost wanted << 31
  __shared__ float s_qx[kSweepQueue], s_qy[kSweepQueue];
  __shared__ uint32_t s_qt[kSweepQueue];
  __shared__ uint32_t s_obs[THREADS];
  __shared__ uint32_t s_qn[2];
  __shared__ uint32_t s_more[3];  // rotating by block: any lane of the workgroup still rolling out?

  const uint32_t inst = first + blockIdx.z;  // grid: x = workgroup within its row group, y = row group, z = robot (a robot's workgroups are dispatched together)
  const uint32_t tid = threadIdx.x;
#ifdef NAVGPU_SWEEP_TIMING
  unsigned long long sw_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  SW_STAMP(sw0);
  __builtin_amdgcn_s_setprio(3);  // a new workgroup's waves are the youngest on their SIMDs: get the image in before yielding
  const navgpu_dwa_config& c = pl.cfg;
  const Geom g = geomOf(pl, inst);
  const navgpu_robot_state st = pl.state[inst];
  const uint8_t* master = pl.master + (size_t)inst * pl.cells_padded;
  const uint32_t* dpath = pl.path + (size_t)inst * pl.cells;
  const uint32_t* dgoal = pl.goal + (size_t)inst * pl.cells;
  const uint32_t* dfront = pl.goal_front + (size_t)inst * pl.cells;
  const int32_t* cnt = pl.axis_count + 4 * inst;
  const int n_samples = cnt[3];
  const uint32_t nfp = pl.fp_n[inst];
  const int win = (int)pl.win;

  // ---- stage (one batch of loads, one barrier): the robot's image; the lane's own sample and reject byte come straight from HBM
  if (tid == 0) s_cnt[0] = s_cnt[1] = 0;
  if (tid < 2) s_qn[tid] = 0;
  if (tid < 3) s_more[tid] = 0;
  s_obs[tid] = 0;
  // the robot's scalars, worked out once by k_score_prep_tab (planner_score.hip): window origin (robot cell - win / 2), the
  // forward point's margin band, whether this launch can screen at all; and the loads the lanes need later, issued now
  const uint8_t* rej_bytes = pl.prep + (size_t)inst * pl.prep_stride + score_prep_reject_offset(pl);
  const int32_t* aux = reinterpret_cast<const int32_t*>(rej_bytes - kScoreAuxBytes);
  const int wx0 = aux[0], wy0 = aux[1];
  const uint32_t fwd_lo = (uint32_t)aux[2], fwd_nx = (uint32_t)aux[3], fwd_ny = (uint32_t)aux[4];
  const bool need_margin = aux[5] != 0, fwd_screen = aux[6] != 0;
  const int free_cells = aux[7];  // Euclidean distance (rounded down) from the robot's cell to the nearest cell with a screen set that still counts after step 0 (or the window's edge)
  const int start_fail = aux[8];  // 4 / 5: the path / goal critic fails at the robot's own cell, i.e. for every sample at step 0; 0: neither
  const uint32_t osc = pl.osc_flags[inst];
  const int32_t align_on = pl.align_on[inst];
  const int win_bytes = (win * win + 15) & ~15;
  const int nw = (win + 31) >> 5;
  const bool walk_swap = pl.cfg.allow_unknown != 0;  // the window bytes are kept in walk order (planner_score.hip)
  const uint32_t walk_fail = walk_swap ? 255u : 254u;
  const int K = (int)pl.tab_steps;
  const int tnfp = (int)pl.tab_nfp;
  const int lrows = (int)pl.tab_rows;
  double* s_trig = reinterpret_cast<double*>(s_dyn + win_bytes + score_bits_bytes(win));  // [rows][K][4] cs, sn, cs2, sn2
  double* s_rot = s_trig + (size_t)lrows * K * 4;                                          // [rows][K][tnfp][2]
  // Lane mapping.  Lanes are v_theta-major so that a wave shares one heading sequence.  The v_theta rows are cut into groups
  // of tab_rows (what the LDS budget holds); a group takes bpg consecutive workgroups, which enumerate its rows x (vx, vy)
  // pairs.  Blocks past the last group idle.
  const int nxy = max(cnt[0] * cnt[1], 1);
  const uint32_t part_slot = blockIdx.y * gridDim.x + blockIdx.x;
  const int t_row_base = (int)blockIdx.y * (int)pl.tab_rows;
  const int t_rows = min(max(cnt[2] - t_row_base, 0), (int)pl.tab_rows);
  const int t_li0 = (int)blockIdx.x * THREADS;
  if (t_li0 >= t_rows * nxy) {  // no sample for this workgroup (the grid is sized for the largest (vx, vy) grid the configuration allows)
    if (tid == 0) {
      pl.part_cost[(size_t)inst * pl.score_blocks + part_slot] = 1.0e300;
      pl.part_index[(size_t)inst * pl.score_blocks + part_slot] = 0x7FFFFFFF;
    }
    return;
  }
  // ---- lane -> sample slot (x-outer, y, theta-inner, as the reference enumerates: what results are keyed by), and its loads
  const int li = t_li0 + (int)tid;
  const int t_row = divSmall(li, nxy);  // row within the group = row of the tables in LDS
  const int t_r = li - t_row * nxy;     // index of the (vx, vy) pair, x-outer
  const int t_ith = t_row_base + t_row;
  const bool in_range = n_samples > 0 && t_row < t_rows;
  const int sidx = t_r * cnt[2] + t_ith;
  const int nyv = max(cnt[1], 1);
  const int s_ix = divSmall(t_r, nyv), s_iy = t_r - s_ix * nyv;
  const float* axis = pl.axis_samples + (size_t)inst * 3 * pl.max_axis;
  const uint32_t amax = pl.max_axis - 1u;
  const float vs0 = axis[min((uint32_t)s_ix, amax)], vs1 = axis[pl.max_axis + min((uint32_t)s_iy, amax)], vs2 = axis[2u * pl.max_axis + min((uint32_t)t_ith, amax)];
  const uint32_t rej_xy = rej_bytes[in_range ? t_r : 0];  // the (vx, vy) pair's half of the reject tests (k_score_prep_tab)
  {  // window + screens, and of the tables only the v_theta rows this workgroup's samples use, at their usual place
    const uint4* img = reinterpret_cast<const uint4*>(pl.prep + (size_t)inst * pl.prep_stride);
    uint4* lds = reinterpret_cast<uint4*>(s_dyn);
    const int last = max(t_rows, 1) - 1;
    const int r0 = min(divSmall(t_li0, nxy), last);
    const int r1 = min(divSmall(t_li0 + THREADS - 1, nxy), last);
    const int n16w = (int)((win_bytes + score_bits_bytes(win)) >> 4);
    const int ncopy = r1 - r0 + 1;
    const int n_trig = ncopy * K * 2, n_rot = ncopy * K * tnfp;  // 32 B per entry, 16 B per vertex
    const int l_trig = n16w + r0 * K * 2, g_trig = n16w + (t_row_base + r0) * K * 2;
    const int l_rot = n16w + lrows * K * 2 + r0 * K * tnfp, g_rot = n16w + (int)pl.tab_nth * K * 2 + (t_row_base + r0) * K * tnfp;
    // three plain block copies (window + screens; the trig rows; the rotated-footprint rows), their first 2 + 1 + 1 loads per lane
    // issued together (a configs[2] image whole: 375 + 80 + 160 sixteen-byte words); what is larger follows in loops
    const int ia0 = (int)tid, ia1 = (int)tid + THREADS;
    const uint4 va0 = img[min(ia0, n16w - 1)], va1 = img[min(ia1, n16w - 1)];
    const uint4 vb = img[g_trig + min((int)tid, n_trig - 1)], vc = img[g_rot + min((int)tid, max(n_rot, 1) - 1)];
    if (ia0 < n16w) lds[ia0] = va0;
    if (ia1 < n16w) lds[ia1] = va1;
    if ((int)tid < n_trig) lds[l_trig + (int)tid] = vb;
    if ((int)tid < n_rot) lds[l_rot + (int)tid] = vc;
    if (n16w > 2 * THREADS || n_trig > THREADS || n_rot > THREADS) {  // (one wave-uniform test instead of three loop headers)
      for (int i = (int)tid + 2 * THREADS; i < n16w; i += THREADS) lds[i] = img[i];
      for (int i = (int)tid + THREADS; i < n_trig; i += THREADS) lds[l_trig + i] = img[g_trig + i];
      for (int i = (int)tid + THREADS; i < n_rot; i += THREADS) lds[l_rot + i] = img[g_rot + i];
    }
  }
  SW_STAMP(sw1a);
  __syncthreads();
  __builtin_amdgcn_s_setprio(0);
  SW_STAMP(sw1);
  SW_ACC(0, sw1a - sw0);
  SW_ACC(1, sw1 - sw1a);

  // NOTE: the LDS read is unconditional (clamped index) and the global fallback sits in its own rarely-taken branch; a
  // `cond ? lds[i] : global[j]` form makes hipcc merge both into one FLAT load.
  auto inWin = [&](int x, int y) { return (unsigned)(x - wx0) < (unsigned)win && (unsigned)(y - wy0) < (unsigned)win; };
  auto cellCost = [&](int x, int y) -> uint8_t {
    const bool in = inWin(x, y);
    uint32_t v = s_win[in ? (y - wy0) * win + (x - wx0) : 0];
    asm volatile("" : "+v"(v));  // pin the ds_read here so it cannot be re-merged with the global load below
    if (walk_swap && v >= 254u) v ^= 1u;  // back from walk order
    if (__builtin_expect(!in, 0)) v = master[y * g.nx + x];
    return (uint8_t)v;
  };
  const double inv_res = pl.inv_res;
  // Costmap2D::worldToMap (costmap_2d.cpp:208-220) with the two fp64 divisions replaced by a multiply; exact: whenever the
  // product is not clear of an integer by 1e-7 (error bound 5e-10 below 1e6 cells) the division is redone.
  auto w2m = [&](double wx, double wy, uint32_t& mx, uint32_t& my) -> bool {
    const double dx = wx - g.ox, dy = wy - g.oy;
    const double qx = dx * inv_res, qy = dy * inv_res;
    double fx = floor(qx), fy = floor(qy);
    const double rx = qx - fx, ry = qy - fy;
    if (__builtin_expect(fmin(rx, ry) < 1.0e-7 || fmax(rx, ry) > 1.0 - 1.0e-7, 0)) {
      fx = !(dx >= 0.0) ? -1.0 : (qx >= 1.0e6 ? 1.0e6 : (double)(int)(dx / g.res));  // wx < origin -> false (costmap_2d.cpp:210)
      fy = !(dy >= 0.0) ? -1.0 : (qy >= 1.0e6 ? 1.0e6 : (double)(int)(dy / g.res));
    }
    // v_cvt_i32_f64 saturates (a point left of / below the origin floors to a negative cell, one far beyond the grid to
    // INT_MAX: both fail the size test as unsigned numbers), which a C++ cast does not promise
    int ix, iy;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(ix) : "v"(fx));
    asm("v_cvt_i32_f64 %0, %1" : "=v"(iy) : "v"(fy));
    mx = (uint32_t)ix;
    my = (uint32_t)iy;
    return mx < g.nx && my < g.ny;
  };

  // ---- CostmapModel::footprintCost at pose (x, y) with the rotated vertices of table entry te (costmap_model.cpp:50-142):
  // returns true when some vertex is off the map (footprint_cost = -1); else mx = the largest byte on the outline, in walk
  // order (>= walk_fail: some cell fails pointCost).  STRAIGHT-LINE: the vertex / edge loop runs nfp times for every lane
  // (a scalar counter), an edge is one chunk of Bresenham cells (LineIterator, line_iterator.h:38-139) whose addresses do
  // not depend on the bytes read - all ds_reads of a chunk in flight, cells past the end re-read the first cell - and nothing
  // leaves early: a lethal cell or an off-map vertex cannot be undone by what is read after it.  (As nested loops with
  // breaks the same walk compiled to ~950 vector and as many scalar instructions, most of them exec-mask bookkeeping.)
  // outside: an edge's end points are not both inside the LDS window (never, with a correctly sized window): the caller
  // repeats the walk on the costmap itself (walkCostmap).
  auto walkWindow = [&](const double x, const double y, const int te, uint32_t& mx, bool& outside) -> bool {
    const double* rot = s_rot + (size_t)te * tnfp * 2;
    bool off = false;
    outside = false;
    mx = 0;
    uint32_t ux, uy;
    off |= !w2m(x + rot[0], y + rot[1], ux, uy);  // world_model.h:72-73 (the rotation is the tables')
    const int fx0 = (int)ux, fy0 = (int)uy;
    int pxc = fx0, pyc = fy0;
    for (uint32_t v = 1; v <= nfp; ++v) {  // edges v-1 -> v, and last -> first
      int vx = fx0, vy = fy0;
      if (v < nfp) {  // (wave-uniform)
        off |= !w2m(x + rot[2 * v], y + rot[2 * v + 1], ux, uy);
        vx = (int)ux;
        vy = (int)uy;
      }
      const int dx = vx - pxc, dy = vy - pyc;
      const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
      const bool xmaj = adx >= ady;
      const int sx = (vx >= pxc) ? 1 : -1, sy = (vy >= pyc) ? win : -win;
      const int den = max(adx, ady), numadd = min(adx, ady), numpixels = den;
      int num = den >> 1;
      const bool in = inWin(pxc, pyc) && inWin(vx, vy);  // every cell of the line lies in the end points' bounding box
      outside |= !in;
      const int inc1 = in ? (xmaj ? sy : sx) : 0, inc2 = in ? (xmaj ? sx : sy) : 0;
      int addr = in ? (pyc - wy0) * win + (pxc - wx0) : 0;
      const int addr_first = addr;
      int cp = 0;
      do {
        uint32_t cellv[CHUNK];
#pragma unroll
        for (int u = 0; u < CHUNK; ++u) {
          cellv[u] = s_win[(cp + u <= numpixels) ? addr : addr_first];
          num += numadd;
          const bool ge = num >= den;
          num -= ge ? den : 0;
          addr += (ge ? inc1 : 0) + inc2;
        }
#pragma unroll
        for (int u = 0; u < CHUNK; ++u) mx = max(mx, cellv[u]);
        cp += CHUNK;
      } while (__ballot(cp <= numpixels) != 0ull);  // (a second chunk only for edges longer than the launch's CHUNK)
      pxc = vx;
      pyc = vy;
    }
    return off;
  };
  // the same walk on the costmap in HBM, cell by cell, for an outline that leaves the LDS window (vertices known on the map)
  auto walkCostmap = [&](const double x, const double y, const int te) -> uint32_t {
    const double* rot = s_rot + (size_t)te * tnfp * 2;
    uint32_t mx = 0, ux, uy;
    w2m(x + rot[0], y + rot[1], ux, uy);
    const int fx0 = (int)ux, fy0 = (int)uy;
    int pxc = fx0, pyc = fy0;
    for (uint32_t v = 1; v <= nfp; ++v) {
      int vx = fx0, vy = fy0;
      if (v < nfp) {
        w2m(x + rot[2 * v], y + rot[2 * v + 1], ux, uy);
        vx = (int)ux;
        vy = (int)uy;
      }
      int deltax = vx - pxc, deltay = vy - pyc;
      deltax = deltax < 0 ? -deltax : deltax;
      deltay = deltay < 0 ? -deltay : deltay;
      int lx = pxc, ly = pyc;
      int xinc1, xinc2, yinc1, yinc2, den, num, numadd, numpixels;
      xinc1 = xinc2 = (vx >= pxc) ? 1 : -1;
      yinc1 = yinc2 = (vy >= pyc) ? 1 : -1;
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
      for (int cp = 0; cp <= numpixels; ++cp) {
        const uint32_t cc = master[ly * g.nx + lx];
        const uint32_t ct = (walk_swap && cc >= 254u) ? (cc ^ 1u) : cc;  // walk order, like the LDS bytes
        mx = ct > mx ? ct : mx;
        num += numadd;
        if (num >= den) {
          num -= den;
          lx += xinc1;
          ly += yinc1;
        }
        lx += xinc2;
        ly += yinc2;
      }
      pxc = vx;
      pyc = vy;
    }
    return mx;
  };

  double total = -1.0;
  int status = NAVGPU_SAMPLE_REJECTED;
  const float vs[3] = {vs0, vs1, vs2};
  // generateTrajectory: reject tests (:193-200); the step count is the tables' (ceil(sim_time / sim_granularity), host)
  const bool reject = !in_range || ((rej_xy & 1u) && (c.min_rot_vel >= 0 && fabs((double)vs2) + 1e-4 < c.min_rot_vel)) || (rej_xy & 2u) || K <= 0;  // `return num_steps > 0` (:250)
  if (in_range && !reject) status = NAVGPU_SAMPLE_SCORED;
  const double dt = pl.tab_dt;
  const double xv = vs[0], yv = vs[1], thv = vs[2];  // traj.xv_, yv_, thetav_
  const bool osc_fail = ((osc & NAVGPU_OSC_FORWARD_POS_ONLY) && xv < 0.0) || ((osc & NAVGPU_OSC_FORWARD_NEG_ONLY) && xv > 0.0) ||
                        ((osc & NAVGPU_OSC_STRAFE_POS_ONLY) && yv < 0.0) || ((osc & NAVGPU_OSC_STRAFE_NEG_ONLY) && yv > 0.0) ||
                        ((osc & NAVGPU_OSC_ROT_POS_ONLY) && thv < 0.0) || ((osc & NAVGPU_OSC_ROT_NEG_ONLY) && thv > 0.0);
  const double sc_obs = pl.scale_obstacle, sc_gf = pl.scale_goal, sc_al = align_on ? pl.scale_path : 0.0, sc_path = pl.scale_path,
               sc_goal = pl.scale_goal;
  const bool en_obs = sc_obs != 0, en_gf = sc_gf != 0, en_al = sc_al != 0, en_path = sc_path != 0, en_goal = sc_goal != 0;
  // first_fail: order index of the earliest critic in the list that has failed (1 obstacle .. 5 goal), 6 = none; fail_code its
  // code - the only one scoreTrajectory's in-order sum can return (simple_scored_sampling_planner.cpp:59-66)
  int first_fail = 6;
  int fail_code = 0;
  uint32_t d_gf = 0, d_al = 0, d_path = 0, d_goal = 0;  // the map-grid critics' values (aggregation Last: the final point's)
  // The distance grids are read ONCE per lane, after the rollout, all loads in flight together (a look-up where it comes up is a
  // dependent L2 round trip in the middle of a wave's sweep: the points that looked closer took 1.7 us each).  Until then:
  //   fail_cell  cell where the path / goal critic failed by its screen bit (the bit IS the test; the grid says which code)
  //   last_c, last_f  centre and forward cell of the final point; last_f = ~0: that point was never reached
  uint32_t fail_cell = 0xFFFFFFFFu, last_c = 0, last_f = 0xFFFFFFFFu;
  const double fpd = c.forward_point_distance;
  const uint32_t N_obst = pl.cells, N_unreach = pl.cells + 1;
  if (en_obs && nfp == 0) {  // "Footprint spec is empty" (obstacle_cost_function.cpp:78-82)
    fail_code = -9;
    first_fail = 1;
  }
  // a critic is live while no critic before it in the order has failed; the lowest enabled order decides when nothing is
  // left to evaluate
  const int min_order = en_obs ? 1 : en_gf ? 2 : en_al ? 3 : en_path ? 4 : en_goal ? 5 : 6;
  uint32_t fb_off = (uint32_t)win_bytes;  // (in a vector register: as a scalar it is spilled and read back with v_readlane at every point)
  asm volatile("" : "+v"(fb_off));
  const uint4* s_fb4 = reinterpret_cast<const uint4*>(s_dyn + fb_off);
  const bool scr_sum = c.sum_scores != 0;  // the obstacle screen: dilated "not free" with sum_scores, else dilated "can fail"
  const bool screen_on = fwd_screen && (nfp >= 3 || !en_obs);
  float px = st.pos[0], py = st.pos[1];
  int step = 0;  // the lane's next trajectory point; stops at K or where the rollout ended
  // all-ones while the lane's rollout goes on (kept as a mask in a vector register: as a bool the compiler branches on it)
  uint32_t alive_m = (in_range && !reject && !osc_fail && first_fail > min_order) ? 0xFFFFFFFFu : 0u;
#ifdef NAVGPU_SWEEP_X_NOSTEPS  // timing experiment: prologue + epilogue only
  alive_m = 0;
#endif
  const bool margin_on = __builtin_amdgcn_readfirstlane((int)need_margin) != 0;
  const uint32_t lane = tid & 63u;
  const int te0 = t_row * K;
  const double vxd = vs[0], vyd = vs[1];
  // the path / goal screens count while their critics are live (a critic that has failed, or that follows one that has, cannot
  // change the outcome any more); all-ones where the launch has no screen at all
  uint32_t scr_z = 0xFFFFFFFFu, scr_w = 0xFFFFFFFFu;
  const uint32_t scr_off = screen_on ? 0u : 0xFFFFFFFFu;
  SW_STAMP(sw2);
  SW_ACC(2, sw2 - sw1);
  // k_free: the steps (0 .. k_free - 1) at which NO lane of the wave can be `free_cells` cells (Euclidean) from the robot's cell yet.  A pose
  // moves |v| dt per step (the rotated velocity's length; each float rounding adds < 1e-6 m), a cell index trails its pose by < 1 cell
  // in x and in y, as the start cell does: point k's cell is fewer than k s + sqrt(2) cells from the start cell, s = |v| dt / res rounded
  // up - so k < (free_cells - 2) / s is safe.  Only where the launch screens at all, the window lies clear of the map's margin band, and never the last point.
  int k_free = 0;
  {
    const float spd = sqrtf(vs0 * vs0 + vs1 * vs1) * (float)(dt * inv_res) * 1.0001f + 1.0e-4f;
    int kl = K - 1;  // (a lane that does not roll out does not hold the wave back)
    if (alive_m != 0u) kl = (screen_on && !margin_on && free_cells >= 3) ? (int)fminf((float)(free_cells - 2) / spd, (float)(K - 1)) : 0;
    for (int off = 32; off > 0; off >>= 1) kl = min(kl, __shfl_xor(kl, off));
    k_free = __builtin_amdgcn_readfirstlane(kl);
  }
  // ---- the free run: the first k_free points of the whole wave - every screen is clear there, the point is in the window and not the
  // last: nothing to look at, the pose just moves on (computeNewPositions :253-260; dead lanes compute along)
  for (int k = 0; k < k_free; ++k) {
    const int te = te0 + k;
    const double cs = s_trig[4 * te], sn = s_trig[4 * te + 1], cs2 = s_trig[4 * te + 2], sn2 = s_trig[4 * te + 3];
    const double tx = TRIGF ? (double)(vs[0] * (float)cs) : vxd * cs, ty = TRIGF ? (double)(vs[0] * (float)sn) : vxd * sn;
    const float nxp = (float)(px + (tx + vyd * cs2) * dt);
    const float nyp = (float)(py + (ty + vyd * sn2) * dt);
    px = nxp;
    py = nyp;
  }
  if (alive_m != 0u) step = k_free;
  SW_COUNT(9, k_free);
  SW_COUNT(10, (unsigned)k_free * (unsigned)__popcll(__ballot(alive_m != 0u)));
  if (k_free > 0 && start_fail != 0 && alive_m != 0u) {  // what looking closer at point 0 would have found (the free run skipped it)
    first_fail = start_fail;
    fail_code = 0;
    fail_cell = (uint32_t)(wy0 + win / 2) * g.nx + (uint32_t)(wx0 + win / 2);
    scr_z = first_fail > 4 ? 0xFFFFFFFFu : 0u;
    scr_w = 0u;
    alive_m = first_fail > min_order ? 0xFFFFFFFFu : 0u;
  }
  for (uint32_t blk = 0;; ++blk) {
    SW_STAMP(sb0);
    // ---- sweep: up to kSweepBlockSteps points per lane.  The screened path is STRAIGHT-LINE for the whole wave - lanes whose
    // rollout is over compute along (their state is never read again) - and ends in one wave-uniform branch: does any live
    // lane have to look closer?  (Written with the usual per-lane conditions the compiler spends as many scalar instructions
    // on exec-mask bookkeeping as the point costs in vector ones, and the scalar unit is shared by a CU's four SIMDs.)
    if (__ballot(alive_m != 0u) != 0ull) {
      for (int it = 0; it < kSweepBlockSteps; ++it) {
        const int sc = min(step, K - 1);
        const int te = te0 + sc;
        const double x = px, y = py;
        const double cs = s_trig[4 * te], sn = s_trig[4 * te + 1];
        SW_COUNT(4, 1);
        SW_COUNT(8, __popcll(__ballot(alive_m != 0u)));
        // ---- advance (computeNewPositions :253-260): fp64 on fp32 state, rounded back to fp32 (the heading is the tables')
        const double cs2 = s_trig[4 * te + 2], sn2 = s_trig[4 * te + 3];
        const double tx = TRIGF ? (double)(vs[0] * (float)cs) : vxd * cs, ty = TRIGF ? (double)(vs[0] * (float)sn) : vxd * sn;
        const float nxp = (float)(px + (tx + vyd * cs2) * dt);
        const float nyp = (float)(py + (ty + vyd * sn2) * dt);
        uint32_t cx, cy;
        const bool ok_c = w2m(x, y, cx, cy);
        // ---- screen: on every point but the last a critic can only FAIL (its value is overwritten: aggregation Last; with
        // sum_scores the obstacle critic adds the point's cost, which is 0 when everything in reach is free).  One 16-byte
        // LDS read says whether any critic could fail here; if none can, the point is done.  The screen word is read whatever
        // the point is (clamped address) and the NEXT pose is computed while that read is in flight.
        const uint32_t lx = cx - (uint32_t)wx0, ly = cy - (uint32_t)wy0;
        bool in_w = max(lx, ly) < (uint32_t)win;
        uint32_t force = (sc == K - 1) ? 0xFFFFFFFFu : scr_off;  // no screen for the last point, or for this launch
        if (margin_on) {  // (wave-uniform, rare: the window reaches beyond the map or into the forward point's margin band)
          in_w = in_w && ok_c;
          if (!((cx - fwd_lo < fwd_nx) && (cy - fwd_lo < fwd_ny))) force = 0xFFFFFFFFu;
        }
        const uint32_t fb_i = in_w ? ly * (uint32_t)nw + (lx >> 5) : 0u;
        const uint4 fbw = s_fb4[fb_i];
        const uint32_t any = (scr_sum ? fbw.x : fbw.y) | (fbw.z & scr_z) | (fbw.w & scr_w) | force | (in_w ? 0u : 0xFFFFFFFFu);
        const bool unscr = ((any >> (lx & 31u)) & alive_m & 1u) != 0u;
        px = nxp;
        py = nyp;
        step -= (int)alive_m;  // (+1 while alive)
        if (__ballot(unscr) != 0ull) {  // (wave-uniform)
          SW_STAMP(ss0);
          SW_COUNT(3, 1);
          SW_COUNT(0, __popcll(__ballot(unscr)));
          // ---- looking closer, in the same style as the screened path: every lane computes along, WHO is concerned is a mask (u: the
          // lanes that look closer; all-ones / zero words combined with integer ops, picked with sel), the rare cases - a footprint of
          // fewer than three points, a point outside the LDS window, the forward point's worldToMap - sit behind one wave-uniform
          // ballot each.  (As nested per-lane ifs this block compiled to ~400 instructions of which a wave that entered it ran
          // most: a sixth of the kernel.)
          auto sel = [](uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); };
          const uint32_t u = unscr ? 0xFFFFFFFFu : 0u;
          const uint32_t m_last = sc == K - 1 ? 0xFFFFFFFFu : 0u;
          const uint32_t m_on = ok_c ? 0xFFFFFFFFu : 0u;  // the centre is on the map (cx < nx && cy < ny)
          const uint32_t m_inw = in_w ? 0xFFFFFFFFu : 0u;
          const uint32_t bit = lx & 31u;
          const uint32_t b_nf = 0u - ((fbw.x >> bit) & 1u), b_cf = 0u - ((fbw.y >> bit) & 1u), b_z = 0u - ((fbw.z >> bit) & 1u), b_w = 0u - ((fbw.w >> bit) & 1u);
          uint32_t m_stall = 0u;
          // ---- obstacle critic: decide here, walk later
          if (en_obs) {  // (uniform)
            const uint32_t m_off = u & ~m_on;  // CostmapModel::footprintCost: centre off the map -> -1 -> -6 (obstacle_cost_function.cpp:127-131)
            first_fail = (int)sel(m_off, 1u, (uint32_t)first_fail);
            fail_code = (int)sel(m_off, (uint32_t)-6, (uint32_t)fail_code);
            if (nfp < 3) {  // (uniform, rare) the centre cell alone (costmap_model.cpp:63-72)
              if ((u & m_on) != 0u) {
                const uint8_t cc = cellCost(cx, cy);
                if (cc == kLethal || cc == kInscribed || (cc == kNoInfo && c.allow_unknown == 0)) {
                  fail_code = -6;
                  first_fail = 1;
                } else if (scr_sum || m_last != 0u) {
                  atomicAdd(&s_obs[tid], (uint32_t)cc);  // occ = max(f_cost, centre cell) = the cell's cost
                }
              }
            } else {
              // all_free: every cell the footprint can touch is FREE_SPACE -> the point costs exactly 0.  Without sum_scores only the
              // LAST point's footprint cost survives (cost = f_cost), the earlier points only have to be legal: no failing cell in
              // reach is enough.  Outside the window there is no screen: walk.
              uint32_t m_need = u & m_on & ((b_nf & ((scr_sum ? 0xFFFFFFFFu : 0u) | m_last | b_cf)) | ~m_inw);
#ifdef NAVGPU_SWEEP_X_NOWALK  // timing experiment: the sweep alone
              m_need = 0u;
#endif
              const unsigned long long pm = __ballot(m_need != 0u);
#ifdef NAVGPU_SWEEP_COUNTS
              {
                const unsigned long long pc = __ballot(m_need != 0u && (scr_sum || m_last != 0u)), pl_ = __ballot(m_need != 0u && m_last != 0u);
                if (lane == 0) {
                  atomicAdd(&g_sweep_counts[1], (unsigned long long)__popcll(pm & ~pc));
                  atomicAdd(&g_sweep_counts[2], (unsigned long long)__popcll(pc));
                  atomicAdd(&g_sweep_counts[11], (unsigned long long)__popcll(pl_));
                }
              }
#endif
              if (pm != 0ull) {  // one LDS atomic per wave and step
                const int leader = __ffsll((long long)pm) - 1;
                uint32_t base = 0;
                if ((int)lane == leader) base = atomicAdd(&s_qn[blk & 1u], (uint32_t)__popcll(pm));
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
                const uint32_t slot = base + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull));
                m_stall = m_need & (slot >= (uint32_t)kSweepQueue ? 0xFFFFFFFFu : 0u);  // the queue is full: this point is taken again in the next block
                if ((m_need & ~m_stall) != 0u) {
                  s_qx[slot] = (float)x;
                  s_qy[slot] = (float)y;
                  s_qt[slot] = tid | ((uint32_t)sc << 10) | ((uint32_t)t_row << 17) | (((scr_sum ? 0xFFFFFFFFu : 0u) | m_last) & 0x80000000u);
                }
#ifdef NAVGPU_SWEEP_COUNTS
                if (m_stall != 0u) atomicAdd(&g_sweep_counts[12], 1ull);
#endif
              }
            }
          }
          // ---- the other critics, for the lanes that did not stall and have not failed the obstacle critic
          const uint32_t m_go = u & ~m_stall & (first_fail > 1 ? 0xFFFFFFFFu : 0u);
          {
            // path / goal: a point of the window whose screen bit is clear cannot fail that critic (the bit IS the failure test, taken
            // from this cycle's grid by the prep launch); a set bit fails it: -3 or -2, the grid says which after the rollout.  The last
            // point's cell is remembered (aggregation Last: its distances survive).  Off the map: -4.
            const uint32_t live_p = (en_path && 4 < first_fail) ? 0xFFFFFFFFu : 0u, live_g = (en_goal && 5 < first_fail) ? 0xFFFFFFFFu : 0u;
            const uint32_t m_pg = m_go & (live_p | live_g);
            const uint32_t m_pgoff = m_pg & ~m_on, m_pgon = m_pg & m_on;
            const uint32_t cell = cy * g.nx + cx;
            last_c = sel(m_pgon & m_last, cell, last_c);
            const uint32_t m_scr = m_pgon & ~m_last & m_inw;
            const uint32_t hz = m_scr & live_p & b_z, hw = m_scr & ~hz & live_g & b_w;
            const uint32_t ff_off = sel(live_p, 4u, 5u);
            first_fail = (int)sel(m_pgoff, ff_off, sel(hz, 4u, sel(hw, 5u, (uint32_t)first_fail)));
            fail_code = (int)sel(m_pgoff, (uint32_t)-4, sel(hz | hw, 0u, (uint32_t)fail_code));
            fail_cell = sel(hz | hw, cell, fail_cell);
            const uint32_t m_out = m_pgon & ~m_last & ~m_inw;  // outside the LDS window (never, with a correctly sized one): look the grids up here
            if (__ballot(m_out != 0u) != 0ull) {
              if (m_out != 0u) {
                if (en_path && 4 < first_fail) {
                  const uint32_t d = dpath[cell];
                  if (d == N_obst || d == N_unreach) {
                    fail_code = d == N_obst ? -3 : -2;
                    first_fail = 4;
                  }
                }
                if (en_goal && 5 < first_fail) {
                  const uint32_t d = dgoal[cell];
                  if (d == N_obst || d == N_unreach) {
                    fail_code = d == N_obst ? -3 : -2;
                    first_fail = 5;
                  }
                }
              }
            }
            // the forward point: on the map for sure while the point lies in the window and the window clear of the margin band
            // (margin_on), and only the final point's cell is ever read - so most points that look closer skip its worldToMap
            const uint32_t live_f = ((en_gf && 2 < first_fail) || (en_al && 3 < first_fail)) ? 0xFFFFFFFFu : 0u;
            const uint32_t m_fwd = m_go & live_f & (m_last | (margin_on ? 0xFFFFFFFFu : 0u) | ~m_inw);
            if (__ballot(m_fwd != 0u) != 0ull) {
              if (m_fwd != 0u) {
                double sx = x, sy = y;
                if (fpd != 0.0) {
                  sx = x + fpd * cs;
                  sy = y + fpd * sn;
                }
                uint32_t ux, uy;
                if (!w2m(sx, sy, ux, uy)) {
                  fail_code = -4;
                  first_fail = (en_gf && 2 < first_fail) ? 2 : 3;
                } else if (m_last != 0u) {
                  last_f = uy * g.nx + ux;
                }
              }
            }
            last_f = sel(m_go & ~m_fwd & m_last, 0u, last_f);  // (reached; no forward critic to read for)
          }
          scr_z = first_fail > 4 ? 0xFFFFFFFFu : 0u;  // (always the lane's first_fail put as masks: it only changes in here)
          scr_w = first_fail > 5 ? 0xFFFFFFFFu : 0u;
          // a stalled lane goes back to the point as it was: (float)x is the old px exactly
          px = m_stall != 0u ? (float)x : px;
          py = m_stall != 0u ? (float)y : py;
          step = (int)sel(m_stall, (uint32_t)sc, (uint32_t)step);
          alive_m &= first_fail > min_order ? 0xFFFFFFFFu : 0u;
          SW_STAMP(ss1);
          SW_ACC(4, ss1 - ss0);
          SW_ACC(9, 1);
        }
        alive_m = step < K ? alive_m : 0u;
      }
    }
    SW_STAMP(sb1);
    SW_ACC(3, sb1 - sb0);
    // One barrier ends the block: the queue is complete, and s_more says whether any lane of the workgroup goes on.  The NEXT block's
    // counter and flag are cleared before it.  The counter alternates: a wave that reads the last block's count late (that block
    // skipped its second barrier) reads the 0 that is being written - its queue was empty.  The flag rotates through THREE words: the
    // one cleared here was last read two blocks ago, and every wave has passed a barrier since.
    const uint32_t mi = blk % 3u;
    if (__ballot(alive_m != 0u) != 0ull && lane == 0) s_more[mi] = 1;
    if (tid == 0) {
      s_qn[(blk + 1u) & 1u] = 0;
      s_more[mi == 2u ? 0u : mi + 1u] = 0;
    }
    __syncthreads();
    SW_STAMP(sb2);
    SW_ACC(5, sb2 - sb1);
    const int nq = (int)min(s_qn[blk & 1u], (uint32_t)kSweepQueue);
    const uint32_t more = s_more[mi];
    if (nq == 0) {  // (uniform over the workgroup) nothing to walk, no results to wait for
      if (!more) break;
      continue;
    }
    // ---- walk: the workgroup's lanes take the entries, whoever pushed them
    for (int e = (int)tid; e < nq; e += THREADS) {
      const uint32_t tag = s_qt[e];
      const double x = s_qx[e], y = s_qy[e];
      const int e_step = (int)((tag >> 10) & 0x7Fu), e_row = (int)((tag >> 17) & 0x3FFFu);
      const uint32_t owner = tag & 0x3FFu;
      const int te = e_row * K + e_step;
      uint32_t mx_cost;
      bool outside;
      const bool off = walkWindow(x, y, te, mx_cost, outside);
      if (__builtin_expect(outside && !off, 0)) mx_cost = walkCostmap(x, y, te);
#ifdef NAVGPU_SWEEP_COUNTS
      atomicAdd(&g_sweep_counts[5], 1ull);
      if (off || mx_cost >= walk_fail) atomicAdd(&g_sweep_counts[6], 1ull);
      if (!(off || mx_cost >= walk_fail) && !(tag >> 31)) atomicAdd(&g_sweep_counts[13], 1ull);  // walked for legality only, legal
      if ((tid & 63u) == (uint32_t)(__ffsll((long long)__ballot(true)) - 1)) atomicAdd(&g_sweep_counts[7], 1ull);
#endif
      if (off || mx_cost >= walk_fail) {  // footprint_cost < 0
        atomicOr(&s_obs[owner], kWalkFailed);
      } else if (tag >> 31) {
        // the centre is on the map here (a point off the map never gets into the queue), so the -7 branch
        // (obstacle_cost_function.cpp:135-137) cannot fire; occ_cost = max(max(0, footprint_cost), centre cell)
        const double f_cost = (walk_swap && mx_cost == 254u) ? 255.0 : (double)mx_cost;  // an allowed NO_INFORMATION cell costs 255
        uint32_t cx, cy;
        w2m(x, y, cx, cy);
        const double occ = fmax(fmax(0.0, f_cost), (double)cellCost(cx, cy));
        atomicAdd(&s_obs[owner], (uint32_t)occ);
      }
    }
    SW_STAMP(sb3);
    SW_ACC(6, sb3 - sb2);
    __syncthreads();  // the owners' words are complete
    SW_STAMP(sb4);
    SW_ACC(7, sb4 - sb3);
    if ((s_obs[tid] & kWalkFailed) != 0u && first_fail > 1) {  // a walked point of this lane failed: footprint_cost < 0 -> -6 (also after its last point)
      fail_code = -6;
      first_fail = 1;
      alive_m = 0;
    }
    if (!more) break;  // (taken before the walks: a workgroup whose last live lanes they have just ended comes round once more, finds nobody alive and leaves)
  }

  {  // ---- the distance grids, once: every load in flight before the first is used (clamped addresses; unused ones are ignored)
    const bool reached = last_f != 0xFFFFFFFFu && first_fail > 1;  // the final point was scored (and no walk failed since)
    const uint32_t cmax = pl.cells - 1u;
    const uint32_t v_fail = (first_fail == 4 ? dpath : dgoal)[min(fail_cell, cmax)];
    const uint32_t v_gf = dfront[min(last_f, cmax)], v_al = dpath[min(last_f, cmax)];
    const uint32_t v_path = dpath[min(last_c, cmax)], v_goal = dgoal[min(last_c, cmax)];
    if ((first_fail == 4 || first_fail == 5) && fail_code == 0) fail_code = v_fail == N_obst ? -3 : -2;
    if (reached) {  // in the critics' order; a critic that failed earlier on the trajectory (or follows one that did) is not read
      if (en_gf && 2 < first_fail) d_gf = v_gf;
      if (en_al && 3 < first_fail) d_al = v_al;
      if (en_path && 4 < first_fail) {
        if (v_path == N_obst || v_path == N_unreach) {
          fail_code = v_path == N_obst ? -3 : -2;
          first_fail = 4;
        } else
          d_path = v_path;
      }
      if (en_goal && 5 < first_fail) {
        if (v_goal == N_obst || v_goal == N_unreach) {
          fail_code = v_goal == N_obst ? -3 : -2;
          first_fail = 5;
        } else
          d_goal = v_goal;
      }
    }
  }
  if (in_range && !reject) {
    if (osc_fail) {
      total = -5.0;
    } else if (first_fail < 6) {
      total = (double)fail_code;
    } else {
      // scoreTrajectory's sum in critic order (a term that is 0 is not scaled: `if (cost != 0) cost *= scale`)
      total = 0.0;
      auto add = [&](bool en, double value, double scale) {
        if (!en) return;
        double cost = value;
        if (cost != 0) cost *= scale;
        total += cost;
      };
      add(en_obs, (double)(s_obs[tid] & ~kWalkFailed), sc_obs);
      add(en_gf, (double)d_gf, sc_gf);
      add(en_al, (double)d_al, sc_al);
      add(en_path, (double)d_path, sc_path);
      add(en_goal, (double)d_goal, sc_goal);
    }
  }
  if (in_range && pl.sample_cost) {
    pl.sample_cost[(size_t)inst * pl.max_samples + sidx] = total;
    pl.sample_status[(size_t)inst * pl.max_samples + sidx] = status;
  }

  // ---- workgroup argmin (lowest index wins ties == first strict minimum of the sequential loop)
  const bool valid = in_range && status == NAVGPU_SAMPLE_SCORED && total >= 0.0;
  double bc = valid ? total : 1.0e300;
  int bi = valid ? sidx : 0x7FFFFFFF;
  for (int off = 32; off > 0; off >>= 1) {
    double oc = __shfl_down(bc, off);
    int oi = __shfl_down(bi, off);
    if (oc < bc || (oc == bc && oi < bi)) {
      bc = oc;
      bi = oi;
    }
  }
  const unsigned long long m_scored = __ballot(in_range && status == NAVGPU_SAMPLE_SCORED);
  const unsigned long long m_valid = __ballot(valid);
  if ((tid & 63) == 0) {
    s_rc[tid >> 6] = bc;
    s_ri[tid >> 6] = bi;
    atomicAdd(&s_cnt[0], __popcll(m_scored));
    atomicAdd(&s_cnt[1], __popcll(m_valid));
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < THREADS / 64; ++w)
      if (s_rc[w] < bc || (s_rc[w] == bc && s_ri[w] < bi)) {
        bc = s_rc[w];
        bi = s_ri[w];
      }
    pl.part_cost[(size_t)inst * pl.score_blocks + part_slot] = bc;
    pl.part_index[(size_t)inst * pl.score_blocks + part_slot] = bi;
    if (s_cnt[0]) atomicAdd(&pl.counters[2 * inst], s_cnt[0]);
    if (s_cnt[1]) atomicAdd(&pl.counters[2 * inst + 1], s_cnt[1]);
  }
#ifdef NAVGPU_SWEEP_TIMING
  if ((tid & 63u) == 0 && (blockIdx.x & 3u) == 1 && (blockIdx.z & 3u) == 2) {
    const unsigned long long sw9 = __builtin_amdgcn_s_memtime();
    sw_t[8] = sw9 - sw0;
    for (int i = 0; i < 10; ++i) atomicAdd(&g_sweep_stats[i], sw_t[i]);
    atomicAdd(&g_sweep_stats[10], 1ull);
  }
#endif
}

// The sweep takes every launch of the table variant (use_dwa, discretize_by_time, DWAPlanner's own MapGrid options): its tags
// hold 10 bits of lane, 7 of step and 14 of table row.
static uint32_t score_sweep_blocks(const PlannerDev& pl) {
  const uint32_t max_nxy = pl.max_samples / std::max(pl.tab_nth, 1u), groups = (pl.tab_nth + std::max(pl.tab_rows, 1u) - 1) / std::max(pl.tab_rows, 1u);
  return groups * ((max_nxy * pl.tab_rows + kSweepThreads - 1) / kSweepThreads);
}
bool score_sweep_applies(const PlannerDev& pl) {
  return pl.use_tables && !pl.mg_generic && score_sweep_blocks(pl) <= pl.score_blocks && pl.tab_steps >= 1 && pl.tab_steps <= 127 && pl.tab_rows < (1u << 14) && kSweepThreads <= 1024;
}
uint32_t launch_score_sweep(const PlannerDev& pl, uint32_t first, uint32_t count, hipStream_t s) {
  const size_t lds = score_window_bytes(pl.win) + score_table_lds_bytes(pl);
  // row groups x workgroups per group, for the largest (vx, vy) grid the configuration can produce
  const uint32_t max_nxy = pl.max_samples / std::max(pl.tab_nth, 1u), groups = (pl.tab_nth + pl.tab_rows - 1) / pl.tab_rows;
  const uint32_t bpg = (max_nxy * pl.tab_rows + kSweepThreads - 1) / kSweepThreads;
  const uint32_t blocks = groups * bpg;  // (<= score_blocks: score_sweep_applies)
#define NAVGPU_SCORE_SWEEP(C)                                                                                                             \
  {                                                                                                                                       \
    if (lds > 40 * 1024) {                                                                                                                \
      hipFuncSetAttribute((const void*)k_score_sweep<C, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
      hipFuncSetAttribute((const void*)k_score_sweep<C, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
    }                                                                                                                                     \
    if (pl.cfg.rollout_trig)                                                                                                              \
      hipLaunchKernelGGL((k_score_sweep<C, true>), dim3(bpg, groups, count), dim3(kSweepThreads), lds, s, pl, first);                      \
    else                                                                                                                                  \
      hipLaunchKernelGGL((k_score_sweep<C, false>), dim3(bpg, groups, count), dim3(kSweepThreads), lds, s, pl, first);                     \
  }
  if (pl.fp_chunk <= 6) NAVGPU_SCORE_SWEEP(6)
  else if (pl.fp_chunk <= 9) NAVGPU_SCORE_SWEEP(9)
  else if (pl.fp_chunk <= 12) NAVGPU_SCORE_SWEEP(12)
  else NAVGPU_SCORE_SWEEP(16)
#undef NAVGPU_SCORE_SWEEP
  return blocks;
}

#ifdef NAVGPU_SWEEP_COUNTS
extern "C" int navgpu_debug_sweep_counts(unsigned long long* out16, int reset) {
  if (out16) hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_sweep_counts), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_sweep_counts), z, sizeof(z));
  }
  return 0;
}
#endif
#ifdef NAVGPU_SWEEP_TIMING
extern "C" int navgpu_debug_sweep_stats(unsigned long long* out16, int reset) {
  if (out16) hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_sweep_stats), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_sweep_stats), z, sizeof(z));
  }
  return 0;
}
#endif

}  // namespace navgpu
