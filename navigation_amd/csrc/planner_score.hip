// k_score (gfx950): generateTrajectory + the six DWA critics, one lane per velocity sample.
// Compiled with -ffp-contract=off: the fp64 step arithmetic on fp32 state has to round exactly
// like the reference (simple_trajectory_generator.cpp:253-260, SURVEY 7 hard part 2).
#include "planner_score.h"

namespace navgpu {

// ------------------------------------------------------------------------------------------------
// k_score: one lane per velocity sample.
//   SimpleTrajectoryGenerator::generateTrajectory / computeNewPositions / computeNewVelocities
//     (simple_trajectory_generator.cpp:180-276)
//   SimpleScoredSamplingPlanner::scoreTrajectory (simple_scored_sampling_planner.cpp:50-79) with the
//     critic order of dwa_planner.cpp:167-173: oscillation, obstacle, goal_front, alignment, path, goal
//   ObstacleCostFunction::scoreTrajectory/footprintCost (obstacle_cost_function.cpp:74-142),
//   WorldModel::footprintCost (world_model.h:65-86), CostmapModel::footprintCost/lineCost/pointCost
//     (costmap_model.cpp:50-142), LineIterator (line_iterator.h:38-139)
//   MapGridCostFunction::scoreTrajectory (map_grid_cost_function.cpp:75-129, aggregation Last)
//   OscillationCostFunction::scoreTrajectory (oscillation_cost_function.cpp:166-176)
// Every sample is scored in full (no early-out against the incumbent): critic terms are
// non-negative, so the first strict minimum is the same sample the reference keeps.
// The costmap window the trajectories can reach is staged in LDS; cells outside it (never needed
// with a correctly sized window) fall back to a global load, so results never depend on it.
// ------------------------------------------------------------------------------------------------
#if defined(NAVGPU_SCORE_TIMING) && !defined(NAVGPU_SCORE_STATS)  // phase stamps only (no counters in the loop)
__device__ unsigned long long g_score_stats[24];
#endif
#ifdef NAVGPU_SCORE_STATS  // experiment builds only (make EXTRA=-DNAVGPU_SCORE_STATS, tools/probe_score_stats.py)
__device__ unsigned long long g_score_stats[24];  // lane-steps, unscreened lanes, wave-steps, waves with an unscreened lane, walk lanes, waves with a walk, last-step waves
#define SCORE_STAT(i, v) atomicAdd(&g_score_stats[i], (unsigned long long)(v))
#else
#define SCORE_STAT(i, v)
#endif
#ifdef NAVGPU_PREP_TIMING  // experiment builds only (tools/probe_prep_timing.py): where k_score_prep_tab's time goes (one workgroup in 16 reports)
__device__ unsigned long long g_prep_stats[16];
#define PREP_STAMP(i) \
  if (PREP == 1) prep_t[i] = wall_clock64()
#else
#define PREP_STAMP(i)
#endif
template <bool EXPLICIT, bool TABLES, int THREADS, int PREP = 0, int CHUNK = 12, bool AGG = false>
__device__ __forceinline__ void score_body(const PlannerDev& pl, uint32_t first, const float* explicit_sample) {
  extern __shared__ __align__(16) uint8_t s_dyn[];
  uint8_t* s_win = s_dyn;
  __shared__ double s_fp[2 * kMaxFootprint];
  __shared__ float s_axis[3][kMaxAxis];
  __shared__ double s_rc[THREADS / 64];
  __shared__ int s_ri[THREADS / 64];
  __shared__ int s_cnt[2];

  const uint32_t inst = first + blockIdx.y;
  const uint32_t tid = threadIdx.x;
#ifdef NAVGPU_PREP_TIMING
  unsigned long long prep_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  PREP_STAMP(0);
#ifdef NAVGPU_SCORE_TIMING
  const unsigned long long ts0 = wall_clock64();
#endif
  // A scoring workgroup's prologue (a few dependent loads, the image copy, barriers) is a handful of instructions, but its
  // waves are the YOUNGEST on their SIMDs and lose every issue arbitration against the five older workgroups in their
  // rollout loops: measured 40 % of a workgroup's residence before its first trajectory point.  Raised priority until the
  // image is in place gets it out of the way.
  if (PREP == 2) __builtin_amdgcn_s_setprio(3);
  const navgpu_dwa_config& c = pl.cfg;
  const Geom g = geomOf(pl, inst);
  const navgpu_robot_state st = pl.state[inst];
  const uint8_t* master = pl.master + (size_t)inst * pl.cells_padded;
  const uint32_t* dpath = pl.path + (size_t)inst * pl.cells;
  const uint32_t* dgoal = pl.goal + (size_t)inst * pl.cells;
  const uint32_t* dfront = pl.goal_front + (size_t)inst * pl.cells;
  const int32_t* cnt = pl.axis_count + 4 * inst;
  const int n_samples = EXPLICIT ? 1 : cnt[3];
  const uint32_t nfp = pl.fp_n[inst];
  const int win = (int)pl.win;

  // ---- stage: footprint, per-axis samples, costmap window around the robot
  // A scoring workgroup (PREP 2) keeps what it loads here in registers and writes it to LDS together with its image
  // further down: ONE batch of loads in flight and one barrier instead of five dependent round trips and three barriers
  // (a load takes several microseconds while 24 waves per CU gather from the distance grids; measured 40 % of a
  // workgroup's residence was spent before its first trajectory point).
  constexpr int kAxisChunks = (3 * kMaxAxis + THREADS - 1) / THREADS;
  double pre_fp = 0.0;
  float pre_axis[kAxisChunks];
  if (PREP == 2) {
    pre_fp = pl.fp_spec[(size_t)inst * kMaxFootprint * 2 + (tid < 2 * nfp ? tid : 0)];
#pragma unroll
    for (int u = 0; u < kAxisChunks; ++u) {
      const uint32_t i = min(tid + (uint32_t)u * THREADS, 3u * kMaxAxis - 1), a = i / kMaxAxis, k = i - a * kMaxAxis;
      pre_axis[u] = pl.axis_samples[((size_t)inst * 3 + a) * pl.max_axis + min(k, pl.max_axis - 1)];
      if (k >= pl.max_axis) pre_axis[u] = 0.f;
    }
  } else {
    if (tid < 2 * nfp) s_fp[tid] = pl.fp_spec[(size_t)inst * kMaxFootprint * 2 + tid];
    if (!EXPLICIT) {
      for (uint32_t i = tid; i < 3 * kMaxAxis; i += blockDim.x) {
        uint32_t a = i / kMaxAxis, k = i - a * kMaxAxis;
        s_axis[a][k] = k < pl.max_axis ? pl.axis_samples[((size_t)inst * 3 + a) * pl.max_axis + k] : 0.f;
      }
    }
  }
  if (tid == 0) s_cnt[0] = s_cnt[1] = 0;
  int wx0 = 0, wy0 = 0;
  {
    // window origin: robot cell (floor of the map coordinate, also valid when the robot is off the map)
    double fx = floor(((double)st.pos[0] - g.ox) / g.res), fy = floor(((double)st.pos[1] - g.oy) / g.res);
    fx = fmin(fmax(fx, -1.0e6), 1.0e6);
    fy = fmin(fmax(fy, -1.0e6), 1.0e6);
    wx0 = (int)fx - win / 2;
    wy0 = (int)fy - win / 2;
    // (eight loads of a lane in flight at a time - unconditional, clamped: a conditional load in a rolled loop is waited
    // for on its own, nine latencies in a row for a 65 x 65 window)
    for (int i0 = tid; PREP != 2 && i0 < win * win; i0 += 8 * (int)blockDim.x) {
      uint8_t v[8];
      bool in_map[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int ic = min(i0 + u * (int)blockDim.x, win * win - 1);
        const int ly = ic / win, lx = ic - ly * win;
        const int gx = wx0 + lx, gy = wy0 + ly;
        in_map[u] = gx >= 0 && gy >= 0 && gx < (int)g.nx && gy < (int)g.ny;
        v[u] = master[min(max(gy, 0), (int)g.ny - 1) * g.nx + min(max(gx, 0), (int)g.nx - 1)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * (int)blockDim.x;
        if (i < win * win) s_win[i] = in_map[u] ? v[u] : (uint8_t)0;
      }
    }
  }
  // ---- per-cell screens of the window, four bitmaps interleaved per 32-cell word: s_fb[(y * nw + j) * 4 + k]
  // Footprint shortcuts: the cells a footprint with centre cell c can touch lie within the Chebyshev radius
  // fp_rcells of c (>= circumscribed radius in cells + 1).  Two bitmaps of the window, dilated by that radius, are
  // built once per robot (bit-parallel: rows of 32-cell words, shifts for the horizontal pass, word ORs for the
  // vertical one; everything outside the window or off the map counts as set):
  //   k = 0: some cell in reach is not FREE_SPACE  -> clear = the point's footprint cost is exactly 0
  //   k = 1: some cell in reach fails pointCost    -> clear = the point is legal (cost not needed)
  // MapGrid screens (not dilated): a trajectory point only has to NOT be an obstacle / unreachable cell of the path and
  // goal grids unless it is the last one (aggregation Last, map_grid_cost_function.cpp:92-127):
  //   k = 2: path grid holds obstacleCosts() or unreachableCellCosts() here    k = 3: the goal grid does
  const int win_bytes = (win * win + 15) & ~15;
  const int nw = (win + 31) >> 5;
  uint32_t* s_fb = reinterpret_cast<uint32_t*>(s_dyn + win_bytes);  // [win][nw][4]
  // build scratch behind the image (window + bitmaps + tables): [win][nw][2] raw, [win][nw][2] after the horizontal pass
  uint32_t* s_ba = reinterpret_cast<uint32_t*>(s_dyn + win_bytes + score_bits_bytes(win) + (TABLES ? pl.tab_bytes : 0u));
  const int rc = (int)pl.fp_rcells;
  const uint8_t fail_span_w = (pl.cfg.allow_unknown != 0) ? 0 : 1;
  if (PREP != 2) __syncthreads();
  PREP_STAMP(1);  // staged: footprint, axis samples, window bytes
#ifdef NAVGPU_SCORE_TIMING
  const unsigned long long ts0a = wall_clock64();
#endif
  if (PREP != 2) {
  {  // the raw bits: 64 consecutive cells of a (padded) window row per wave step, packed by ballot
    const int row_cells = nw * 32;
    const bool obst_off = pl.scale_obstacle == 0;  // obstacle critic off (scale 0: skipped, simple_scored_sampling_planner.cpp:55-57): nothing to screen
    for (int base = (int)(tid & ~63u); base < win * row_cells; base += (int)blockDim.x) {
      const int idx = base + (int)(tid & 63u);
      const int y = idx / row_cells, lx = idx - y * row_cells;
      // cells off the MAP count as set like cells outside the window: a footprint vertex there fails
      // worldToMap, i.e. footprintCost = -1 (costmap_model.cpp:77-99), which only the polygon walk reports
      const int gx = wx0 + lx, gy = wy0 + y;
      const bool in = lx < win && y < win && gx >= 0 && gy >= 0 && gx < (int)g.nx && gy < (int)g.ny;
      const uint8_t cc = s_win[in ? y * win + lx : 0];
      const bool o = !obst_off && (!in || cc != 0), f = !obst_off && (!in || (uint8_t)(cc - kLethal) <= fail_span_w);
      const unsigned long long mo = __ballot(o), mf = __ballot(f);
      if ((tid & 63u) == 0) {
        const int w = idx >> 5;  // linear word index y * nw + j; a wave covers two words (possibly of two rows)
        s_ba[2 * w] = (uint32_t)mo;
        s_ba[2 * w + 1] = (uint32_t)mf;
        if (w + 1 < win * nw) {
          s_ba[2 * (w + 1)] = (uint32_t)(mo >> 32);
          s_ba[2 * (w + 1) + 1] = (uint32_t)(mf >> 32);
        }
      }
    }
  }
  __syncthreads();
  PREP_STAMP(2);  // raw bits
  // The dilation: a DISC (PlannerDev::fp_halfw: per row offset dy the largest |dx| an outline cell can have), not the Chebyshev
  // square around it - a fifth to a quarter fewer cells, and every cell less is trajectory points that need not be looked at.
  // Grouped by dx: the rows that contribute at |dx| = d are |dy| <= Y(d) (the half widths fall with |dy|), so the rows are OR-ed
  // in from the centre outwards (the word and its two neighbours) and row offset Y shifts the running OR by the d's it is the
  // last row for: halfw[Y + 1] < d <= halfw[Y].  One shift pair per d instead of one per cell of the disc.
  for (int it = tid; it < 2 * win * nw; it += blockDim.x) {
    const int cell = it >> 1;
    const int y = cell / nw, j = cell - y * nw;
    uint32_t m = (y - rc < 0 || y + rc >= win || rc > 31) ? 0xFFFFFFFFu : 0u;  // (reach beyond the neighbouring words: no shortcut)
    if (!m) {
      uint32_t vc = 0, vl = j > 0 ? 0u : 0xFFFFFFFFu, vr = j + 1 < nw ? 0u : 0xFFFFFFFFu;
      for (int Y = 0; Y <= rc; ++Y) {
        const int w = (int)pl.fp_halfw[Y];
        if (w == 0xFF) break;
        const int up = it + 2 * Y * nw, dn = it - 2 * Y * nw;
        vc |= s_ba[up] | s_ba[dn];
        if (j > 0) vl |= s_ba[up - 2] | s_ba[dn - 2];
        if (j + 1 < nw) vr |= s_ba[up + 2] | s_ba[dn + 2];
        const int wn = (Y + 1 <= rc && pl.fp_halfw[Y + 1] != 0xFF) ? (int)pl.fp_halfw[Y + 1] : -1;  // (the outermost row: every d down to 0)
        for (int d = wn + 1; d <= w; ++d) m |= d == 0 ? vc : (__builtin_amdgcn_alignbit(vc, vl, 32 - d) | __builtin_amdgcn_alignbit(vr, vc, d));
      }
    }
    s_fb[4 * cell + (it & 1)] = m;
  }
  PREP_STAMP(3);  // dilation
  {  // MapGrid screens: 64 consecutive cells of a (padded) window row per wave step, packed by ballot
    const uint32_t n_obst = pl.cells, n_unreach = pl.cells + 1;
    const int row_cells = nw * 32;
    // (four steps of a wave = eight distance loads per lane in flight, unconditional and clamped, as for the window above)
    for (int base0 = (int)(tid & ~63u); base0 < win * row_cells; base0 += 4 * (int)blockDim.x) {
      uint32_t dp[4], dg[4];
      bool in_map[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = min(base0 + u * (int)blockDim.x + (int)(tid & 63u), win * row_cells - 1);
        const int y = idx / row_cells, lx = idx - y * row_cells;
        const int gx = wx0 + lx, gy = wy0 + y;
        in_map[u] = lx < win && gx >= 0 && gy >= 0 && gx < (int)g.nx && gy < (int)g.ny;
        const uint32_t cell = (uint32_t)(min(max(gy, 0), (int)g.ny - 1)) * g.nx + (uint32_t)min(max(gx, 0), (int)g.nx - 1);
        dp[u] = dpath[cell];
        dg[u] = dgoal[cell];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int base = base0 + u * (int)blockDim.x;
        if (base < win * row_cells) {  // (wave-uniform)
          const int idx = base + (int)(tid & 63u);
          bool pf = true, gf = true;
          if (in_map[u] && idx < win * row_cells) {
            pf = dp[u] == n_obst || dp[u] == n_unreach;
            gf = dg[u] == n_obst || dg[u] == n_unreach;
          }
          pf = pf && pl.scale_path != 0;  // a critic with scale 0 is never evaluated: its screen stays clear
          gf = gf && pl.scale_goal != 0;
          const unsigned long long mp = __ballot(pf), mg = __ballot(gf);
          if ((tid & 63u) == 0) {
            const int w = idx >> 5;  // linear word index y * nw + j; a wave covers two words (possibly of two rows)
            s_fb[4 * w + 2] = (uint32_t)mp;
            s_fb[4 * w + 3] = (uint32_t)mg;
            if (w + 1 < win * nw) {
              s_fb[4 * (w + 1) + 2] = (uint32_t)(mp >> 32);
              s_fb[4 * (w + 1) + 3] = (uint32_t)(mg >> 32);
            }
          }
        }
      }
    }
  }
  }
  // The LDS window is kept in "walk order": with allow_unknown the bytes 254 (LETHAL) and 255 (NO_INFORMATION)
  // are swapped, so that in both modes a footprint cell fails pointCost iff its stored byte >= walk_fail and the
  // polygon walk needs nothing but a running maximum per cell (cellCost() undoes the swap).
  const bool walk_swap = pl.cfg.allow_unknown != 0;
  const uint32_t walk_fail = walk_swap ? 255u : 254u;
  if (PREP != 2 && walk_swap) {
    __syncthreads();
    for (int i = tid; i < win * win; i += blockDim.x) {
      const uint8_t cc = s_win[i];
      if (cc >= 254) s_win[i] = cc ^ 1u;
    }
  }
  // ---- TABLES: per-(v_theta sample, step) heading, trig, rotated footprint, forward-point offset
  const int K = TABLES ? (int)pl.tab_steps : 0;
  const int tnfp = TABLES ? (int)pl.tab_nfp : 0;
  const int nth_s = TABLES ? cnt[2] : 0;
  // rows of the tables in LDS: all tab_nth where they are built (PREP 1); the scoring launch (PREP 2) keeps only the
  // tab_rows v_theta rows of its workgroup's row group (see the lane mapping below)
  const int lrows = TABLES ? (PREP == 2 ? (int)pl.tab_rows : (int)pl.tab_nth) : 0;
  double* s_trig = reinterpret_cast<double*>(s_dyn + win_bytes + score_bits_bytes(win));  // [rows][K][4] cs, sn, cs2, sn2
  double* s_rot = s_trig + (size_t)lrows * K * 4;                                 // [rows][K][tnfp][2]
  float* s_th = reinterpret_cast<float*>(s_rot + (size_t)lrows * K * tnfp * 2);       // [rows][K]
  // TABLES lane mapping.  Lanes are v_theta-major so that a wave shares one heading sequence.  The v_theta rows are cut
  // into groups of tab_rows (what the LDS budget holds: all of them for configs[2]'s 17, 17 of configs[4]'s 33); a
  // group takes bpg consecutive workgroups, which enumerate its rows x (vx, vy) pairs.  Blocks past the last group idle.
  int t_row_base = 0, t_rows = 0, t_li0 = 0;
  if (TABLES && PREP == 2) {
    const int nxy = max(cnt[0] * cnt[1], 1), R = (int)pl.tab_rows;
    const int bpg = (nxy * R + (int)blockDim.x - 1) / (int)blockDim.x;
    const int gi = (int)blockIdx.x / bpg;
    t_row_base = gi * R;
    t_rows = min(max(cnt[2] - t_row_base, 0), R);
    t_li0 = ((int)blockIdx.x - gi * bpg) * (int)blockDim.x;
    if (t_li0 >= t_rows * nxy) {  // no sample for this workgroup (the last group's share is rounded up to the largest): no image either
      if (tid == 0) {
        pl.part_cost[(size_t)inst * pl.score_blocks + blockIdx.x] = 1.0e300;
        pl.part_index[(size_t)inst * pl.score_blocks + blockIdx.x] = 0x7FFFFFFF;
      }
      return;
    }
  }
  if (TABLES && PREP != 2) {
    __syncthreads();  // s_axis, s_fp staged
    PREP_STAMP(4);  // MapGrid screens (+ walk-order swap)
    const double dt_t = c.sim_time / K;
    if ((int)tid < nth_s) {
      float pth = st.pos[2];
      const float vth = s_axis[2][tid];
      for (int k = 0; k < K; ++k) {
        s_th[tid * K + k] = pth;
        pth = (float)(pth + vth * dt_t);  // computeNewPositions :258
      }
    }
    __syncthreads();
    PREP_STAMP(5);  // heading sequences
    for (int e = tid; e < nth_s * K; e += blockDim.x) {
      const double th = s_th[e];
      double sn, cs, sn2, cs2;
      sincos(th, &sn, &cs);
      sincos(M_PI_2 + th, &sn2, &cs2);
      s_trig[4 * e] = cs;
      s_trig[4 * e + 1] = sn;
      s_trig[4 * e + 2] = cs2;
      s_trig[4 * e + 3] = sn2;
      for (int v = 0; v < (int)nfp && v < tnfp; ++v) {
        const double sx = s_fp[2 * v], sy = s_fp[2 * v + 1];
        s_rot[(e * tnfp + v) * 2] = sx * cs - sy * sn;      // world_model.h:72-73
        s_rot[(e * tnfp + v) * 2 + 1] = sx * sn + sy * cs;
      }
    }
  }
  if (PREP != 2) __syncthreads();
  PREP_STAMP(6);  // trig + rotated footprints
#ifdef NAVGPU_SCORE_TIMING
  const unsigned long long ts0b = wall_clock64();
#endif
  if (PREP != 0) {  // the LDS image as 16-byte words: [0, prep_bytes)
    uint4* img = reinterpret_cast<uint4*>(pl.prep + (size_t)inst * pl.prep_stride);
    uint4* lds = reinterpret_cast<uint4*>(s_dyn);
    const uint32_t n16 = pl.prep_bytes >> 4;
    if (PREP == 1) {
      for (uint32_t i = tid; i < n16; i += blockDim.x) img[i] = lds[i];
      if (TABLES) {
        // What every lane of the sweep launch would otherwise work out again: the robot's scalars (two fp64 divisions, a ceil)
        // and the generator's reject tests of the (vx, vy) pairs (a compensated fp64 square root each) - behind the image.
        uint8_t* rej = pl.prep + (size_t)inst * pl.prep_stride + score_prep_reject_offset(pl);
        int32_t* aux = reinterpret_cast<int32_t*>(rej - kScoreAuxBytes);
        // d0: how far (Euclidean, cells, rounded down) from the robot's own cell - the window's centre - the nearest cell lies at which ANY screen is
        // set (or the window ends).  A trajectory point fewer cells away than that passes every screen whatever else: the sweep skips
        // its worldToMap and look-up for as many steps as the sample's speed cannot cover d0 cells in (k_score_sweep).
        // The path / goal screen at the robot's own cell decides those critics for EVERY sample at step 0 (all rollouts start there):
        // start_fail = 4 (the path critic fails there; the goal critic, later in the list, no longer counts either) or 5 (the goal
        // critic does - e.g. a goal inside an inflated wall, 17 % of the benchmark's robots) or 0.  The distance is taken over the
        // screens that still count after that.
        const int c0 = win / 2;
        const uint32_t w_c0 = (uint32_t)(c0 * nw + (c0 >> 5));
        const int start_fail = ((s_fb[4 * w_c0 + 2] >> (c0 & 31)) & 1u) ? 4 : (((s_fb[4 * w_c0 + 3] >> (c0 & 31)) & 1u) ? 5 : 0);
        if (tid == 0) s_cnt[0] = (win / 2) * (win / 2);  // (squared: the distance is Euclidean - a pose moves |v| dt whatever its bearing)
        __syncthreads();
        PREP_STAMP(7);  // image stored
        {
          int dmin = win * win;
          for (int it = tid; it < win * nw; it += blockDim.x) {
            const int y = it / nw, j = it - y * nw;
            uint32_t m = (c.sum_scores ? s_fb[4 * it] : s_fb[4 * it + 1]) | (start_fail == 4 ? 0u : s_fb[4 * it + 2]) | (start_fail != 0 ? 0u : s_fb[4 * it + 3]);
            const int dy = y > c0 ? y - c0 : c0 - y;
            // the set cell nearest to column c0: the lowest set bit at or right of it, the highest left of it (no loop over the bits)
            const int p = c0 - 32 * j;  // c0's bit position in this word (may lie outside it)
            const uint32_t right = p <= 0 ? m : (p >= 32 ? 0u : m & (0xFFFFFFFFu << p)), left = m & ~right;
            if (right) dmin = min(dmin, (32 * j + __ffs(right) - 1 - c0) * (32 * j + __ffs(right) - 1 - c0) + dy * dy);
            if (left) dmin = min(dmin, (c0 - (32 * j + 31 - __clz(left))) * (c0 - (32 * j + 31 - __clz(left))) + dy * dy);
          }
          atomicMin(&s_cnt[0], dmin);
        }
        __syncthreads();
        if (tid == 0) {
          const double inv_res = pl.inv_res, fpd = c.forward_point_distance;
          const bool en_fwd = pl.scale_goal != 0 || (pl.align_on[inst] && pl.scale_path != 0);  // goal_front or alignment critic on
          // the forward point (x + fpd cos, y + fpd sin) stays on the map whenever the centre cell is this many cells away from
          // every border; only then may a step skip its worldToMap
          const uint32_t fwd_margin = (uint32_t)fmin(ceil(fabs(fpd) * inv_res) + 1.0, 1.0e6);
          const bool fwd_screen = !en_fwd || (2u * fwd_margin < g.nx && 2u * fwd_margin < g.ny);
          const uint32_t fwd_lo = en_fwd ? fwd_margin : 0u, fwd_nx = g.nx - 2u * fwd_lo, fwd_ny = g.ny - 2u * fwd_lo;
          // the forward-margin test is only needed when the LDS window reaches into the margin band of the map
          const bool need_margin = !((uint32_t)wx0 - fwd_lo < fwd_nx && (uint32_t)(wx0 + win - 1) - fwd_lo < fwd_nx && (uint32_t)wy0 - fwd_lo < fwd_ny &&
                                     (uint32_t)(wy0 + win - 1) - fwd_lo < fwd_ny);
          aux[0] = wx0;
          aux[1] = wy0;
          aux[2] = (int32_t)fwd_lo;
          aux[3] = (int32_t)fwd_nx;
          aux[4] = (int32_t)fwd_ny;
          aux[5] = need_margin ? 1 : 0;
          aux[6] = fwd_screen ? 1 : 0;
          aux[7] = (int)floor(sqrt((double)s_cnt[0]));
          aux[8] = start_fail;
        }
        // generateTrajectory's reject tests (simple_trajectory_generator.cpp:193-200), the part that depends on (vx, vy) only:
        // bit 0: vmag + eps < min_trans_vel (rejects together with the v_theta half), bit 1: vmag - eps > max_trans_vel
        const int nyv = max(cnt[1], 1), nxyv = cnt[0] * cnt[1];
        for (int i = tid; i < nxyv; i += blockDim.x) {
          const int ix = i / nyv, iy = i - ix * nyv;
          const double vmag = hyp2((double)s_axis[0][ix], (double)s_axis[1][iy]);
          const double eps = 1e-4;
          rej[i] = (uint8_t)(((c.min_trans_vel >= 0 && vmag + eps < c.min_trans_vel) ? 1 : 0) | ((c.max_trans_vel >= 0 && vmag - eps > c.max_trans_vel) ? 2 : 0));
        }
#ifdef NAVGPU_PREP_TIMING
        PREP_STAMP(8);  // free distance, scalars, reject bytes
        if (PREP == 1 && tid == 0 && (blockIdx.y & 15u) == 3u) {
          for (int i = 0; i < 8; ++i) atomicAdd(&g_prep_stats[i], prep_t[i + 1] - prep_t[i]);
          atomicAdd(&g_prep_stats[15], 1ull);
        }
#endif
      }
      return;
    }
    if (!TABLES) {
      for (uint32_t i = tid; i < n16; i += blockDim.x) lds[i] = img[i];
      if (PREP == 2) {
        if (tid < 2 * nfp) s_fp[tid] = pre_fp;
#pragma unroll
        for (int u = 0; u < kAxisChunks; ++u) {
          const uint32_t i = tid + (uint32_t)u * THREADS;
          if (i < 3u * kMaxAxis) s_axis[i / kMaxAxis][i % kMaxAxis] = pre_axis[u];
        }
      }
    } else {
      // window + screens, and of the tables only the v_theta rows this workgroup's samples use (lanes are
      // v_theta-major: 512 lanes of a 33 x 33 (vx, vy) grid span two of the 17 rows), at their usual place
      // r0..r1: rows of the group (relative to its first) that this workgroup's lanes use
      const int nxy = max(cnt[0] * cnt[1], 1), last = max(t_rows, 1) - 1;
      const int r0 = min(t_li0 / nxy, last);
      const int r1 = min((t_li0 + (int)blockDim.x - 1) / nxy, last);
      const int n16w = (int)((win_bytes + score_bits_bytes(win)) >> 4);
      const int ncopy = t_rows > 0 ? r1 - r0 + 1 : 0;                // (a workgroup past the last row group has no rows)
      const int n_trig = ncopy * K * 2, n_rot = ncopy * K * tnfp;  // 32 B per entry, 16 B per vertex
      const int l_trig = n16w + r0 * K * 2, g_trig = n16w + (t_row_base + r0) * K * 2;
      const int l_rot = n16w + lrows * K * 2 + r0 * K * tnfp, g_rot = n16w + (int)pl.tab_nth * K * 2 + (t_row_base + r0) * K * tnfp;
      const int n16t = n16w + n_trig + n_rot;
      const float* g_th = reinterpret_cast<const float*>(img + n16w + (size_t)pl.tab_nth * K * (2 + tnfp)) + t_row_base * K;
      auto srcOf = [&](int i) { return i < n16w ? i : (i < n16w + n_trig ? g_trig + (i - n16w) : g_rot + (i - n16w - n_trig)); };
      auto dstOf = [&](int i) { return i < n16w ? i : (i < n16w + n_trig ? l_trig + (i - n16w) : l_rot + (i - n16w - n_trig)); };
      constexpr int kBatch = 4;  // 16-byte loads a lane has in flight (4 x 256 lanes x 16 B = 16 KB: a configs[2] image whole)
      uint4 v[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) v[u] = img[srcOf(min((int)tid + u * THREADS, n16t - 1))];
      const int th_i = r0 * K + (int)tid, th_n = (r0 + ncopy) * K;
      const float th_v = g_th[min(th_i, max(th_n - 1, 0))];
#pragma unroll
      for (int u = 0; u < kBatch; ++u)
        if ((int)tid + u * THREADS < n16t) lds[dstOf((int)tid + u * THREADS)] = v[u];
      if (th_i < th_n) s_th[th_i] = th_v;
      if (tid < 2 * nfp) s_fp[tid] = pre_fp;
#pragma unroll
      for (int u = 0; u < kAxisChunks; ++u) {
        const uint32_t i = tid + (uint32_t)u * THREADS;
        if (i < 3u * kMaxAxis) s_axis[i / kMaxAxis][i % kMaxAxis] = pre_axis[u];
      }
      for (int i = (int)tid + kBatch * THREADS; i < n16t; i += blockDim.x) lds[dstOf(i)] = img[srcOf(i)];  // larger images: the rest
      for (int i = th_i + (int)blockDim.x; i < th_n; i += blockDim.x) s_th[i] = g_th[i];
    }
    __syncthreads();
  }

  if (PREP == 2) __builtin_amdgcn_s_setprio(0);
#ifdef NAVGPU_SCORE_TIMING
  const unsigned long long ts1 = wall_clock64();
#endif
  // NOTE: the LDS read is unconditional (clamped index) and the global fallback sits in its own
  // rarely-taken branch; a `cond ? lds[i] : global[j]` form makes hipcc merge both into one FLAT load.
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
  // Costmap2D::worldToMap with the two fp64 divisions replaced by a multiply; exact: whenever the
  // product is not clear of an integer by 1e-7 (error bound 5e-10 below 1e6 cells) the division is redone.
  // Straight-line: the only branch is the rare redo.
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
  const uint8_t fail_span = (pl.cfg.allow_unknown != 0) ? 0 : 1;  // pointCost: 254, and 255 unless allow_unknown

  // lane -> sample slot.  TABLES: v_theta-major so that a wave shares one heading sequence; the slot
  // index (x-outer, y, theta-inner, as the reference enumerates) is what results are keyed by.
  const int lin = blockIdx.x * blockDim.x + tid;
  bool in_range = lin < n_samples;
  int sidx = lin, t_ith = 0, t_r = 0, t_row = 0;
  if (TABLES) {
    const int nxy = max(cnt[0] * cnt[1], 1);
    const int li = t_li0 + (int)tid;
    t_row = divSmall(li, nxy);   // row within the group = row of the tables in LDS
    t_r = li - t_row * nxy;      // index of the (vx, vy) pair, x-outer
    t_ith = t_row_base + t_row;
    in_range = n_samples > 0 && t_row < t_rows;
    sidx = t_r * cnt[2] + t_ith;
  }
  double total = -1.0;
  int status = NAVGPU_SAMPLE_REJECTED;

  if (in_range) {
    float vs[3];
    if (EXPLICIT) {
      vs[0] = explicit_sample[0];
      vs[1] = explicit_sample[1];
      vs[2] = explicit_sample[2];
    } else {
      const int nth = cnt[2], nyv = cnt[1];
      int ix, iy, ith;
      if (TABLES) {  // sidx = (ix * nyv + iy) * nth + ith with ith = t_ith: one division instead of two
        ix = divSmall(t_r, nyv);
        iy = t_r - ix * nyv;
        ith = t_ith;
      } else {
        ix = sidx / (nyv * nth);
        const int rem = sidx - ix * (nyv * nth);
        iy = rem / nth;
        ith = rem - iy * nth;
      }
      vs[0] = s_axis[0][ix];
      vs[1] = s_axis[1][iy];
      vs[2] = s_axis[2][ith];
    }
    // ---- generateTrajectory: reject tests and step count (:193-216)
    const double vmag = hyp2((double)vs[0], (double)vs[1]);
    const double eps = 1e-4;
    bool reject = false;
    if ((c.min_trans_vel >= 0 && vmag + eps < c.min_trans_vel) && (c.min_rot_vel >= 0 && fabs((double)vs[2]) + eps < c.min_rot_vel)) reject = true;
    if (c.max_trans_vel >= 0 && vmag - eps > c.max_trans_vel) reject = true;
    int num_steps = 0;
    if (!reject) {
      double ns;
      if (TABLES) {
        ns = (double)K;  // = ceil(sim_time / sim_granularity), evaluated once on the host (the tables exist for discretize_by_time only)
      } else if (c.discretize_by_time) {
        ns = ceil(c.sim_time / c.sim_granularity);
      } else {
        double sim_time_distance = vmag * c.sim_time;
        double sim_time_angle = fabs((double)vs[2]) * c.sim_time;
        ns = ceil(fmax(sim_time_distance / c.sim_granularity, sim_time_angle / c.angular_sim_granularity));
      }
      num_steps = (int)ns;
      if (num_steps <= 0) reject = true;  // `return num_steps > 0` (:250)
      if (num_steps > (int)pl.max_sim_steps) num_steps = (int)pl.max_sim_steps;  // host validates the capacity
    }
    // DWAPlanner::checkTrajectory ignores generateTrajectory's return value and scores whatever
    // points exist (dwa_planner.cpp:229-230): a rejected sample is an empty trajectory, cost 0.
    if (EXPLICIT && reject) {
      reject = false;
      num_steps = 0;
    }
    if (!reject) {
      status = NAVGPU_SAMPLE_SCORED;
      const double dt = TABLES ? pl.tab_dt : c.sim_time / num_steps;  // (tab_dt = sim_time / tab_steps, the same division, once on the host)
      const bool continued = TABLES ? false : !c.use_dwa;  // (the tables exist for use_dwa only)
      float px = st.pos[0], py = st.pos[1], pth = st.pos[2];
      float lv[3] = {vs[0], vs[1], vs[2]};
      const float acc[3] = {(float)c.acc_lim_x, (float)c.acc_lim_y, (float)c.acc_lim_theta};
      auto newVel = [&](const float* vel_in, float* out) {  // computeNewVelocities (:265-276)
        for (int i = 0; i < 3; ++i) {
          if (vel_in[i] < vs[i])
            out[i] = (float)fmin((double)vs[i], vel_in[i] + acc[i] * dt);
          else
            out[i] = (float)fmax((double)vs[i], vel_in[i] - acc[i] * dt);
        }
      };
      if (continued) {
        float t0[3];
        newVel(st.vel, t0);
        lv[0] = t0[0];
        lv[1] = t0[1];
        lv[2] = t0[2];
      }
      const double xv = lv[0], yv = lv[1], thv = lv[2];  // traj.xv_, yv_, thetav_

      // ---- critics
      const uint32_t osc = EXPLICIT ? 0u : pl.osc_flags[inst];
      const bool osc_fail = ((osc & NAVGPU_OSC_FORWARD_POS_ONLY) && xv < 0.0) || ((osc & NAVGPU_OSC_FORWARD_NEG_ONLY) && xv > 0.0) ||
                            ((osc & NAVGPU_OSC_STRAFE_POS_ONLY) && yv < 0.0) || ((osc & NAVGPU_OSC_STRAFE_NEG_ONLY) && yv > 0.0) ||
                            ((osc & NAVGPU_OSC_ROT_POS_ONLY) && thv < 0.0) || ((osc & NAVGPU_OSC_ROT_NEG_ONLY) && thv > 0.0);
      const double sc_obs = pl.scale_obstacle, sc_gf = pl.scale_goal, sc_al = pl.align_on[inst] ? pl.scale_path : 0.0,
                   sc_path = pl.scale_path, sc_goal = pl.scale_goal;
      const bool en_obs = sc_obs != 0, en_gf = sc_gf != 0, en_al = sc_al != 0, en_path = sc_path != 0, en_goal = sc_goal != 0;
      double fail_code = 0;  // code of critic `first_fail`: the only one scoreTrajectory's in-order sum can return
      double v_obs = 0, v_gf = 0, v_al = 0, v_path = 0, v_goal = 0;
      if constexpr (AGG) {  // `if (aggregationType_ == Product) cost = 1.0` (:77-79)
        if (pl.mg_agg[0] == 2) v_path = 1.0;
        if (pl.mg_agg[1] == 2) v_goal = 1.0;
        if (pl.mg_agg[2] == 2) v_gf = 1.0;
        if (pl.mg_agg[3] == 2) v_al = 1.0;
      }
      int first_fail = 6;  // order index of the earliest critic that failed (1..5), 6 = none
      const bool allow_unknown = c.allow_unknown != 0;
      const double fpd = c.forward_point_distance;
      const uint32_t N_obst = pl.cells, N_unreach = pl.cells + 1;

      if (en_obs && nfp == 0) {  // "Footprint spec is empty" (obstacle_cost_function.cpp:78-82)
        fail_code = -9.0;
        first_fail = 1;
      }
      // a critic is live while no critic before it in the order has failed; the lowest enabled order decides when
      // nothing is left to evaluate
      const int min_order = en_obs ? 1 : en_gf ? 2 : en_al ? 3 : en_path ? 4 : en_goal ? 5 : 6;
      // the forward point (x + fpd cos, y + fpd sin) stays on the map whenever the centre cell is this many cells
      // away from every border; only then may a step skip its worldToMap
      const uint32_t fwd_margin = (uint32_t)fmin(ceil(fabs(fpd) * inv_res) + 1.0, 1.0e6);
      const bool fwd_screen = !(en_gf || en_al) || (2u * fwd_margin < g.nx && 2u * fwd_margin < g.ny);
      const uint32_t fwd_lo = (en_gf || en_al) ? fwd_margin : 0u, fwd_nx = g.nx - 2u * fwd_lo, fwd_ny = g.ny - 2u * fwd_lo;
      // (the screens' LDS offset in a vector register: as a scalar it is spilled and read back with v_readlane at every point)
      uint32_t fb_off = (uint32_t)win_bytes;
      asm volatile("" : "+v"(fb_off));
      const uint4* s_fb4 = reinterpret_cast<const uint4*>(s_dyn + fb_off);
      // which of the four screens count: obstacle (dilated "not free" with sum_scores, else dilated "can fail"), path, goal
      const bool scr_sum = c.sum_scores != 0;  // the obstacle screen: dilated "not free" with sum_scores, else dilated "can fail"
      const bool screen_on = !AGG && fwd_screen && (nfp >= 3 || !en_obs);
      // the forward-margin test is only needed when the LDS window reaches into the margin band of the map (wave-uniform)
      const bool need_margin = !((uint32_t)wx0 - fwd_lo < fwd_nx && (uint32_t)(wx0 + win - 1) - fwd_lo < fwd_nx && (uint32_t)wy0 - fwd_lo < fwd_ny &&
                                 (uint32_t)(wy0 + win - 1) - fwd_lo < fwd_ny);
      uint32_t scr_z = 0xFFFFFFFFu, scr_w = 0xFFFFFFFFu;  // the path / goal screens count while their critics are live
      if (osc_fail) {
        total = -5.0;
      } else {
        for (int step = 0; step < num_steps; ++step) {
          if (first_fail <= min_order) break;
          const int te = TABLES ? t_row * K + step : 0;
          if (TABLES) pth = s_th[te];
          const double x = px, y = py, th = pth;
          double sn, cs;
          if (TABLES) {
            cs = s_trig[4 * te];
            sn = s_trig[4 * te + 1];
          } else {
            sincos(th, &sn, &cs);
          }
          uint32_t cx = 0, cy = 0;
          const bool ok_c = w2m(x, y, cx, cy);
          // ---- screen: on every point but the last a critic can only FAIL (its value is overwritten: aggregation
          // Last; with sum_scores the obstacle critic adds the point's cost, which is 0 when everything in reach is
          // free).  One 16-byte LDS read says whether any critic could fail here; if none can, the point is done.
          // Branch-free up to the decision: the screen word is read whatever the point is (clamped address) and the NEXT
          // pose is computed while that read is in flight - the step's LDS round trips used to be waited for one by one,
          // each behind its own exec-mask branch (38 % of the kernel's wave cycles were spent parked, profiles/round3_b).
          const bool in_w = ok_c && inWin((int)cx, (int)cy);
          const int lxw = in_w ? (int)cx - wx0 : 0;
          const uint4 fbw = s_fb4[(in_w ? (int)cy - wy0 : 0) * nw + (lxw >> 5)];
          // ---- advance (computeNewPositions :253-260): fp64 on fp32 state, rounded back to fp32
          if (continued) {
            float t1[3];
            newVel(lv, t1);
            lv[0] = t1[0];
            lv[1] = t1[1];
            lv[2] = t1[2];
          }
          double sn2 = 0.0, cs2 = 0.0;
          if (TABLES) {
            cs2 = s_trig[4 * te + 2];
            sn2 = s_trig[4 * te + 3];
          } else if (lv[1] != 0.0f) {
            sincos(M_PI_2 + th, &sn2, &cs2);
          }
          // (rollout_trig 1: cos(pos[2]) names the float function and vel[0] * cos(pos[2]) is a float product - navgpu.h; its value is
          // taken as the double function's, rounded to float)
          const double tx = c.rollout_trig ? (double)(lv[0] * (float)cs) : lv[0] * cs, ty = c.rollout_trig ? (double)(lv[0] * (float)sn) : lv[0] * sn;
          const float nxp = (float)(px + (tx + lv[1] * cs2) * dt);
          const float nyp = (float)(py + (ty + lv[1] * sn2) * dt);
          const float ntp = (float)(pth + lv[2] * dt);
          // (a critic that has already failed, or that follows one that has, cannot change the outcome any more: scr_z / scr_w)
          const uint32_t any = (scr_sum ? fbw.x : fbw.y) | (fbw.z & scr_z) | (fbw.w & scr_w);
          const bool margin_ok = !need_margin || ((cx - fwd_lo < fwd_nx) && (cy - fwd_lo < fwd_ny));
          const bool screened = screen_on && step != num_steps - 1 && in_w && !((any >> (lxw & 31)) & 1u) && margin_ok;
#ifdef NAVGPU_SCORE_STATS
          if (screen_on && step != num_steps - 1 && in_w) {
            SCORE_STAT(8, (fbw.y >> (lxw & 31)) & 1u);
            SCORE_STAT(9, (fbw.z >> (lxw & 31)) & 1u);
            SCORE_STAT(10, (fbw.w >> (lxw & 31)) & 1u);
            SCORE_STAT(11, !((cx - fwd_lo < fwd_nx) && (cy - fwd_lo < fwd_ny)));
          } else if (step != num_steps - 1) {
            SCORE_STAT(12, !ok_c);
            SCORE_STAT(13, ok_c && !inWin((int)cx, (int)cy));
            SCORE_STAT(14, !screen_on);
          }
#endif
#ifdef NAVGPU_SCORE_STATS
          {
            const unsigned long long act = __ballot(true), uns = __ballot(!screened);
            if (__ffsll((long long)act) - 1 == (int)(tid & 63)) {
              SCORE_STAT(0, __popcll(act));
              SCORE_STAT(1, __popcll(uns));
              SCORE_STAT(2, 1);
              SCORE_STAT(3, uns != 0);
              SCORE_STAT(6, step == num_steps - 1);
            }
          }
#endif
          if (!screened) {
          const bool live_obs = en_obs && 1 < first_fail;
          // all_free: every cell the footprint can touch is FREE_SPACE -> the step costs exactly 0.
          // Without sum_scores only the LAST point's footprint cost survives (obstacle_cost_function.cpp:
          // cost = f_cost), the earlier points only have to be legal: no failing cell in reach is enough.
          bool all_free = false;
          if (live_obs && nfp >= 3 && in_w) {  // (the same screen word as above)
            const bool not_free = (fbw.x >> (lxw & 31)) & 1u, can_fail = (fbw.y >> (lxw & 31)) & 1u;
            all_free = !not_free || (!c.sum_scores && step != num_steps - 1 && !can_fail);
          }
          if (live_obs && all_free) {
            v_obs = c.sum_scores ? v_obs + 0.0 : 0.0;
          } else if (live_obs) {
#ifdef NAVGPU_SCORE_STATS
            {
              const unsigned long long wk = __ballot(ok_c && nfp >= 3);
              if (__ffsll((long long)__ballot(true)) - 1 == (int)(tid & 63)) {
                SCORE_STAT(4, __popcll(wk));
                SCORE_STAT(5, wk != 0);
              }
            }
#endif
            double f_cost = 0.0;
            bool bad = !ok_c;  // CostmapModel::footprintCost: centre off the map -> -1
            if (!bad) {
              if (nfp < 3) {
                uint8_t cc = cellCost(cx, cy);
                if (cc == kLethal || cc == kInscribed || (cc == kNoInfo && !allow_unknown))
                  bad = true;
                else
                  f_cost = cc;
              } else {
                int fx0 = 0, fy0 = 0, pxc = 0, pyc = 0;
                uint32_t mx_cost = 0;  // maximum over the perimeter cells, in walk order
                for (uint32_t v = 0; v <= nfp && !bad; ++v) {
                  int vx, vy;
                  if (v < nfp) {
                    double wx, wy;
                    if (TABLES) {
                      wx = x + s_rot[(te * tnfp + v) * 2];
                      wy = y + s_rot[(te * tnfp + v) * 2 + 1];
                    } else {
                      const double sx = s_fp[2 * v], sy = s_fp[2 * v + 1];
                      wx = x + (sx * cs - sy * sn);
                      wy = y + (sx * sn + sy * cs);
                    }
                    uint32_t ux, uy;
                    if (!w2m(wx, wy, ux, uy)) {
                      bad = true;
                      break;
                    }
                    vx = (int)ux;
                    vy = (int)uy;
                    if (v == 0) {
                      fx0 = vx;
                      fy0 = vy;
                      pxc = vx;
                      pyc = vy;
                      continue;
                    }
                  } else {  // closing edge: last -> first
                    vx = fx0;
                    vy = fy0;
                  }
                  // lineCost over LineIterator(pxc, pyc, vx, vy)
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
                  if (__builtin_expect(inWin(pxc, pyc) && inWin(vx, vy), 1)) {
                    // every cell of the line lies in the endpoints' bounding box, hence in the window.
                    // The Bresenham addresses do not depend on the bytes read, so the cells are fetched
                    // kChunk at a time with all ds_reads in flight together (one wait per chunk instead
                    // of one dependent LDS round trip per cell); cells past the end re-read the first
                    // cell, cells past a lethal cell cannot change the outcome (-1 either way).
                    constexpr int kChunk = CHUNK;
                    // (the LDS address itself is stepped: with an index, the window's base is added again for every cell)
                    const uint8_t* pw = s_win + ((pyc - wy0) * win + (pxc - wx0));
                    const uint8_t* const pw_first = pw;
                    const int inc1 = yinc1 * win + xinc1, inc2 = yinc2 * win + xinc2;
                    for (int cp = 0; cp <= numpixels && !bad; cp += kChunk) {
                      uint32_t cellv[kChunk];
#pragma unroll
                      for (int u = 0; u < kChunk; ++u) {
                        cellv[u] = *((cp + u <= numpixels) ? pw : pw_first);
                        num += numadd;
                        if (num >= den) {
                          num -= den;
                          pw += inc1;
                        }
                        pw += inc2;
                      }
#pragma unroll
                      for (int u = 0; u < kChunk; ++u) mx_cost = max(mx_cost, cellv[u]);
                      bad = mx_cost >= walk_fail;
                    }
                  } else {
                    for (int cp = 0; cp <= numpixels; ++cp) {
                      const uint8_t cc = master[ly * g.nx + lx];
                      if ((uint8_t)(cc - kLethal) <= fail_span) {
                        bad = true;
                        break;
                      }
                      const uint32_t ct = (walk_swap && cc >= 254) ? (cc ^ 1u) : cc;  // walk order, like the LDS bytes
                      mx_cost = ct > mx_cost ? ct : mx_cost;
                      num += numadd;
                      if (num >= den) {
                        num -= den;
                        lx += xinc1;
                        ly += yinc1;
                      }
                      lx += xinc2;
                      ly += yinc2;
                    }
                  }
                  pxc = vx;
                  pyc = vy;
                }
                f_cost = (walk_swap && mx_cost == 254u) ? 255.0 : (double)mx_cost;  // an allowed NO_INFORMATION cell costs 255
              }
            }
            if (bad) {
              fail_code = -6.0;
              first_fail = 1;
            } else {
              // ok_c holds here, so the -7 branch (obstacle_cost_function.cpp:135-137) cannot fire
              const double occ = fmax(fmax(0.0, f_cost), (double)cellCost(cx, cy));
              v_obs = c.sum_scores ? v_obs + occ : occ;
            }
          }
          if constexpr (AGG) {
            // the general MapGridCostFunction step, critic by critic in the order DWAPlanner lists them
            auto critic = [&](bool en, int order, const uint32_t* grid, double xs, double ys, bool stop_on_failure, int agg, double& v) {
              if (!(en && order < first_fail)) return;
              double sx = x, sy = y;
              if (xs != 0.0) {
                sx = sx + xs * cs;
                sy = sy + xs * sn;
              }
              if (ys != 0.0) {
                double s2, c2;
                sincos(th + M_PI_2, &s2, &c2);
                sx = sx + ys * c2;
                sy = sy + ys * s2;
              }
              uint32_t ux, uy;
              if (!w2m(sx, sy, ux, uy)) {
                fail_code = -4.0;
                first_fail = order;
                return;
              }
              const uint32_t d = grid[uy * g.nx + ux];
              if (stop_on_failure && (d == N_obst || d == N_unreach)) {
                fail_code = d == N_obst ? -3.0 : -2.0;
                first_fail = order;
                return;
              }
              const double gd = (double)d;
              if (agg == 0)
                v = gd;
              else if (agg == 1)
                v += gd;
              else if (v > 0)
                v *= gd;
            };
            critic(en_gf, 2, dfront, fpd, pl.mg_yshift[2], false, pl.mg_agg[2], v_gf);
            critic(en_al, 3, dpath, fpd, pl.mg_yshift[3], false, pl.mg_agg[3], v_al);
            critic(en_path, 4, dpath, 0.0, pl.mg_yshift[0], true, pl.mg_agg[0], v_path);
            critic(en_goal, 5, dgoal, 0.0, pl.mg_yshift[1], true, pl.mg_agg[1], v_goal);
          } else {
          if ((en_path && 4 < first_fail) || (en_goal && 5 < first_fail)) {
            if (!ok_c) {
              if (en_path && 4 < first_fail) {
                fail_code = -4.0;
                first_fail = 4;
              } else {
                fail_code = -4.0;
                first_fail = 5;
              }
            } else {
              const uint32_t cell = cy * g.nx + cx;
              // A point of the window whose path / goal screen bit is clear cannot fail that critic (the bit IS the failure
              // test, taken from this cycle's grid by the prep launch), and of the distances only the LAST point's survives
              // (aggregation Last): no look-up.  Points that leave the screened path because an obstacle is near - most of
              // them - used to wait for two L2 round trips here.
              const bool last_pt = step == num_steps - 1;
              const bool look_p = last_pt || !in_w || ((fbw.z >> (lxw & 31)) & 1u);
              const bool look_g = last_pt || !in_w || ((fbw.w >> (lxw & 31)) & 1u);
              if (en_path && 4 < first_fail && look_p) {
                const uint32_t d = dpath[cell];
                if (d == N_obst) {
                  fail_code = -3.0;
                  first_fail = 4;
                } else if (d == N_unreach) {
                  fail_code = -2.0;
                  first_fail = 4;
                } else
                  v_path = d;
              }
              if (en_goal && 5 < first_fail && look_g) {
                const uint32_t d = dgoal[cell];
                if (d == N_obst) {
                  fail_code = -3.0;
                  first_fail = 5;
                } else if (d == N_unreach) {
                  fail_code = -2.0;
                  first_fail = 5;
                } else
                  v_goal = d;
              }
            }
          }
          if ((en_gf && 2 < first_fail) || (en_al && 3 < first_fail)) {
            double sx = x, sy = y;
            if (fpd != 0.0) {
              sx = x + fpd * cs;
              sy = y + fpd * sn;
            }
            uint32_t ux, uy;
            if (!w2m(sx, sy, ux, uy)) {
              if (en_gf && 2 < first_fail) {
                fail_code = -4.0;
                first_fail = 2;
              } else {
                fail_code = -4.0;
                first_fail = 3;
              }
            } else if (step == num_steps - 1) {  // aggregation Last: only the final point's value survives
              const uint32_t cell = uy * g.nx + ux;
              if (en_gf && 2 < first_fail) v_gf = dfront[cell];
              if (en_al && 3 < first_fail) v_al = dpath[cell];
            }
          }
          }  // !AGG
          scr_z = first_fail > 4 ? 0xFFFFFFFFu : 0u;  // (first_fail only changes in here)
          scr_w = first_fail > 5 ? 0xFFFFFFFFu : 0u;
          }  // !screened
          px = nxp;
          py = nyp;
          pth = ntp;
        }
        // ---- scoreTrajectory sum in critic order
        // (the first failing critic in that order ends the sum with its code: simple_scored_sampling_planner.cpp:59-66)
        total = 0.0;
        if (first_fail < 6) {
          total = fail_code;
        } else {
          auto add = [&](bool en, double value, double scale) {
            if (!en) return;
            double cost = value;
            if (cost != 0) cost *= scale;
            total += cost;
          };
          add(en_obs, v_obs, sc_obs);
          add(en_gf, v_gf, sc_gf);
          add(en_al, v_al, sc_al);
          add(en_path, v_path, sc_path);
          add(en_goal, v_goal, sc_goal);
        }
      }
    }
    if (!EXPLICIT && pl.sample_cost) {
      pl.sample_cost[(size_t)inst * pl.max_samples + sidx] = total;
      pl.sample_status[(size_t)inst * pl.max_samples + sidx] = status;
    }
  }

#ifdef NAVGPU_SCORE_TIMING
  const unsigned long long ts2 = wall_clock64();
#endif
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
#ifdef NAVGPU_SCORE_TIMING
  // (one workgroup in 256 reports: with every wave adding to the same few words the atomics themselves stretched the launch sixfold
  // and made the prologue look like 39 % of a workgroup's residence; sampled, it is 6 %)
  if (PREP == 2 && (tid & 63) == 0 && (blockIdx.x & 15) == 3 && (blockIdx.y & 15) == 5) {
    const unsigned long long ts3 = wall_clock64();
    atomicAdd(&g_score_stats[16], ts1 - ts0);  // image load, per wave
    atomicAdd(&g_score_stats[22], ts0a - ts0);  // ... of which: staging of footprint / axis samples up to the first barrier
    atomicAdd(&g_score_stats[23], ts0b - ts0a); // ... lane mapping up to the second barrier
    atomicAdd(&g_score_stats[17], ts2 - ts1);  // sample setup + rollout, per wave
    atomicAdd(&g_score_stats[18], ts3 - ts2);  // reduction + wait for the slowest wave of the workgroup
    atomicAdd(&g_score_stats[19], 1ull);
    if (tid == 0) atomicAdd(&g_score_stats[20], ts3 - ts0);  // workgroup residence
    if (tid == 0) atomicAdd(&g_score_stats[21], 1ull);
  }
#endif
  if (tid == 0) {
    for (int w = 1; w < THREADS / 64; ++w)
      if (s_rc[w] < bc || (s_rc[w] == bc && s_ri[w] < bi)) {
        bc = s_rc[w];
        bi = s_ri[w];
      }
    pl.part_cost[(size_t)inst * pl.score_blocks + blockIdx.x] = bc;  // score_blocks = capacity (256-thread blocks)
    pl.part_index[(size_t)inst * pl.score_blocks + blockIdx.x] = bi;
    if (s_cnt[0]) atomicAdd(&pl.counters[2 * inst], s_cnt[0]);
    if (s_cnt[1]) atomicAdd(&pl.counters[2 * inst + 1], s_cnt[1]);
  }
}

// three entry points over the same body: the table variant is compiled for 6 waves/SIMD (80 VGPRs) in 256-thread
// workgroups whose image (window + screens + the v_theta rows of their row group) stays below 26 KB: 6 per CU = 24 waves
constexpr int kScoreThreadsTab = NAVGPU_SCORE_TAB_THREADS;
constexpr int kScorePrepThreads = NAVGPU_SCORE_PREP_THREADS;  // the workgroup that builds a robot's image
template <int CHUNK>
__global__ __launch_bounds__(kScoreThreadsTab, NAVGPU_SCORE_TAB_WAVES) void k_score_tab(PlannerDev pl, uint32_t first, const float* explicit_sample) {
  score_body<false, true, kScoreThreadsTab, 2, CHUNK>(pl, first, explicit_sample);
}
template <int CHUNK>
__global__ __launch_bounds__(kScoreThreads) void k_score_gen(PlannerDev pl, uint32_t first, const float* explicit_sample) {
  score_body<false, false, kScoreThreads, 2, CHUNK>(pl, first, explicit_sample);
}
__global__ __launch_bounds__(kScorePrepThreads) void k_score_prep_tab(PlannerDev pl, uint32_t first) {
  score_body<false, true, kScorePrepThreads, 1>(pl, first, nullptr);
}
__global__ __launch_bounds__(kScoreThreads) void k_score_prep_gen(PlannerDev pl, uint32_t first) {
  score_body<false, false, kScoreThreads, 1>(pl, first, nullptr);
}
__global__ __launch_bounds__(kScoreThreads) void k_score_explicit(PlannerDev pl, uint32_t first, const float* explicit_sample) {
  score_body<true, false, kScoreThreads>(pl, first, explicit_sample);
}
// the same two entry points with the general MapGridCostFunction step (aggregation Sum / Product, sideways shift)
__global__ __launch_bounds__(kScoreThreads) void k_score_gen_agg(PlannerDev pl, uint32_t first, const float* explicit_sample) {
  score_body<false, false, kScoreThreads, 2, 16, true>(pl, first, explicit_sample);
}
__global__ __launch_bounds__(kScoreThreads) void k_score_explicit_agg(PlannerDev pl, uint32_t first, const float* explicit_sample) {
  score_body<true, false, kScoreThreads, 0, 12, true>(pl, first, explicit_sample);
}

size_t score_window_bytes(uint32_t win) {  // costmap window + the four per-cell screens
  return (((size_t)win * win + 15) & ~(size_t)15) + score_bits_bytes((int)win);
}
size_t score_table_row_bytes(const PlannerDev& pl) { return (size_t)pl.tab_steps * ((4 + 2 * pl.tab_nfp) * sizeof(double) + sizeof(float)); }
size_t score_table_bytes(const PlannerDev& pl) {  // all v_theta rows: the image k_score_prep_tab builds
  return ((size_t)pl.tab_nth * score_table_row_bytes(pl) + 15) & ~(size_t)15;
}
size_t score_table_lds_bytes(const PlannerDev& pl) {  // the tab_rows rows of one row group: what a k_score_tab workgroup holds
  return ((size_t)pl.tab_rows * score_table_row_bytes(pl) + 15) & ~(size_t)15;
}
// v_theta rows per row group and the LDS budget they were sized for; 0 = no tables.  A workgroup's image (window +
// screens + rows) should leave room for three workgroups per CU (52 KB each), else two (78 KB); the image of ALL
// rows has to fit the one workgroup that builds it.
uint32_t score_table_rows(const PlannerDev& pl, uint32_t win) {
  const size_t wb = score_window_bytes(win), row = score_table_row_bytes(pl);
  if (row == 0 || wb + score_table_bytes(pl) + score_scratch_bytes((int)win) > 150u * 1024u) return 0;
  for (size_t budget : {(size_t)NAVGPU_SCORE_TAB_LDS_KB * 1024, (size_t)52 * 1024, (size_t)78 * 1024}) {
    if (wb + 16 >= budget) continue;
    const size_t r = (budget - wb - 16) / row;
    if (r >= 1) return (uint32_t)std::min<size_t>(r, pl.tab_nth);
  }
  return 0;
}
size_t score_prep_bytes(const PlannerDev& pl) {  // the LDS image k_score_prep* stores per robot
  return score_window_bytes(pl.win) + (pl.use_tables ? score_table_bytes(pl) : 0);
}
size_t score_prep_slot_bytes(const PlannerDev& pl) {  // a robot's slot of pl.prep: the image, the sweep launch's scalars, the pairs' reject bytes
  return ((score_prep_bytes(pl) + 255) & ~(size_t)255) + kScoreAuxBytes + score_prep_reject_bytes(pl);
}
uint32_t launch_score(const PlannerDev& pl_in, uint32_t first, uint32_t count, const float* explicit_sample, hipStream_t s) {
  PlannerDev pl = pl_in;
  // A/B switch for the per-(v_theta, step) tables (tools/probe_score.py)
  if (NAVGPU_DEBUG_ENV("NAVGPU_DEBUG_NO_TABLES") && atoi(NAVGPU_DEBUG_ENV("NAVGPU_DEBUG_NO_TABLES"))) pl.use_tables = 0;  // tool builds only
  const size_t win_bytes = score_window_bytes(pl.win);
  const size_t scratch = score_scratch_bytes((int)pl.win);  // only where the image is built
  pl.tab_bytes = 0;
  if (explicit_sample) {
    const size_t lds_x = win_bytes + scratch;
    if (pl.mg_generic) {
      if (lds_x > 48 * 1024) hipFuncSetAttribute((const void*)k_score_explicit_agg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_x);
      hipLaunchKernelGGL(k_score_explicit_agg, dim3(1, count), dim3(kScoreThreads), lds_x, s, pl, first, explicit_sample);
      return 1;
    }
    if (lds_x > 48 * 1024) hipFuncSetAttribute((const void*)k_score_explicit, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_x);
    hipLaunchKernelGGL(k_score_explicit, dim3(1, count), dim3(kScoreThreads), lds_x, s, pl, first, explicit_sample);
    return 1;
  }
  if (pl.mg_generic) pl.use_tables = 0;  // (the general step has no table variant)
  pl.prep_bytes = (uint32_t)score_prep_bytes(pl);  // (after the debug override of use_tables)
  if (pl.use_tables) {
    // the prep launch builds all rows (its LDS holds the whole image); a scoring workgroup holds one row group
    pl.tab_bytes = (uint32_t)score_table_bytes(pl);
    const size_t lds_prep = win_bytes + score_table_bytes(pl) + scratch;
    if (lds_prep > 48 * 1024) hipFuncSetAttribute((const void*)k_score_prep_tab, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_prep);
    hipLaunchKernelGGL(k_score_prep_tab, dim3(1, count), dim3(kScorePrepThreads), lds_prep, s, pl, first);
    pl.tab_bytes = (uint32_t)score_table_lds_bytes(pl);
    if (score_sweep_applies(pl) && !NAVGPU_DEBUG_ENV("NAVGPU_DEBUG_NO_SWEEP")) return launch_score_sweep(pl, first, count, s);
    const size_t lds = win_bytes + score_table_lds_bytes(pl);
    // row groups x workgroups per group, for the largest (vx, vy) grid the configuration can produce
    const uint32_t max_nxy = pl.max_samples / std::max(pl.tab_nth, 1u), groups = (pl.tab_nth + pl.tab_rows - 1) / pl.tab_rows;
    const uint32_t blocks = std::min(groups * ((max_nxy * pl.tab_rows + kScoreThreadsTab - 1) / kScoreThreadsTab), pl.score_blocks);
#define NAVGPU_SCORE_TAB(C)                                                                                              \
  {                                                                                                                      \
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k_score_tab<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(k_score_tab<C>, dim3(blocks, count), dim3(kScoreThreadsTab), lds, s, pl, first, explicit_sample);  \
  }
    if (pl.fp_chunk <= 6) NAVGPU_SCORE_TAB(6)
    else if (pl.fp_chunk <= 9) NAVGPU_SCORE_TAB(9)
    else if (pl.fp_chunk <= 12) NAVGPU_SCORE_TAB(12)
    else NAVGPU_SCORE_TAB(16)
#undef NAVGPU_SCORE_TAB
    return blocks;
  }
  const uint32_t gen_blocks = (pl.max_samples + kScoreThreads - 1) / kScoreThreads;  // (score_blocks is the capacity of the partial results)
  if (win_bytes + scratch > 48 * 1024) hipFuncSetAttribute((const void*)k_score_prep_gen, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(win_bytes + scratch));
  hipLaunchKernelGGL(k_score_prep_gen, dim3(1, count), dim3(kScoreThreads), win_bytes + scratch, s, pl, first);
#define NAVGPU_SCORE_GEN(C)                                                                                                    \
  {                                                                                                                            \
    if (win_bytes > 48 * 1024) hipFuncSetAttribute((const void*)k_score_gen<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)win_bytes); \
    hipLaunchKernelGGL(k_score_gen<C>, dim3(gen_blocks, count), dim3(kScoreThreads), win_bytes, s, pl, first, explicit_sample);          \
  }
  if (pl.mg_generic) {
    if (win_bytes > 48 * 1024) hipFuncSetAttribute((const void*)k_score_gen_agg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)win_bytes);
    hipLaunchKernelGGL(k_score_gen_agg, dim3(gen_blocks, count), dim3(kScoreThreads), win_bytes, s, pl, first, explicit_sample);
    return gen_blocks;
  }
  if (pl.fp_chunk <= 6) NAVGPU_SCORE_GEN(6)
  else if (pl.fp_chunk <= 9) NAVGPU_SCORE_GEN(9)
  else if (pl.fp_chunk <= 12) NAVGPU_SCORE_GEN(12)
  else NAVGPU_SCORE_GEN(16)
#undef NAVGPU_SCORE_GEN
  return gen_blocks;
}

#ifdef NAVGPU_PREP_TIMING
extern "C" int navgpu_debug_prep_stats(unsigned long long* out16, int reset) {
  if (out16) hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_prep_stats), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_prep_stats), z, sizeof(z));
  }
  return 0;
}
#endif
#if defined(NAVGPU_SCORE_STATS) || defined(NAVGPU_SCORE_TIMING)
extern "C" int navgpu_debug_score_stats(unsigned long long* out8, int reset) {
  if (out8) hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_score_stats), sizeof(unsigned long long) * 24);
  if (reset) {
    unsigned long long z[24] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_score_stats), z, sizeof(z));
  }
  return 0;
}
#endif

}  // namespace navgpu
