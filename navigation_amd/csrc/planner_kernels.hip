// HIP kernels (gfx950) for the local-planner half of the hot path:
//   k_samples : VelocityIterator + SimpleTrajectoryGenerator::initialise (per-axis sample lists)
//   k_bfs     : MapGridCostFunction::prepare — MapGrid wavefronts as bit-parallel level-synchronous BFS
//   k_score   : generateTrajectory + the six DWA critics, one lane per velocity sample
//   k_select  : first-strict-minimum selection, winner trajectory, oscillation flag update
// Compiled with -ffp-contract=off: the fp64 step arithmetic on fp32 state has to round exactly
// like the reference (simple_trajectory_generator.cpp:253-260, SURVEY §7 hard part 2).
#include <cstdlib>

#include "navgpu_device.h"

namespace navgpu {

constexpr int kMaxAxis = 128;  // per-axis sample capacity staged in LDS (vsamples + 1 <= 128)

__device__ __forceinline__ Geom geomOf(const PlannerDev& pl, uint32_t inst) {
  return Geom{pl.origin[2 * inst], pl.origin[2 * inst + 1], pl.res, pl.nx, pl.ny};
}

// ------------------------------------------------------------------------------------------------
// k_samples: base_local_planner/include/base_local_planner/velocity_iterator.h:49-74 and
// SimpleTrajectoryGenerator::initialise (src/simple_trajectory_generator.cpp:60-135).
// One lane per axis: `next += step_size` is a sequential fp64 accumulation and must stay one.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bfsFreeWord(const uint8_t* master, uint32_t row, uint32_t nx, uint32_t wi, uint32_t unknown_is_obstacle);
__device__ __forceinline__ uint32_t bfsObstacleNibble(uint32_t v, uint32_t unknown_is_obstacle);
// Three kinds of 128-thread blocks, all latency-bound and independent of each other, side by side:
//   [0, count)          one robot's wavefront region, pocket floods and care words (wave 0: the region's flood, wave 1: the
//                       large area's), 
//   [count, 2 count)    wave 0: the robot's three items ranked for the longest-first dispatch; wave 1: its velocity samples,
//   behind them         the traversable-cell bitmaps of the launch, 128 words per block.
constexpr int kSamplesThreads = 128;
__global__ __launch_bounds__(kSamplesThreads) void k_samples(PlannerDev pl, uint32_t first, uint32_t count) {
  const uint32_t tid = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  if (blockIdx.x >= 2 * count) {
    // the traversable-cell bitmaps of the launch (what k_free_bits does on its own for the other callers of launch_bfs):
    // throughput work that fills the CUs while the per-robot waves wait on memory
    const uint32_t W = (pl.nx + 31) >> 5, words = pl.ny * W, per = (words + kSamplesThreads - 1) / kSamplesThreads;
    const uint32_t b = blockIdx.x - 2 * count, r = b / per, i = (b - r * per) * kSamplesThreads + threadIdx.x;
    if (i < words) {
      const uint32_t row = i / W, wi = i - row * W;
      pl.bfs_free[(size_t)(first + r) * words + i] = bfsFreeWord(pl.master + (size_t)(first + r) * pl.cells_padded, row, pl.nx, wi, pl.cfg.allow_unknown != 0 ? 0u : 1u);
    }
    return;
  }
  const navgpu_dwa_config& c = pl.cfg;
  if (blockIdx.x >= count) {
    const uint32_t robot = blockIdx.x - count, inst = first + robot;
    if (wv == 0) {
      // dispatch order of this launch's wavefronts (k_bfs_wave takes items off a counter): longest first, predicted by
      // the level count of the robot's previous cycle.  item = g * count + robot, g = 0 goal_front, 1 goal, 2 path;
      // every robot ranks its three items among all of them (count * 3 keys: a dozen loads per lane)
      const uint32_t total = 3 * count;
      uint32_t key[3], before[3] = {0, 0, 0};
#pragma unroll
      for (uint32_t g = 0; g < 3; ++g) key[g] = pl.bfs_levels[(size_t)inst * 3 + (2 - g)];
#pragma unroll 4
      for (uint32_t j = tid; j < total; j += 64) {
        const uint32_t gj = j / count, rj = j - gj * count;
        const uint32_t kj = pl.bfs_levels[(size_t)(first + rj) * 3 + (2 - gj)];
#pragma unroll
        for (uint32_t g = 0; g < 3; ++g) before[g] += (kj > key[g] || (kj == key[g] && j < g * count + robot)) ? 1u : 0u;
      }
#pragma unroll
      for (uint32_t g = 0; g < 3; ++g)
        for (int o = 32; o > 0; o >>= 1) before[g] += __shfl_xor(before[g], o);
      if (tid < 3) pl.bfs_order[(size_t)first * 3 + (tid == 0 ? before[0] : (tid == 1 ? before[1] : before[2]))] = tid * count + robot;
      if (robot == 0 && tid < 2) pl.bfs_next_item[tid] = 0;  // the work counters of the launch_bfs that follows
      return;
    }
    // ---- velocity samples, one lane per axis: `next += step_size` is a sequential fp64 accumulation and must stay one
    const navgpu_robot_state st = pl.state[inst];
    int32_t* cnt = pl.axis_count + 4 * inst;
    int n = 0;
    if (tid < 3) {
      const int a = tid;
      const float vsamp = a == 0 ? (float)c.vx_samples : (a == 1 ? (float)c.vy_samples : (float)c.vth_samples);
      const double max_vel_th = c.max_rot_vel, min_vel_th = -1.0 * max_vel_th;
      double lim_min = a == 0 ? c.min_vel_x : (a == 1 ? c.min_vel_y : min_vel_th);
      double lim_max = a == 0 ? c.max_vel_x : (a == 1 ? c.max_vel_y : max_vel_th);
      const float acc = a == 0 ? (float)c.acc_lim_x : (a == 1 ? (float)c.acc_lim_y : (float)c.acc_lim_theta);
      const float v = st.vel[a];
      float maxv, minv;
      if (!c.use_dwa) {
        // goal = last pose of the plan narrowed to float (dwa_planner.cpp:305-306)
        const double* P = pl.plan + (size_t)inst * pl.max_plan * 2;
        const uint32_t np = pl.plan_count[inst];
        const float gx = (float)P[2 * (np - 1)], gy = (float)P[2 * (np - 1) + 1];
        double dist = hyp2((double)(gx - st.pos[0]), (double)(gy - st.pos[1]));
        if (a < 2) lim_max = fmax(fmin(lim_max, dist / c.sim_time), lim_min);
        maxv = (float)fmin(lim_max, v + acc * c.sim_time);
        minv = (float)fmax(lim_min, v - acc * c.sim_time);
      } else {
        maxv = (float)fmin(lim_max, v + acc * c.sim_period);
        minv = (float)fmax(lim_min, v - acc * c.sim_period);
      }
      float* out = pl.axis_samples + ((size_t)inst * 3 + a) * pl.max_axis;
      const double mn = minv, mx = maxv;
      if (mn == mx) {
        out[n++] = (float)mn;
      } else {
        int num_samples = (int)vsamp;
        num_samples = num_samples > 2 ? num_samples : 2;
        double step_size = (mx - mn) / double(num_samples - 1 > 1 ? num_samples - 1 : 1);
        double current, next = mn;
        for (int j = 0; j < num_samples - 1; ++j) {
          current = next;
          next += step_size;
          if (n < (int)pl.max_axis) out[n] = (float)current;
          ++n;
          if ((current < 0) && (next > 0)) {
            if (n < (int)pl.max_axis) out[n] = 0.0f;
            ++n;
          }
        }
        if (n < (int)pl.max_axis) out[n] = (float)mx;
        ++n;
      }
      n = n < (int)pl.max_axis ? n : (int)pl.max_axis;
      cnt[a] = n;
    }
    const int n0 = __shfl(n, 0), n1 = __shfl(n, 1), n2 = __shfl(n, 2);
    if (tid == 0) {
      float prod = (float)c.vx_samples * (float)c.vy_samples * (float)c.vth_samples;
      cnt[3] = prod > 0 ? n0 * n1 * n2 : 0;
      pl.counters[2 * inst] = 0;
      pl.counters[2 * inst + 1] = 0;
    }
    return;
  }
  const uint32_t inst = first + blockIdx.x;
  const navgpu_robot_state st = pl.state[inst];
  // Bounded wavefronts (k_bfs_wave).  The box = every cell a MapGrid look-up of this robot's samples can fall in: the
  // staged reach around the robot's cell.  The region = the box grown by two cells, clipped to the map.  A search may
  // stop when (a) no cell of the box that it could still reach is open and (b) no frontier cell is inside the region.
  // "Could still reach" leaves out the POCKETS: free cells that no 4-connected chain of free cells joins to the rim of
  // the area looked at (one cell enclosed by inflated obstacles is enough to keep a search going over the whole map
  // otherwise).  A wavefront gets into a pocket only from a seed next to it, which (b) waits for - so the region has
  // to contain every pocket that counts.  Two areas are flooded from their rims, bit-parallel, two rows per lane, whole
  // words filled along a row with an add-carry, neighbour rows by lane shuffles: the region (wave 0), and the largest
  // area the mask can hold around it (128 rows x 4 words; wave 1).  When the large one finds pockets in the box that
  // the region alone does not (a pocket that straddles the region's rim), the large area becomes this robot's region.
  __shared__ uint32_t s_pocket[2][2][kCareWords][64];  // [region | large][row half][word][lane]
  __shared__ int s_ok[2];
  {
    int4 region = make_int4(0, -1, 0, -1);
    int care_ok = 0;
    const uint32_t reach = pl.bfs_reach[inst];
    const Geom g = geomOf(pl, inst);
    uint32_t mx = 0, my = 0;
    if (reach && worldToMap(g, (double)st.pos[0], (double)st.pos[1], mx, my)) {  // (uniform over the block)
      const int R = (int)reach + 2, nxi = (int)pl.nx, nyi = (int)pl.ny, Wm = (nxi + 31) >> 5;
      region.x = max((int)mx - R, 0);
      region.y = min((int)mx + R, nxi - 1);
      region.z = max((int)my - R, 0);
      region.w = min((int)my + R, nyi - 1);
      const int rows = region.w - region.z + 1, wx0 = region.x >> 5, nw = (region.y >> 5) - wx0 + 1;
      if (rows <= kCareRows && nw <= kCareWords) {
        const uint8_t* master = pl.master + (size_t)inst * pl.cells_padded;
        const uint32_t unknown_is_obstacle = pl.cfg.allow_unknown != 0 ? 0u : 1u;
        const int bx0 = max((int)mx - (int)reach, 0), bx1 = min((int)mx + (int)reach, nxi - 1);
        const int by0 = max((int)my - (int)reach, 0), by1 = min((int)my + (int)reach, nyi - 1);
        // the large area: kCareRows rows and kCareWords words around the region (it contains the region)
        const int fy0 = max(min((int)my - kCareRows / 2, nyi - kCareRows), 0), fy1 = min(fy0 + kCareRows - 1, nyi - 1);
        const int fw0 = max(min(wx0 - (kCareWords - nw) / 2, Wm - kCareWords), 0), fw1 = min(fw0 + kCareWords - 1, Wm - 1);
        const int fx0 = fw0 * 32, fx1 = min(fw1 * 32 + 31, nxi - 1);
        auto colMask = [&](int w, int x0, int x1) -> uint32_t {  // bits of word fw0 + w inside [x0, x1]
          const int lo = max(x0 - (fw0 + w) * 32, 0), hi = min(x1 - (fw0 + w) * 32, 31);
          return hi >= lo ? ((0xFFFFFFFFu >> (31 - hi)) & (0xFFFFFFFFu << lo)) : 0u;
        };
        // this wave's area: x0..x1, y0..y1 (wave 0: the region, wave 1: the large area)
        const int ax0 = wv ? fx0 : region.x, ax1 = wv ? fx1 : region.y, ay0 = wv ? fy0 : region.z, ay1 = wv ? fy1 : region.w;
        uint32_t fm[2][kCareWords], F[2][kCareWords];
        // the traversable-cell words of this lane's two rows.  Usual case (rows 16-byte aligned, whole words): all sixteen
        // 16-byte loads are issued before the first is used - bfsFreeWord's general form waits for each word on its own
        uint32_t fword[2][kCareWords];
        if ((pl.nx & 15u) == 0 && (uint32_t)(fw1 * 32 + 31) < pl.nx) {
          uint4 ca[2][kCareWords], cb[2][kCareWords];
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int w = 0; w < kCareWords; ++w) {
              const uint4* p = reinterpret_cast<const uint4*>(master + (size_t)min(fy0 + 2 * (int)tid + h, fy1) * pl.nx + (size_t)min(fw0 + w, fw1) * 32);
              ca[h][w] = p[0];
              cb[h][w] = p[1];
            }
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int w = 0; w < kCareWords; ++w) {
              const uint4 a = ca[h][w], b = cb[h][w];
              fword[h][w] = ~(bfsObstacleNibble(a.x, unknown_is_obstacle) | (bfsObstacleNibble(a.y, unknown_is_obstacle) << 4) |
                              (bfsObstacleNibble(a.z, unknown_is_obstacle) << 8) | (bfsObstacleNibble(a.w, unknown_is_obstacle) << 12) |
                              (bfsObstacleNibble(b.x, unknown_is_obstacle) << 16) | (bfsObstacleNibble(b.y, unknown_is_obstacle) << 20) |
                              (bfsObstacleNibble(b.z, unknown_is_obstacle) << 24) | (bfsObstacleNibble(b.w, unknown_is_obstacle) << 28));
            }
        } else {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int w = 0; w < kCareWords; ++w) {
              const int row = fy0 + 2 * (int)tid + h;
              fword[h][w] = (row <= fy1 && fw0 + w <= fw1) ? bfsFreeWord(master, (uint32_t)row, pl.nx, (uint32_t)(fw0 + w), unknown_is_obstacle) : 0u;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = fy0 + 2 * (int)tid + h;
          const bool in_a = row >= ay0 && row <= ay1;
#pragma unroll
          for (int w = 0; w < kCareWords; ++w) {
            fm[h][w] = F[h][w] = 0;
            if (row <= fy1 && fw0 + w <= fw1) {
              const uint32_t cm = in_a ? colMask(w, ax0, ax1) : 0u;
              const uint32_t fw_ = fword[h][w];
              fm[h][w] = fw_ & cm;
              const uint32_t rim = (row == ay0 || row == ay1) ? cm : (colMask(w, ax0, ax0) | colMask(w, ax1, ax1));
              F[h][w] = fm[h][w] & rim;
            }
          }
        }
        bool ok = false;
        for (int it = 0; it < 256; ++it) {
          uint32_t changed = 0;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int w = 0; w < kCareWords; ++w) {
              uint32_t up, dn;
              if (h == 0) {
                up = __shfl_up(F[1][w], 1);
                if (tid == 0) up = 0;
                dn = F[1][w];
              } else {
                up = F[0][w];
                dn = __shfl_down(F[0][w], 1);
                if (tid == 63) dn = 0;
              }
              const uint32_t cur = F[h][w], f = fm[h][w];
              uint32_t n = cur | up | dn | (cur << 1) | (cur >> 1);
              if (w > 0) n |= F[h][w - 1] >> 31;
              if (w + 1 < kCareWords) n |= F[h][w + 1] << 31;
              n &= f;
              // fill the runs of free cells the set bits lie in: towards bit 31 with an add-carry, towards bit 0 mirrored
              n |= f & ~(f + n);
              const uint32_t fr_ = __brev(f), nr = __brev(n);
              n |= __brev(fr_ & ~(fr_ + nr));
              changed |= n ^ cur;
              F[h][w] = n;
            }
          }
          if (__builtin_amdgcn_ballot_w64(changed != 0) == 0) {
            ok = true;
            break;
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int w = 0; w < kCareWords; ++w) s_pocket[wv][h][w][tid] = fm[h][w] & ~F[h][w];
        if (tid == 0) s_ok[wv] = ok ? 1 : 0;
        __syncthreads();
        const bool okR = s_ok[0] != 0, okL = s_ok[1] != 0;
        care_ok = okR ? 1 : 0;  // not settled within the bound: no pocket is left out (the search is exact either way)
        // pockets of the box that only the large area shows -> the large area is the region
        uint32_t extra = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = fy0 + 2 * (int)tid + h;
#pragma unroll
          for (int w = 0; w < kCareWords; ++w)
            if (row >= by0 && row <= by1) extra |= (s_pocket[1][h][w][tid] ^ s_pocket[0][h][w][tid]) & colMask(w, bx0, bx1);
        }
        const bool large = okR && okL && __builtin_amdgcn_ballot_w64(extra != 0) != 0;
        if (large) region = make_int4(fx0, fx1, fy0, fy1);
        if (okR && wv == 0) {
          const int ry0 = region.z, rw0 = region.x >> 5;
          uint32_t* care = pl.bfs_care + (size_t)inst * kCareRows * kCareWords;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int row = fy0 + 2 * (int)tid + h, rr = row - ry0;
#pragma unroll
            for (int w = 0; w < kCareWords; ++w) {
              const int ww = fw0 + w - rw0;
              if (rr < 0 || rr >= kCareRows || row > region.w || ww < 0 || ww >= kCareWords) continue;
              const uint32_t pocket = s_pocket[large ? 1 : 0][h][w][tid];
              care[rr * kCareWords + ww] = (row >= by0 && row <= by1) ? (colMask(w, bx0, bx1) & ~pocket) : 0u;
            }
          }
        }
      }
    }
    if (threadIdx.x == 0) {
      int* b = pl.bfs_box + (size_t)inst * 8;
      b[0] = region.x;
      b[1] = region.y;
      b[2] = region.z;
      b[3] = region.w;
      b[4] = care_ok;
    }
  }
}
void launch_samples(const PlannerDev& pl, uint32_t first, uint32_t count, hipStream_t s) {
  const uint32_t words = pl.ny * ((pl.nx + 31) / 32);
  // (+ the bitmaps and the zeroed work counters of launch_bfs(..., free_ready))
  hipLaunchKernelGGL(k_samples, dim3(2 * count + count * ((words + kSamplesThreads - 1) / kSamplesThreads)), dim3(kSamplesThreads), 0, s, pl, first, count);
}

// ------------------------------------------------------------------------------------------------
// k_bfs: MapGridCostFunction::prepare (map_grid_cost_function.cpp:59-68) =
//   MapGrid::resetPathDist + adjustPlanResolution (:135-171) + setTargetCells (:174-213) |
//   setLocalGoal (:216-258) + computeTargetDistance (:262-310) with updatePathCell (:103-122).
// The FIFO wavefront over a 4-connected unit-cost grid has a unique answer, so it is computed
// level-synchronously: the grid is a bitmap (32 cells per word), one 1024-thread workgroup owns
// one grid, every thread keeps the `free` and `visited` words it owns in registers and only the
// two frontier bitmaps live in LDS.  One barrier per level.
// Kept from the reference: seeds are distance 0 whatever their cost (only != 255 is checked) and
// do propagate; an obstacle cell gets obstacleCosts() = N only when a visited free neighbour
// touches it, else it stays unreachableCellCosts() = N+1; obstacle cells never propagate.
// which: 0 = path_costs_ (and alignment_costs_), 1 = goal_costs_, 2 = goal_front_costs_.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t blockExclusiveScan1024(uint32_t v, uint32_t* s_wave, uint32_t* total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t t = __shfl_up(incl, off);
    if ((int)lane >= off) incl += t;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  const uint32_t nw = blockDim.x >> 6;
  for (uint32_t w = 0; w < nw; ++w) {
    uint32_t t = s_wave[w];
    if (w < wave) base += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}
__device__ __forceinline__ uint32_t blockMin1024(uint32_t v, uint32_t* s_wave) {
  for (int off = 32; off > 0; off >>= 1) v = min(v, (uint32_t)__shfl_down(v, off));
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_wave[wave] = v;
  __syncthreads();
  uint32_t r = 0xFFFFFFFFu;
  const uint32_t nw = blockDim.x >> 6;
  for (uint32_t w = 0; w < nw; ++w) r = min(r, s_wave[w]);
  __syncthreads();
  return r;
}

// enumerate the adjusted plan points of original pose i (the inserted ones first, then the pose
// itself), calling f(local_index, x, y); returns how many there are.  adjustPlanResolution :135-171
template <class F>
__device__ __forceinline__ uint32_t adjustedPoints(const double* P, uint32_t i, double x_last_override, double y_last_override,
                                                   bool override_last, uint32_t n, double resolution, bool count_only, F&& f) {
  auto px = [&](uint32_t k) { return (override_last && k == n - 1) ? x_last_override : P[2 * k]; };
  auto py = [&](uint32_t k) { return (override_last && k == n - 1) ? y_last_override : P[2 * k + 1]; };
  const double loop_x = px(i), loop_y = py(i);
  uint32_t cnt = 0;
  if (i > 0) {
    const double last_x = px(i - 1), last_y = py(i - 1);
    const double min_sq_resolution = resolution * resolution * 4;
    double sqdist = (loop_x - last_x) * (loop_x - last_x) + (loop_y - last_y) * (loop_y - last_y);
    if (sqdist > min_sq_resolution) {
      int steps = (int)(((sqrt(sqdist) - sqrt(min_sq_resolution)) / resolution) - 1);
      if (steps > 1) {
        if (!count_only) {
          double deltax = (loop_x - last_x) / steps;
          double deltay = (loop_y - last_y) / steps;
          for (int j = 1; j < steps; ++j) f(cnt + j - 1, last_x + j * deltax, last_y + j * deltay);
        }
        cnt += steps - 1;
      }
    }
  }
  if (!count_only) f(cnt, loop_x, loop_y);
  return cnt + 1;
}

// Thread -> bitmap mapping: the grid has W = ceil(nx/32) word columns; thread t owns the vertical
// strip of RPT rows [r0, r0+RPT) in word column wi (t = strip*W + wi).  Its `free`, `visited` and
// current-frontier words stay in registers; only the left/right words and the rows just above and
// below the strip are read from the LDS copy of the frontier (2*RPT+2 reads instead of 5*RPT).
// Distances are not scattered cell by cell inside the level loop (divergent bit loops and one store
// instruction per new cell were the bulk of a level's cost): each owned word carries kPlanes
// bit-sliced level planes in registers, `plane[b] |= newly_visited` when bit b of the level is set,
// and the 32 distances of a word are decoded and stored once, 16 bytes at a time, after the loop.
// Levels >= 2^kPlanes (corridor mazes) fall back to direct stores and a `late` bitmap in LDS.
// Measured on MI355X (tools/microbench/lds_latency.hip): barrier 64 clk, dependent LDS read ~90 clk,
// returning LDS atomics ~1 lane/clk/CU — hence no atomics anywhere in the level loop.
constexpr int kPlanes = 10;
// MapCell::within_robot (set by the legacy TrajectoryPlanner for path_map_ only, trajectory_planner.cpp:918-930):
// obstacle cells under the robot's own footprint propagate like free cells (map_grid.cpp:109-115)
__device__ __forceinline__ uint32_t bfsWithinWord(const PlannerDev& pl, int which, uint32_t inst, uint32_t row, uint32_t W, uint32_t wi) {
  return (pl.within != nullptr && which == 0) ? pl.within[((size_t)inst * pl.ny + row) * W + wi] : 0u;
}
template <int RPT>
__global__ __launch_bounds__(1024) void k_bfs(PlannerDev pl, uint32_t first) {
  extern __shared__ __align__(16) uint32_t sm[];
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_flag[3];
  const int which = (int)pl.bfs_grids - 1 - (int)blockIdx.y;  // longest searches (goal grids) are dispatched first
  const uint32_t inst = first + blockIdx.x;
  const uint32_t tid = threadIdx.x;
  const Geom g = geomOf(pl, inst);
  const uint32_t nx = pl.nx, ny = pl.ny, W = (nx + 31) >> 5;
  // frontier bitmaps are padded with a zero border (one word left/right, one row above/below, rows
  // rounded up to whole strips) so that every neighbour read and every store is unconditional
  const uint32_t strips = (ny + RPT - 1) / RPT;
  const uint32_t Wp = W + 2, padded = (strips * RPT + 2) * Wp;
  uint32_t* cur = sm;
  uint32_t* nxt = sm + padded;
  uint32_t* late = sm + 2 * padded;  // [words] cells reached at level >= 2^kPlanes (stored directly)
  const uint8_t* master = pl.master + (size_t)inst * pl.cells_padded;
  uint32_t* dist = (which == 0 ? pl.path : (which == 1 ? pl.goal : pl.goal_front)) + (size_t)inst * pl.cells;
  const uint32_t N_obst = pl.cells, N_unreach = pl.cells + 1;
  const uint32_t unknown_is_obstacle = pl.cfg.allow_unknown != 0 ? 0u : 1u;
  const bool owner = tid < strips * W;
  const uint32_t wi = owner ? tid % W : 0, r0 = owner ? (tid / W) * RPT : 0;
  const uint32_t last_mask = (nx & 31) ? ((1u << (nx & 31)) - 1u) : 0xFFFFFFFFu;
  const uint32_t col_mask = (wi + 1 == W) ? last_mask : 0xFFFFFFFFu;
  const bool aligned4 = (nx & 3) == 0;

  // --- owned words: `free` bitmap from the costmap (updatePathCell's obstacle test), frontiers cleared
  uint32_t freeb[RPT], visited[RPT], fr[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    freeb[k] = 0;
    visited[k] = 0xFFFFFFFFu;  // rows beyond the grid never take part
    fr[k] = 0;
    const uint32_t row = r0 + k;
    if (owner) {
      cur[(row + 1) * Wp + wi + 1] = 0;
      nxt[(row + 1) * Wp + wi + 1] = 0;
    }
    if (owner && row < ny) {
      const uint32_t nb = min(32u, nx - wi * 32);
      uint32_t bits = 0;
      if (aligned4) {
        const uint32_t* p4 = reinterpret_cast<const uint32_t*>(master + row * nx + wi * 32);
        for (uint32_t q = 0; q < nb / 4; ++q) {
          const uint32_t v = p4[q];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t cst = (v >> (8 * j)) & 0xFFu;
            const bool obstacle = cst == kLethal || cst == kInscribed || (cst == kNoInfo && unknown_is_obstacle);
            bits |= (obstacle ? 0u : 1u) << (4 * q + j);
          }
        }
      } else {
        const uint8_t* p = master + row * nx + wi * 32;
        for (uint32_t b = 0; b < nb; ++b) {
          const uint32_t cst = p[b];
          const bool obstacle = cst == kLethal || cst == kInscribed || (cst == kNoInfo && unknown_is_obstacle);
          bits |= (obstacle ? 0u : 1u) << b;
        }
      }
      freeb[k] = bits | (bfsWithinWord(pl, which, inst, row, W, wi) & col_mask);
      visited[k] = ~col_mask;  // bits past the last column count as visited
      late[row * W + wi] = 0;
    }
  }
  // zero border
  for (uint32_t i = tid; i < Wp; i += blockDim.x) {
    cur[i] = 0;
    nxt[i] = 0;
    cur[(strips * RPT + 1) * Wp + i] = 0;
    nxt[(strips * RPT + 1) * Wp + i] = 0;
  }
  for (uint32_t i = tid; i < strips * RPT + 2; i += blockDim.x) {
    cur[i * Wp] = 0;
    nxt[i * Wp] = 0;
    cur[i * Wp + Wp - 1] = 0;
    nxt[i * Wp + Wp - 1] = 0;
  }
  if (tid < 3) s_flag[tid] = 0;
  __syncthreads();

  // --- seeds from the plan
  {
    const uint32_t n = pl.plan_count[inst];
    const double* P = pl.plan + (size_t)inst * pl.max_plan * 2;
    const bool ovr = which == 2;
    const double lx = pl.front_last[2 * inst], ly = pl.front_last[2 * inst + 1];
    const uint32_t chunk = (n + blockDim.x - 1) / blockDim.x;
    const uint32_t i0 = min(n, tid * chunk), i1 = min(n, i0 + chunk);
    uint32_t mine = 0;
    for (uint32_t i = i0; i < i1; ++i) mine += adjustedPoints(P, i, lx, ly, ovr, n, g.res, true, [](uint32_t, double, double) {});
    uint32_t total;
    const uint32_t base = blockExclusiveScan1024(mine, s_wave, &total);
    auto valid = [&](double x, double y, uint32_t& cell) {
      uint32_t mx, my;
      if (!worldToMap(g, x, y, mx, my)) return false;
      cell = my * nx + mx;
      return master[cell] != kNoInfo;
    };
    // f = first valid adjusted index
    uint32_t fmin_ = 0xFFFFFFFFu, b = base;
    for (uint32_t i = i0; i < i1; ++i)
      b += adjustedPoints(P, i, lx, ly, ovr, n, g.res, false, [&](uint32_t k, double x, double y) {
        uint32_t cell;
        if (valid(x, y, cell)) fmin_ = min(fmin_, b + k);
      });
    const uint32_t f = blockMin1024(fmin_, s_wave);
    if (f != 0xFFFFFFFFu) {  // else: no point of the plan is in the map -> every cell stays unreachable
      // e = first invalid adjusted index after f
      uint32_t emin = total;
      b = base;
      for (uint32_t i = i0; i < i1; ++i)
        b += adjustedPoints(P, i, lx, ly, ovr, n, g.res, false, [&](uint32_t k, double x, double y) {
          uint32_t cell;
          if (b + k > f && !valid(x, y, cell)) emin = min(emin, b + k);
        });
      const uint32_t e = blockMin1024(emin, s_wave);
      b = base;
      for (uint32_t i = i0; i < i1; ++i)
        b += adjustedPoints(P, i, lx, ly, ovr, n, g.res, false, [&](uint32_t k, double x, double y) {
          const uint32_t idx = b + k;
          const bool seed = (which == 0) ? (idx >= f && idx < e) : (idx == e - 1);
          if (!seed) return;
          uint32_t cell;
          if (!valid(x, y, cell)) return;  // cannot happen inside [f, e)
          const uint32_t my = cell / nx, mx = cell - my * nx;
          atomicOr(&cur[(my + 1) * Wp + (mx >> 5) + 1], 1u << (mx & 31));  // a few hundred seeds, once
        });
    }
  }
  __syncthreads();
  uint32_t plane[kPlanes][RPT];
#pragma unroll
  for (int b = 0; b < kPlanes; ++b)
#pragma unroll
    for (int k = 0; k < RPT; ++k) plane[b][k] = 0;
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const uint32_t row = r0 + k;
    if (owner && row < ny) {
      fr[k] = cur[(row + 1) * Wp + wi + 1];
      visited[k] |= fr[k];  // seeds: level 0 == all planes clear
    }
  }

  // --- level-synchronous expansion, ONE barrier per level (three rotating "anything new" flags)
  const uint32_t base_w = r0 * W + wi;               // unpadded (late bitmap, dist rows)
  const uint32_t base_p = (r0 + 1) * Wp + wi + 1;    // padded (frontier bitmaps)
  uint32_t level = 0;
  while (true) {
    const uint32_t lvl1 = level + 1;  // distance of the cells reached in this round
    uint32_t any = 0;
    if (owner) {
      const uint32_t top = cur[base_p - Wp];
      const uint32_t bot = cur[base_p + RPT * Wp];
      uint32_t lw[RPT], rw[RPT];
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        lw[k] = cur[base_p + k * Wp - 1];
        rw[k] = cur[base_p + k * Wp + 1];
      }
      uint32_t cand[RPT];
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const uint32_t fc = fr[k];
        const uint32_t u = k > 0 ? fr[k > 0 ? k - 1 : 0] : top;
        const uint32_t d = k + 1 < RPT ? fr[k + 1 < RPT ? k + 1 : 0] : bot;
        cand[k] = ((fc << 1) | (lw[k] >> 31) | (fc >> 1) | (rw[k] << 31) | u | d) & ~visited[k];
      }
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const uint32_t nf = cand[k] & freeb[k];
        visited[k] |= cand[k];
        fr[k] = nf;
        any |= nf;
        nxt[base_p + k * Wp] = nf;
      }
      if (lvl1 < (1u << kPlanes)) {
#pragma unroll
        for (int b = 0; b < kPlanes; ++b) {
          if (lvl1 & (1u << b)) {  // wave-uniform
#pragma unroll
            for (int k = 0; k < RPT; ++k) plane[b][k] |= cand[k];
          }
        }
      } else {  // rare: direct stores + remember which cells already hold their value
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
          uint32_t q = cand[k];
          if (q) {
            late[base_w + k * W] |= q;
            uint32_t* drow = dist + (r0 + k) * nx + wi * 32;
            while (q) {
              const int bpos = __ffs(q) - 1;
              q &= q - 1;
              drow[bpos] = ((freeb[k] >> bpos) & 1u) ? lvl1 : N_obst;
            }
          }
        }
      }
    }
    if (any) s_flag[level % 3] = 1;
    if (tid == 0) s_flag[(level + 1) % 3] = 0;
    __syncthreads();
    if (!s_flag[level % 3]) break;
    uint32_t* t = cur;
    cur = nxt;
    nxt = t;
    ++level;
  }

  // --- decode: unvisited -> unreachableCellCosts(); touched obstacle -> obstacleCosts(); else level
  if (owner) {
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      if (r0 + k >= ny) continue;
      const uint32_t vis = visited[k], fb = freeb[k], lt = late[base_w + k * W];
      uint32_t* drow = dist + (r0 + k) * nx + wi * 32;
      const uint32_t nb = min(32u, nx - wi * 32);
      auto value = [&](uint32_t bpos) -> uint32_t {
        uint32_t lvl = 0;
#pragma unroll
        for (int b = 0; b < kPlanes; ++b) lvl |= ((plane[b][k] >> bpos) & 1u) << b;
        if (!((vis >> bpos) & 1u)) return N_unreach;
        if (!((fb >> bpos) & 1u) && lvl != 0) return N_obst;  // level 0 on an obstacle cell == a seed
        return lvl;
      };
      if (aligned4 && lt == 0) {
        for (uint32_t q = 0; q < nb / 4; ++q) {
          uint4 v;
          v.x = value(4 * q);
          v.y = value(4 * q + 1);
          v.z = value(4 * q + 2);
          v.w = value(4 * q + 3);
          *reinterpret_cast<uint4*>(drow + 4 * q) = v;
        }
      } else {
        for (uint32_t bpos = 0; bpos < nb; ++bpos)
          if (!((lt >> bpos) & 1u)) drow[bpos] = value(bpos);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_bfs_wave: the strip sweep of k_bfs with the lanes of a wave laid out as whole row segments
// (W bitmap words + one idle separator lane per strip), so the left/right neighbour words come from
// the adjacent LANES (DPP wave shifts, plain VALU) and only the first/last row of every strip goes
// through LDS.  A wave none of whose lanes holds or borders a frontier cell skips the level.  The
// candidate mask is `& ~blocked` with blocked = obstacle | outside | reached, so every candidate is a
// free cell; the obstacle cells TOUCHED by an expanded cell (map_grid.cpp:195-215: they get
// obstacleCosts()) are found once, after the sweep, as the obstacle neighbours of the expanded set.
// ------------------------------------------------------------------------------------------------
// free-cell bitmap word of one map row (bit b = cell wi*32+b is traversable)
// four cost bytes -> four "obstacle" bits (LETHAL, INSCRIBED, and NO_INFORMATION unless unknown cells are allowed), SWAR
__device__ __forceinline__ uint32_t bfsObstacleNibble(uint32_t v, uint32_t unknown_is_obstacle) {
  const uint32_t low = v & 0x7F7F7F7Fu;
  uint32_t m = (low + 0x03030303u) & v & 0x80808080u;                                  // byte >= 253
  if (!unknown_is_obstacle) m &= ~((low + 0x01010101u) & v);                           // ... but not 255
  m >>= 7;
  return (m | (m >> 7) | (m >> 14) | (m >> 21)) & 0xFu;
}
__device__ __forceinline__ uint32_t bfsFreeWord(const uint8_t* master, uint32_t row, uint32_t nx, uint32_t wi,
                                                uint32_t unknown_is_obstacle) {
  const uint32_t nb = min(32u, nx - wi * 32);
  uint32_t bits = 0;
  if ((nx & 15) == 0 && nb == 32) {  // whole word, 16-byte aligned: two wide loads, all in flight together
    const uint4* p = reinterpret_cast<const uint4*>(master + row * nx + wi * 32);
    const uint4 a = p[0], b = p[1];
    const uint32_t obst = bfsObstacleNibble(a.x, unknown_is_obstacle) | (bfsObstacleNibble(a.y, unknown_is_obstacle) << 4) |
                          (bfsObstacleNibble(a.z, unknown_is_obstacle) << 8) | (bfsObstacleNibble(a.w, unknown_is_obstacle) << 12) |
                          (bfsObstacleNibble(b.x, unknown_is_obstacle) << 16) | (bfsObstacleNibble(b.y, unknown_is_obstacle) << 20) |
                          (bfsObstacleNibble(b.z, unknown_is_obstacle) << 24) | (bfsObstacleNibble(b.w, unknown_is_obstacle) << 28);
    return ~obst;
  }
  if ((nx & 3) == 0) {
    const uint32_t* p4 = reinterpret_cast<const uint32_t*>(master + row * nx + wi * 32);
    for (uint32_t q = 0; q < nb / 4; ++q) bits |= (bfsObstacleNibble(p4[q], unknown_is_obstacle) ^ 0xFu) << (4 * q);
  } else {
    const uint8_t* p = master + row * nx + wi * 32;
    for (uint32_t b = 0; b < nb; ++b) {
      const uint32_t cst = p[b];
      const bool obstacle = cst == kLethal || cst == kInscribed || (cst == kNoInfo && unknown_is_obstacle);
      bits |= (obstacle ? 0u : 1u) << b;
    }
  }
  return bits;
}
// value of the lane below / above in the wave (0 at the wave's ends)
__device__ __forceinline__ uint32_t fromLaneBelow(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t fromLaneAbove(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
}

// k_free_bits: the traversable-cell bitmap of every robot's costmap, [ny][W] words, once per launch for the two or three
// wavefronts of a robot (each reads its rows twice).  Inside k_bfs_wave the same 160 KB of cost bytes took 14 wide loads
// per lane that the register budget of the sweep serialises: 16 us per read, against 7 dword loads now.
__global__ __launch_bounds__(256) void k_free_bits(PlannerDev pl, uint32_t first) {
  const uint32_t W = (pl.nx + 31) >> 5, words = pl.ny * W;
  const uint32_t inst = first + blockIdx.y;
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= words) return;
  const uint32_t row = i / W, wi = i - row * W;
  pl.bfs_free[(size_t)inst * words + i] = bfsFreeWord(pl.master + (size_t)inst * pl.cells_padded, row, pl.nx, wi, pl.cfg.allow_unknown != 0 ? 0u : 1u);
}

// LEGACY = the two-grid launch of the legacy TrajectoryPlanner with within_robot bits; the DWA instantiation
// keeps the constants (and the register allocation) it was tuned with
// PL = level planes kept in registers: 10 (epochs of 1023 levels) for RPT 7; 3 (epochs of 7 levels, the cells written out
// at every epoch end) for RPT 13, which extends the kernel to maps of up to 640 x 624 cells
// DIRECT = the variant for BOUNDED searches: no level planes at all.  Only the cells of the robot's region are ever read
// (DESIGN 4a), so the few lanes that own region words store a cell's distance the moment the wavefront reaches it
// and the rest of the workgroup just sweeps; without the 70 plane registers nothing spills (the plane variant writes
// 244 B of scratch per lane and item: 9x the bytes of the region's distances) and a level costs a third fewer vector
// instructions.  split: a launch of the plane variant beside a DIRECT one takes the whole-grid searches only.
template <int RPT, bool LEGACY, int PL, bool DIRECT = false>
__device__ __forceinline__ void bfsWaveGrid(const PlannerDev& pl, const uint32_t inst, const int which, const uint32_t item, const int split = 0) {
  static_assert(!DIRECT || !LEGACY, "the legacy planner always searches the whole grid");
  // Bounded search (pl.bfs_bounded): the critics read a MapGrid only at cells the robot's samples can reach, a box
  // around the robot (k_samples).  A level-synchronous wavefront has every reached cell final, so the sweep may stop
  // once no cell of that box is left open (neither reached nor an obstacle); cells it has not reached by then read
  // unreachableCellCosts() and the host completes the grid before anybody reads it outside the box.
  int bx0 = 0, bx1 = -1, by0 = 0, by1 = -1, care_ok = 0;  // the robot's region (box + 2 cells) and whether its pockets are known
  if (!LEGACY && pl.bfs_bounded) {
    const int4 bb = reinterpret_cast<const int4*>(pl.bfs_box)[2 * inst];
    bx0 = __builtin_amdgcn_readfirstlane(bb.x);
    bx1 = __builtin_amdgcn_readfirstlane(bb.y);
    by0 = __builtin_amdgcn_readfirstlane(bb.z);
    by1 = __builtin_amdgcn_readfirstlane(bb.w);
    care_ok = __builtin_amdgcn_readfirstlane(pl.bfs_box[8 * inst + 4]);
  }
  const bool bounded = !LEGACY && bx1 >= bx0 && by1 >= by0;
  if (DIRECT ? !bounded : (split != 0 && bounded)) return;  // (uniform over the workgroup) the other variant's item
  extern __shared__ __align__(16) uint32_t sm[];
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_flag[3];
  __shared__ uint32_t s_open[3];   // bounded search: some cell of the robot's box is still neither reached nor blocked
  __shared__ uint32_t s_prog[18];  // levels published by wave w at [w + 1]; [0] and [17] are sentinels
  // opaque per item: otherwise everything below that depends only on the thread index and the map size is hoisted
  // out of the item loop of k_bfs_wave and kept in registers across the whole search (measured: 400 B of spills)
  uint32_t tid_ = threadIdx.x, nx_ = pl.nx, ny_ = pl.ny;
  asm volatile("" : "+v"(tid_), "+s"(nx_), "+s"(ny_));
  const uint32_t tid = tid_;
  if (pl.bfs_trace && tid == 0) {
    unsigned long long* t = pl.bfs_trace + (size_t)item * 8;
    t[0] = wall_clock64();
  }
  const Geom g = geomOf(pl, inst);
  const uint32_t nx = nx_, ny = ny_, W = (nx + 31) >> 5;
  const uint32_t strips = (ny + RPT - 1) / RPT;
  const uint32_t L = W + 1, spw = 64u / L;  // lanes per strip (one separator), strips per wave
  const uint32_t lane = tid & 63u, slot = lane / L;
  const uint32_t strip_of = (tid >> 6) * spw + slot;
  const bool owner = slot < spw && lane - slot * L < W && strip_of < strips;
  const uint32_t wi = owner ? lane - slot * L : 0u;
  const uint32_t strip = owner ? strip_of : strips;  // everyone else reads/writes the zero border strip
  const uint32_t r0 = strip * RPT;
  // LDS: seed bitmap, late bitmap (both [rows][W]), two edge buffers [strips + 2][first,last][W]
  const uint32_t rows_p = strips * RPT;
  uint32_t* seedm = sm;
  uint32_t* late = sm + rows_p * W;
  uint32_t* edge = sm + 2 * rows_p * W;
  const uint32_t edge_words = (strips + 2) * 2 * W;
  const uint8_t* master = pl.master + (size_t)inst * pl.cells_padded;
  const uint32_t* freew = pl.bfs_free + (size_t)inst * ny * W;  // traversable-cell bitmap of the robot's costmap (k_free_bits)
  uint32_t* dist = (which == 0 ? pl.path : (which == 1 ? pl.goal : pl.goal_front)) + (size_t)inst * pl.cells;
  const uint32_t N_obst = pl.cells, N_unreach = pl.cells + 1;
  const uint32_t last_mask = (nx & 31) ? ((1u << (nx & 31)) - 1u) : 0xFFFFFFFFu;
  const uint32_t col_mask = (wi + 1 == W) ? last_mask : 0xFFFFFFFFu;
  const bool aligned4 = (nx & 3) == 0;

  for (uint32_t i = tid; i < 2 * rows_p * W + 2 * edge_words; i += blockDim.x) sm[i] = 0;
  if (tid < 3) s_flag[tid] = s_open[tid] = 0;
  __syncthreads();
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 4] = wall_clock64();

  // --- seeds from the plan (as k_bfs)
  {
    const uint32_t n = pl.plan_count[inst];
    const double* P = pl.plan + (size_t)inst * pl.max_plan * 2;
    const bool ovr = which == 2;
    const double lx = pl.front_last[2 * inst], ly = pl.front_last[2 * inst + 1];
    const uint32_t chunk = (n + blockDim.x - 1) / blockDim.x;
    const uint32_t i0 = min(n, tid * chunk), i1 = min(n, i0 + chunk);
    uint32_t mine = 0;
    for (uint32_t i = i0; i < i1; ++i) mine += adjustedPoints(P, i, lx, ly, ovr, n, g.res, true, [](uint32_t, double, double) {});
    uint32_t total;
    const uint32_t base = blockExclusiveScan1024(mine, s_wave, &total);
    auto valid = [&](double x, double y, uint32_t& cell) {
      uint32_t mx, my;
      if (!worldToMap(g, x, y, mx, my)) return false;
      cell = my * nx + mx;
      return master[cell] != kNoInfo;
    };
    uint32_t fmin_ = 0xFFFFFFFFu, b = base;
    for (uint32_t i = i0; i < i1; ++i)
      b += adjustedPoints(P, i, lx, ly, ovr, n, g.res, false, [&](uint32_t k, double x, double y) {
        uint32_t cell;
        if (valid(x, y, cell)) fmin_ = min(fmin_, b + k);
      });
    const uint32_t f = blockMin1024(fmin_, s_wave);
    if (f != 0xFFFFFFFFu) {
      uint32_t emin = total;
      b = base;
      for (uint32_t i = i0; i < i1; ++i)
        b += adjustedPoints(P, i, lx, ly, ovr, n, g.res, false, [&](uint32_t k, double x, double y) {
          uint32_t cell;
          if (b + k > f && !valid(x, y, cell)) emin = min(emin, b + k);
        });
      const uint32_t e = blockMin1024(emin, s_wave);
      b = base;
      for (uint32_t i = i0; i < i1; ++i)
        b += adjustedPoints(P, i, lx, ly, ovr, n, g.res, false, [&](uint32_t k, double x, double y) {
          const uint32_t idx = b + k;
          const bool seed = (which == 0) ? (idx >= f && idx < e) : (idx == e - 1);
          if (!seed) return;
          uint32_t cell;
          if (!valid(x, y, cell)) return;
          const uint32_t my = cell / nx, mx = cell - my * nx;
          atomicOr(&seedm[my * W + (mx >> 5)], 1u << (mx & 31));  // a few hundred seeds, once
        });
    }
  }
  __syncthreads();

  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 5] = wall_clock64();
  uint32_t blocked[RPT], fr[RPT];
  uint32_t plane[DIRECT ? 1 : PL][DIRECT ? 1 : RPT];
  if constexpr (!DIRECT) {
#pragma unroll
    for (int b = 0; b < PL; ++b)
#pragma unroll
      for (int k = 0; k < RPT; ++k) plane[b][k] = 0;
  }
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    blocked[k] = 0xFFFFFFFFu;  // rows beyond the grid and idle lanes never take part
    fr[k] = 0;
    const uint32_t row = r0 + k;
    if (owner && row < ny) {
      fr[k] = seedm[row * W + wi];  // seeds expand whatever their cost (map_grid.cpp:160-187)
      blocked[k] = ~((freew[row * W + wi] | (LEGACY ? bfsWithinWord(pl, which, inst, row, W, wi) : 0u)) & col_mask) | fr[k];
    }
  }
  // edge rows of the frontier: E(buffer, strip s, first/last, wi); strip index shifted by one (zero border)
  const uint32_t e_mine = (strip + 1) * 2 * W + wi;
  const uint32_t e_top = strip * 2 * W + W + wi;        // last row of the strip above
  const uint32_t e_bot = (strip + 2) * 2 * W + wi;      // first row of the strip below
  uint32_t* ecur = edge;
  uint32_t* enxt = edge + edge_words;
  if (owner) {
    ecur[e_mine] = fr[0];
    ecur[e_mine + W] = fr[RPT - 1];
  }
  if (tid < 18) s_prog[tid] = (tid == 0 || tid == 17) ? 0xFFFFFFFFu : 1u;
  __syncthreads();

  // --- level-synchronous expansion, ONE barrier per level (three rotating "anything new" flags).
  // Levels are recorded bit-sliced relative to an epoch of 2^PL - 1 levels; in the rare case of a
  // longer search the cells of a finished epoch are written out and the planes start again.
  const uint32_t base_w = r0 * W + wi;
  constexpr uint32_t kEpoch = DIRECT ? 0xFFFFFFFFu : (1u << PL) - 1u;  // (a DIRECT search stores whole distances: no epochs)
  constexpr int kLow = 3;  // planes 0..2 are updated every level, the others once per block of 8 levels
  uint32_t level = 0, epoch_base = 0;
  uint32_t bstart[DIRECT ? 1 : RPT];    // `blocked` at the start of the current block
  if constexpr (!DIRECT) {
#pragma unroll
    for (int k = 0; k < RPT; ++k) bstart[k] = blocked[k];
  }
  auto closeBlock = [&](uint32_t hi) {  // cells reached since the block began get the block's high bits
    if constexpr (!DIRECT) {
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const uint32_t got = blocked[k] & ~bstart[k];
        bstart[k] = blocked[k];
#pragma unroll
        for (int b = kLow; b < PL; ++b)
          if ((hi >> (b - kLow)) & 1u) plane[b][k] |= got;
      }
    }
  };
  // one neighbour-expansion of row k: fc = the row's frontier, up/down = the rows above/below
  auto expandRow = [&](int k, uint32_t fc, uint32_t up, uint32_t down) -> uint32_t {
    const uint32_t lwv = fromLaneBelow(fc), rwv = fromLaneAbove(fc);
    const uint32_t x = __builtin_amdgcn_alignbit(fc, lwv, 31) | __builtin_amdgcn_alignbit(rwv, fc, 1) | up;
    const uint32_t cand = (x | down) & ~blocked[k];
    uint32_t nb;  // blocked | x | down in ONE instruction (the compiler shares x | down and spends two)
    asm("v_or3_b32 %0, %1, %2, %3" : "=v"(nb) : "v"(blocked[k]), "v"(x), "v"(down));
    blocked[k] = nb;
    return cand;
  };
  // explicit LDS pointer: a plain volatile pointer degrades to FLAT accesses, which are neither cheap
  // nor ordered with the ds_writes of the edge rows
  typedef volatile __attribute__((address_space(3))) uint32_t lds_vu32;
  lds_vu32* vprog = (lds_vu32*)s_prog;
  const uint32_t wave_id = tid >> 6;
  uint32_t group = 0, any_grp = 0;  // termination is checked once per block of 2^kLow levels
  uint32_t had = 0;                 // OR of this lane's frontier words
  const bool wave_in_box = bounded && (int)(wave_id * spw * RPT) <= by1 && (int)((wave_id + 1) * spw * RPT) > by0;
  // DIRECT: the lanes that own words of the region store the distances themselves, cell by cell as they are reached
  // (a word of the region sees new cells on a few dozen of the levels; every other lane never enters)
  const bool region_lane = DIRECT && owner && wave_in_box && (int)wi >= (bx0 >> 5) && (int)wi <= (bx1 >> 5);
  auto storeCells = [&](int k, uint32_t cells, uint32_t value) {
    const uint32_t row = r0 + k;
    if (cells == 0 || (int)row < by0 || (int)row > by1 || row >= ny) return;
    uint32_t* drow = dist + row * nx + wi * 32;
    while (cells) {
      const uint32_t bpos = (uint32_t)__ffs(cells) - 1u;
      cells &= cells - 1;
      drow[bpos] = value;
    }
  };
  if (region_lane) {
#pragma unroll
    for (int k = 0; k < RPT; ++k) storeCells(k, fr[k], 0u);  // the seeds: distance 0
  }
#pragma unroll
  for (int k = 0; k < RPT; ++k) had |= fr[k];
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 2] = wall_clock64();
  while (true) {
    bool done = false;
    uint32_t code = 0;
    while (true) {
      code = level + 1 - epoch_base;  // 1..kEpoch: distance of this round's cells, relative to the epoch
      // A level needs the edge rows of the strips above and below only, i.e. of the two neighbouring
      // WAVES: wait for those instead of a workgroup barrier, so the waves of a SIMD drift apart and
      // fill each other's stalls (edge buffers alternate by level; a neighbour is at most one level ahead)
      for (uint32_t spins = 0; spins < (1u << 20); ++spins) {  // bounded: a wave is never left spinning
        const uint32_t pa = vprog[wave_id], pb = vprog[wave_id + 2];
        if (__builtin_amdgcn_readfirstlane(min(pa, pb)) >= level + 1) break;
        __builtin_amdgcn_s_sleep(1);
      }
      asm volatile("" ::: "memory");
      const uint32_t top = ecur[e_top], bot = ecur[e_bot];
      // a wave none of whose lanes holds or borders a frontier cell has nothing to do this level
      if (__builtin_amdgcn_ballot_w64((had | top | bot) != 0) != 0) {
        asm volatile("" ::: "memory");
        // rows 1..RPT-2 first: they do not wait for the LDS reads
        const uint32_t old_first = fr[0], old_second = fr[1], old_before_last = fr[RPT - 2];
        uint32_t prev = old_first, held = 0;
#pragma unroll
        for (int k = 1; k < RPT - 1; ++k) {
          const uint32_t fc = fr[k];
          const uint32_t cand = expandRow(k, fc, prev, fr[k + 1]);
          prev = fc;
          if (k > 1) fr[k - 1] = held;  // row k-1's new frontier, once row k has used the old one
          held = cand;
        }
        fr[RPT - 2] = held;
        fr[0] = expandRow(0, old_first, top, old_second);
        fr[RPT - 1] = expandRow(RPT - 1, fr[RPT - 1], old_before_last, bot);
        had = 0;
#pragma unroll
        for (int k = 0; k < RPT; ++k) had |= fr[k];
        any_grp |= had;
        if constexpr (DIRECT) {
          if (wave_in_box) {  // wave-uniform
            if (region_lane && had) {
#pragma unroll
              for (int k = 0; k < RPT; ++k) storeCells(k, fr[k], level + 1);
            }
          }
        } else {
#pragma unroll
          for (int b = 0; b < kLow; ++b) {
            if (code & (1u << b)) {  // wave-uniform
              asm volatile("" ::: "memory");  // keep it a branch: half of these are skipped
#pragma unroll
              for (int k = 0; k < RPT; ++k) plane[b][k] |= fr[k];
            }
          }
        }
      }
      enxt[e_mine] = fr[0];
      enxt[e_mine + W] = fr[RPT - 1];
      asm volatile("" ::: "memory");
      if (lane == 0) vprog[wave_id + 1] = level + 2;  // LDS keeps a wave's operations in order
      const bool group_end = (code & ((1u << kLow) - 1u)) == (1u << kLow) - 1u;
      if (group_end) {
        closeBlock(code >> kLow);
        if (any_grp) s_flag[group % 3] = 1;
        if (wave_in_box) {  // wave-uniform; recomputed from scratch so that nothing of it lives across the levels
          uint32_t wi_v = wi, r0_v = r0;
          asm volatile("" : "+v"(wi_v), "+v"(r0_v));
          const int c_lo = max(bx0 - (int)(wi_v * 32), 0), c_hi = min(bx1 - (int)(wi_v * 32), 31);
          // care words (k_samples): the box cells that are not in a pocket; stored per region row, four words from the region's first
          const uint32_t cw_i = wi_v - (uint32_t)(bx0 >> 5);
          const uint32_t* care = pl.bfs_care + (size_t)inst * kCareRows * kCareWords + cw_i;
          const bool has_care = care_ok != 0 && cw_i < (uint32_t)kCareWords;
          uint32_t open = 0;  // open cells of the box that still count + frontier cells inside the region
#pragma unroll
          for (int k = 0; k < RPT; ++k) {
            const uint32_t rr = (uint32_t)((int)(r0_v + k) - by0);
            if (rr <= (uint32_t)(by1 - by0)) open |= (~blocked[k] & (has_care ? care[rr * kCareWords] : (care_ok ? 0u : 0xFFFFFFFFu))) | fr[k];
          }
          if (c_hi >= c_lo && (open & (0xFFFFFFFFu >> (31 - c_hi)) & (0xFFFFFFFFu << c_lo)) != 0) s_open[group % 3] = 1;
        }
        if (tid == 0) {
          s_flag[(group + 1) % 3] = 0;
          s_open[(group + 1) % 3] = 0;
        }
        __syncthreads();
      }
      uint32_t* t = ecur;
      ecur = enxt;
      enxt = t;
      ++level;
      if (group_end) {
        done = !s_flag[group % 3] || (bounded && !s_open[group % 3]);  // nothing new in a whole block of levels (or nothing open in the box): the search is over
        ++group;
        any_grp = 0;
        if (done || code == kEpoch) break;
      }
    }
    if (done) break;
    // epoch full: write its cells out, remember them in `late`, restart the planes
    if constexpr (!DIRECT) {
      uint32_t cell0 = r0 * nx + wi * 32;
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        uint32_t m = 0;
#pragma unroll
        for (int b = 0; b < PL; ++b) m |= plane[b][k];
        if (m) late[base_w + k * W] |= m;
        while (m) {
          const uint32_t bpos = (uint32_t)__ffs(m) - 1u;
          m &= m - 1;
          uint32_t code = 0;
#pragma unroll
          for (int b = 0; b < PL; ++b) code |= ((plane[b][k] >> bpos) & 1u) << b;
          dist[cell0 + k * nx + bpos] = epoch_base + code;
        }
#pragma unroll
        for (int b = 0; b < PL; ++b) plane[b][k] = 0;
      }
      epoch_base += kEpoch;
    }
  }

  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 3] = wall_clock64();
  // --- expanded set = reached free cells + seeds; its obstacle neighbours were touched
  uint32_t ex[RPT], freeb[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const uint32_t row = r0 + k;
    const bool in = owner && row < ny;
    freeb[k] = in ? ((freew[row * W + wi] | (LEGACY ? bfsWithinWord(pl, which, inst, row, W, wi) : 0u)) & col_mask) : 0u;
    ex[k] = in ? ((blocked[k] & freeb[k]) | seedm[row * W + wi]) : 0u;
  }
  if (owner) {
    enxt[e_mine] = ex[0];
    enxt[e_mine + W] = ex[RPT - 1];
  }
  __syncthreads();
  {
    uint32_t prev = enxt[e_top];
    const uint32_t bot = enxt[e_bot];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const uint32_t fc = ex[k];
      const uint32_t lwv = fromLaneBelow(fc), rwv = fromLaneAbove(fc);
      const uint32_t d = k + 1 < RPT ? ex[k + 1 < RPT ? k + 1 : 0] : bot;
      const uint32_t nb = (fc << 1) | (lwv >> 31) | (fc >> 1) | (rwv << 31) | prev | d;
      prev = fc;
      blocked[k] = nb & ~freeb[k] & ~fc & col_mask;  // reuse: the touched obstacle cells
    }
  }
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 6] = wall_clock64();
  // --- decode: expanded -> level (0 for seeds); touched obstacle -> obstacleCosts(); else unreachableCellCosts()
  uint32_t lt[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) lt[k] = (owner && r0 + k < ny) ? late[base_w + k * W] : 0u;
  auto cellValue = [&](const uint32_t (&pl10)[PL], uint32_t exw, uint32_t tkw, uint32_t bpos) -> uint32_t {
    uint32_t lvl = 0;
#pragma unroll
    for (int b = 0; b < PL; ++b) lvl |= ((pl10[b] >> bpos) & 1u) << b;
    if ((exw >> bpos) & 1u) return lvl ? epoch_base + lvl : 0u;  // relative code 0 == a seed
    return ((tkw >> bpos) & 1u) ? N_obst : N_unreach;
  };
  // Coalesced path: a lane's word is 128 B of a row; stored directly, one instruction would put 16 B into
  // 64 different lines.  The wave passes its words (planes, expanded, touched, late) through a private
  // LDS area instead, 16 words per bitmap word, and eight lanes decode one word: a store instruction then
  // covers 8 whole words = 1 KB of contiguous distances.
  const uint32_t nwords = spw * W;
  const bool coalesced = aligned4 && (size_t)16 * nwords * 16 <= (size_t)2 * rows_p * W + 2 * edge_words;
  // A bounded search is only ever read inside the robot's region (the host completes the grid before anything else
  // looks at it), so only those rows and words are written: 65 x 96 cells instead of 400 x 400.
  const uint32_t dec_y0 = bounded ? (uint32_t)by0 : 0u, dec_rows = bounded ? (uint32_t)(by1 - by0 + 1) : ny;
  const uint32_t dec_w0 = bounded ? (uint32_t)(bx0 >> 5) : 0u, dec_w1 = bounded ? (uint32_t)(bx1 >> 5) : W - 1;
  __syncthreads();  // everyone is done with seedm / late / the edge buffers
  if constexpr (DIRECT) {
    // the reached cells have their distances already; what is left of the region: obstacle cells an expanded cell
    // touched -> obstacleCosts(), everything else -> unreachableCellCosts()
    if (region_lane) {
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        storeCells(k, blocked[k], N_obst);
        storeCells(k, ~ex[k] & ~blocked[k] & col_mask, N_unreach);
      }
    }
  } else if (bounded && !wave_in_box) {
    // none of this wave's rows is in the region
  } else if (coalesced) {
    uint4* stage = reinterpret_cast<uint4*>(sm) + (size_t)wave_id * nwords * 4;
    const bool holder = slot < spw && lane - slot * L < W;
    const uint32_t myw = holder ? slot * W + (lane - slot * L) : 0u;
    const uint32_t c0 = 4u * (lane & 7u);
    uint32_t cellbase[8], rowj[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t wj = 8u * j + (lane >> 3);
      const uint32_t sj = wj / W, wij = wj - sj * W, strip_j = wave_id * spw + sj;
      const bool ok = wj < nwords && strip_j < strips && wij * 32 + c0 < nx && wij >= dec_w0 && wij <= dec_w1;
      rowj[j] = ok ? strip_j * RPT : 0xFFFFFF00u;  // rows of a word that does not exist are never < ny
      cellbase[j] = strip_j * RPT * nx + wij * 32 + c0;
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      if (holder) {  // record of one bitmap word: the planes, expanded, touched, late
        if constexpr (PL == 10) {
          stage[myw * 4 + 0] = make_uint4(plane[0][k], plane[1][k], plane[2][k], plane[3][k]);
          stage[myw * 4 + 1] = make_uint4(plane[4][k], plane[5][k], plane[6][k], plane[7][k]);
          stage[myw * 4 + 2] = make_uint4(plane[8][k], plane[9][k], ex[k], blocked[k]);
          stage[myw * 4 + 3] = make_uint4(lt[k], 0u, 0u, 0u);
        } else {
          static_assert(PL == 10 || PL == 3, "record layout");
          stage[myw * 4 + 0] = make_uint4(plane[0][k], plane[1][k], plane[2][k], ex[k]);
          stage[myw * 4 + 1] = make_uint4(blocked[k], lt[k], 0u, 0u);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (rowj[j] + k - dec_y0 < dec_rows) {
          const uint32_t wj = 8u * j + (lane >> 3);
          uint32_t* out = dist + cellbase[j] + k * nx;
          uint4 v;
          uint32_t l4;  // cells already written when their epoch was flushed
          if constexpr (PL == 10) {
            const uint4 a = stage[wj * 4 + 0], b4 = stage[wj * 4 + 1], c4 = stage[wj * 4 + 2], d4 = stage[wj * 4 + 3];
            const uint32_t pl10[PL] = {a.x, a.y, a.z, a.w, b4.x, b4.y, b4.z, b4.w, c4.x, c4.y};
            v.x = cellValue(pl10, c4.z, c4.w, c0);
            v.y = cellValue(pl10, c4.z, c4.w, c0 + 1);
            v.z = cellValue(pl10, c4.z, c4.w, c0 + 2);
            v.w = cellValue(pl10, c4.z, c4.w, c0 + 3);
            l4 = (d4.x >> c0) & 0xFu;
          } else {
            const uint4 a = stage[wj * 4 + 0], b4 = stage[wj * 4 + 1];
            const uint32_t pl3[PL] = {a.x, a.y, a.z};
            v.x = cellValue(pl3, a.w, b4.x, c0);
            v.y = cellValue(pl3, a.w, b4.x, c0 + 1);
            v.z = cellValue(pl3, a.w, b4.x, c0 + 2);
            v.w = cellValue(pl3, a.w, b4.x, c0 + 3);
            l4 = (b4.y >> c0) & 0xFu;
          }
          if (l4 == 0) {
            *reinterpret_cast<uint4*>(out) = v;
          } else {
            if (!(l4 & 1u)) out[0] = v.x;
            if (!(l4 & 2u)) out[1] = v.y;
            if (!(l4 & 4u)) out[2] = v.z;
            if (!(l4 & 8u)) out[3] = v.w;
          }
        }
      }
    }
  } else if (owner) {
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      if (r0 + k >= ny || r0 + k - dec_y0 >= dec_rows || wi < dec_w0 || wi > dec_w1) continue;
      uint32_t pl10[PL];
      if constexpr (PL == 10) {
        const uint32_t t10[PL] = {plane[0][k], plane[1][k], plane[2][k], plane[3][k], plane[4][k],
                                  plane[5][k], plane[6][k], plane[7][k], plane[8][k], plane[9][k]};
#pragma unroll
        for (int b = 0; b < PL; ++b) pl10[b] = t10[b];
      } else {
#pragma unroll
        for (int b = 0; b < PL; ++b) pl10[b] = plane[b][k];
      }
      uint32_t* drow = dist + (r0 + k) * nx + wi * 32;
      const uint32_t nb = min(32u, nx - wi * 32);
      for (uint32_t bpos = 0; bpos < nb; ++bpos)
        if (!((lt[k] >> bpos) & 1u)) drow[bpos] = cellValue(pl10, ex[k], blocked[k], bpos);
    }
  }
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 1] = wall_clock64() | ((unsigned long long)level << 48);
  if (!LEGACY && tid == 0) pl.bfs_levels[(size_t)inst * 3 + which] = level;  // next cycle's dispatch order
}
// Persistent launch: one workgroup per CU takes (grid, robot) items off a counter until none is left.  The hardware
// hands the workgroups of a plain launch to XCDs and shader engines round-robin and IN ORDER, so with searches of
// very different length (bounded ones end after 60..600 levels) a CU that is done early waits for the head of its
// engine's queue: measured 0.71 ms of work per CU spread over 1.21 ms.  Items are ordered longest first (goal_front,
// goal, path).  Every workgroup leaves the loop as soon as the counter has passed the last item.
// A bounded search keeps few of its 16 waves busy at any level (the wavefront is a thin ring on its way to the robot's
// box) and its levels are a chain of neighbour hand-shakes: latency, not issue slots.  The DIRECT variant is therefore
// built for 64 registers, two workgroups per CU, and a CU works on two searches at once.
template <int RPT, bool LEGACY, int PL, bool DIRECT = false>
__global__ __launch_bounds__(1024, DIRECT ? 8 : 4) void k_bfs_wave(PlannerDev pl, uint32_t first, uint32_t count, uint32_t* next_item, const uint32_t* order, int split) {
  __shared__ uint32_t s_item;
  const uint32_t total = count * (LEGACY ? 2u : 3u);
  for (;;) {
    if (threadIdx.x == 0) s_item = atomicAdd(next_item, 1u);
    __syncthreads();
    const uint32_t slot = s_item;
    if (slot >= total) break;
    const uint32_t item = order ? order[slot] : slot;  // longest searches first: by the level count of the previous cycle (k_samples)
    const uint32_t g = item / count;
    bfsWaveGrid<RPT, LEGACY, PL, DIRECT>(pl, first + (item - g * count), (LEGACY ? 1 : 2) - (int)g, item, split);
    __syncthreads();  // s_item and the LDS staging of the decode are reused by the next item
  }
}
// ------------------------------------------------------------------------------------------------
// k_bfs_rows: the BOUNDED searches of a DWA cycle, one lane per costmap ROW.
// The dense sweep of k_bfs_wave spends its instructions on rows x words that hold no frontier: a wavefront is a thin
// diamond ring, and of the 13 words of a 400-cell row it touches two or three (measured offline on the benchmark's maps:
// 23 % of the (64-row, word) pairs hold or border a frontier cell at a given level).  Here a lane owns one row and keeps
// the row's W words of `blocked` and `frontier` in registers, so
//   * left / right neighbours are the adjacent REGISTERS (one v_alignbit each), up / down the adjacent LANES (DPP);
//   * which GROUPS of four words are live is a wave-uniform bit mask (ballots of the groups that produced new cells,
//     grown by the neighbouring group where a group's first / last word got some, plus the groups in which a
//     neighbouring wave's edge row holds frontier cells): a dead group costs two scalar instructions.  A live group is
//     one straight-line block of four interleaved dependency chains - a level is a latency chain (below), so what
//     counts is the length of a wave's instruction path, not the number of instructions the chip issues;
//   * TEMPORAL BLOCKING: a wave holds 64 rows of which the middle 50 are its own and 7 on either side are copies of
//     its neighbours' rows (with their real `blocked` words).  After an exchange all 64 are exact; every level after
//     that the outermost still-exact row on either side is lost (it lacks a neighbour), so for 7 levels the wave's own
//     50 rows stay exact WITHOUT any communication - no progress words, no polling, no LDS traffic in the level loop.
//     Every 7 levels the waves publish their outer 7 own rows (frontier + blocked), meet at a barrier, take the
//     neighbours' rows into their halo lanes and read the stop flags.  A level of the word-by-word, handshake-per-level
//     form of this kernel cost 4 265 clocks per wave (NAVGPU_BFS_STATS, tools/probe_bfs_stats.py): 1 735 waiting for the
//     neighbours, 1 124 in 4.7 live words (a scalar branch pair and a dependent 8-instruction chain each; one wave
//     issues one vector instruction per 4 clocks), 886 at the barrier that then closed every 8 levels.  A level is a
//     latency chain: what counts is the length of ONE wave's instruction path and how often it must wait for another;
//   * 400 rows are 8 waves of 50: three searches per CU are resident at once (80 registers), and the 768 items of the
//     256-robot fleet all run side by side.
// Distances are stored the moment the wavefront reaches a cell of the robot's region, by the lane whose row
// it is; the touched obstacles and the unreached cells of the region after the sweep.  Everything else (seeds from the
// plan, the stop condition with its pockets, bfs_levels for the dispatch order) is k_bfs_wave's.
// map_grid.cpp:103-122, 174-310.
// ------------------------------------------------------------------------------------------------
#ifdef NAVGPU_BFS_STATS  // experiment builds only (make EXTRA=-DNAVGPU_BFS_STATS, tools/probe_bfs_stats.py): where a level's time goes
__device__ unsigned long long g_bfs_stats[16];  // shader clocks per wave: [0] poll [1] halo + words [2] stores [3] publish [4] group end; [5] wave-levels [6] spins [7] active wave-levels [8] active groups
#define BFS_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define BFS_ACC(i, x) bst[i] += (x)
#else
#define BFS_STAMP(v)
#define BFS_ACC(i, x)
#endif
constexpr int kRowsHalo = 7;                       // rows a wave copies from either neighbour = levels between two exchanges
constexpr int kRowsPerWave = 64 - 2 * kRowsHalo;   // rows a wave owns
__host__ __device__ inline uint32_t bfs_rows_waves(uint32_t ny) { return (ny + kRowsPerWave - 1) / kRowsPerWave; }
template <int W>
__host__ __device__ inline size_t bfs_rows_lds_words(uint32_t nx, uint32_t ny) {
  constexpr uint32_t Wp = (W + 3) & ~3u;
  const uint32_t nw = bfs_rows_waves(ny);
  return (((size_t)nw * kRowsPerWave * ((nx + 31) >> 5) + 3) & ~(size_t)3) + (size_t)kCareRows * kCareWords + (size_t)(nw + 2) * 2 * kRowsHalo * 2 * Wp;
}
// One group of four words (A B C D, left neighbour word L, right neighbour word R) of one level, skipped as a whole when
// bit g of the wave's active mask is clear.  Per word:
//   x = (f << 1 | left >> 31) | (f >> 1 | right << 31) | up | down;   cand = x & ~blocked;   blocked |= x
// (six vector instructions: two v_alignbit, two v_or with the DPP row shift folded in, v_bitop3, v_or3)
// cand* leave in h* (the words are written back by rowsCommit4 once every group has read the old frontier).
// nz: bit g set when any lane has new cells in the group; lo / hi: when its first / last word has (the neighbouring
// group borders them next level).
__device__ __forceinline__ void rowsGroup4(const int g, const uint32_t aw, uint32_t& nz, uint32_t& lo, uint32_t& hi,
                                           uint32_t& bA, uint32_t& bB, uint32_t& bC, uint32_t& bD, const uint32_t fL, const uint32_t fA,
                                           const uint32_t fB, const uint32_t fC, const uint32_t fD, const uint32_t fR, uint32_t& hA, uint32_t& hB,
                                           uint32_t& hC, uint32_t& hD) {
  uint32_t tA, tB, tC, tD, uA, uB, uC, uD, st;
  asm volatile(
      "s_bitcmp1_b32 %[aw], %[g]\n\t"
      "s_cbranch_scc0 1f\n\t"
      "v_alignbit_b32 %[uA], %[fA], %[fL], 31\n\t"
      "v_alignbit_b32 %[uB], %[fB], %[fA], 31\n\t"
      "v_alignbit_b32 %[uC], %[fC], %[fB], 31\n\t"
      "v_alignbit_b32 %[uD], %[fD], %[fC], 31\n\t"
      "v_alignbit_b32 %[hA], %[fB], %[fA], 1\n\t"
      "v_alignbit_b32 %[hB], %[fC], %[fB], 1\n\t"
      "v_alignbit_b32 %[hC], %[fD], %[fC], 1\n\t"
      "v_alignbit_b32 %[hD], %[fR], %[fD], 1\n\t"
      "v_or_b32_dpp %[tA], %[fA], %[uA] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_or_b32_dpp %[tB], %[fB], %[uB] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_or_b32_dpp %[tC], %[fC], %[uC] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_or_b32_dpp %[tD], %[fD], %[uD] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_or_b32_dpp %[uA], %[fA], %[hA] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_or_b32_dpp %[uB], %[fB], %[hB] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_or_b32_dpp %[uC], %[fC], %[hC] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_or_b32_dpp %[uD], %[fD], %[hD] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_bitop3_b32 %[hA], %[tA], %[bA], %[uA] bitop3:0x32\n\t"
      "v_bitop3_b32 %[hB], %[tB], %[bB], %[uB] bitop3:0x32\n\t"
      "v_bitop3_b32 %[hC], %[tC], %[bC], %[uC] bitop3:0x32\n\t"
      "v_bitop3_b32 %[hD], %[tD], %[bD], %[uD] bitop3:0x32\n\t"
      "v_or3_b32 %[bA], %[bA], %[tA], %[uA]\n\t"
      "v_or3_b32 %[bB], %[bB], %[tB], %[uB]\n\t"
      "v_or3_b32 %[bC], %[bC], %[tC], %[uC]\n\t"
      "v_or3_b32 %[bD], %[bD], %[tD], %[uD]\n\t"
      "v_or3_b32 %[tA], %[hA], %[hB], %[hC]\n\t"
      "v_or_b32_e32 %[tA], %[tA], %[hD]\n\t"
      "v_cmp_ne_u32_e32 vcc, 0, %[tA]\n\t"
      "s_cbranch_vccz 1f\n\t"
      "s_bitset1_b32 %[nz], %[g]\n\t"
      "v_cmp_ne_u32_e32 vcc, 0, %[hA]\n\t"
      "s_nop 0\n\t"
      "s_cmp_lg_u64 vcc, 0\n\t"
      "s_cselect_b32 %[st], 1, 0\n\t"
      "s_lshl_b32 %[st], %[st], %[g]\n\t"
      "s_or_b32 %[lo], %[lo], %[st]\n\t"
      "v_cmp_ne_u32_e32 vcc, 0, %[hD]\n\t"
      "s_nop 0\n\t"
      "s_cmp_lg_u64 vcc, 0\n\t"
      "s_cselect_b32 %[st], 1, 0\n\t"
      "s_lshl_b32 %[st], %[st], %[g]\n\t"
      "s_or_b32 %[hi], %[hi], %[st]\n\t"
      "1:\n\t"
      : [bA] "+v"(bA), [bB] "+v"(bB), [bC] "+v"(bC), [bD] "+v"(bD), [hA] "=&v"(hA), [hB] "=&v"(hB), [hC] "=&v"(hC), [hD] "=&v"(hD), [nz] "+s"(nz),
        [lo] "+s"(lo), [hi] "+s"(hi), [tA] "=&v"(tA), [tB] "=&v"(tB), [tC] "=&v"(tC), [tD] "=&v"(tD),
        [uA] "=&v"(uA), [uB] "=&v"(uB), [uC] "=&v"(uC), [uD] "=&v"(uD), [st] "=&s"(st)
      : [aw] "s"(aw), [g] "n"(g), [fL] "v"(fL), [fA] "v"(fA), [fB] "v"(fB), [fC] "v"(fC), [fD] "v"(fD), [fR] "v"(fR)
      : "vcc", "scc");
}
__device__ __forceinline__ void rowsCommit4(const int g, const uint32_t aw, uint32_t& fA, uint32_t& fB, uint32_t& fC, uint32_t& fD, const uint32_t hA,
                                            const uint32_t hB, const uint32_t hC, const uint32_t hD) {
  asm volatile(
      "s_bitcmp1_b32 %[aw], %[g]\n\t"
      "s_cbranch_scc0 2f\n\t"
      "v_mov_b32 %[fA], %[hA]\n\t"
      "v_mov_b32 %[fB], %[hB]\n\t"
      "v_mov_b32 %[fC], %[hC]\n\t"
      "v_mov_b32 %[fD], %[hD]\n\t"
      "2:\n\t"
      : [fA] "+v"(fA), [fB] "+v"(fB), [fC] "+v"(fC), [fD] "+v"(fD)
      : [aw] "s"(aw), [g] "n"(g), [hA] "v"(hA), [hB] "v"(hB), [hC] "v"(hC), [hD] "v"(hD)
      : "scc");
}
// The seed cells of wavefront `which` of robot `inst`, from its plan (as bfsWaveGrid; map_grid.cpp:160-187, 190-233): every
// lane of the workgroup takes a slice of the plan, `set(mx, my)` is called once per seed cell.
template <typename Set>
__device__ __forceinline__ void rowsPlanSeeds(const PlannerDev& pl, const uint32_t inst, const int which, const Geom& g, const uint8_t* master,
                                              const uint32_t nx, const uint32_t tid, uint32_t* s_wave, Set&& set) {
  const uint32_t n = pl.plan_count[inst];
  const double* P = pl.plan + (size_t)inst * pl.max_plan * 2;
  const bool ovr = which == 2;
  const double lx = pl.front_last[2 * inst], ly = pl.front_last[2 * inst + 1];
  const uint32_t chunk = (n + blockDim.x - 1) / blockDim.x;
  const uint32_t i0 = min(n, tid * chunk), i1 = min(n, i0 + chunk);
  uint32_t mine = 0;
  for (uint32_t i = i0; i < i1; ++i) mine += adjustedPoints(P, i, lx, ly, ovr, n, g.res, true, [](uint32_t, double, double) {});
  uint32_t total;
  const uint32_t base = blockExclusiveScan1024(mine, s_wave, &total);
  auto valid = [&](double x, double y, uint32_t& cell) {
    uint32_t mx, my;
    if (!worldToMap(g, x, y, mx, my)) return false;
    cell = my * nx + mx;
    return master[cell] != kNoInfo;
  };
  uint32_t fmin_ = 0xFFFFFFFFu, b = base;
  for (uint32_t i = i0; i < i1; ++i)
    b += adjustedPoints(P, i, lx, ly, ovr, n, g.res, false, [&](uint32_t k, double x, double y) {
      uint32_t cell;
      if (valid(x, y, cell)) fmin_ = min(fmin_, b + k);
    });
  const uint32_t f = blockMin1024(fmin_, s_wave);
  if (f == 0xFFFFFFFFu) return;  // (uniform over the workgroup)
  uint32_t emin = total;
  b = base;
  for (uint32_t i = i0; i < i1; ++i)
    b += adjustedPoints(P, i, lx, ly, ovr, n, g.res, false, [&](uint32_t k, double x, double y) {
      uint32_t cell;
      if (b + k > f && !valid(x, y, cell)) emin = min(emin, b + k);
    });
  const uint32_t e = blockMin1024(emin, s_wave);
  b = base;
  for (uint32_t i = i0; i < i1; ++i)
    b += adjustedPoints(P, i, lx, ly, ovr, n, g.res, false, [&](uint32_t k, double x, double y) {
      const uint32_t idx = b + k;
      const bool seed = (which == 0) ? (idx >= f && idx < e) : (idx == e - 1);
      if (!seed) return;
      uint32_t cell;
      if (!valid(x, y, cell)) return;
      const uint32_t my = cell / nx;
      set(cell - my * nx, my);
    });
}
template <int W>
__device__ __forceinline__ void bfsRowsGrid(const PlannerDev& pl, const uint32_t inst, const int which, const uint32_t item) {
  constexpr int NG = (W + 3) / 4;         // groups of four words
  constexpr int WP = NG * 4;              // words kept per row: W rounded up (the extra ones are blocked everywhere)
  constexpr int D = kRowsHalo;
  int bx0 = 0, bx1 = -1, by0 = 0, by1 = -1, care_ok = 0;  // the robot's region (box + 2 cells) and whether its pockets are known
  if (pl.bfs_bounded) {
    const int4 bb = reinterpret_cast<const int4*>(pl.bfs_box)[2 * inst];
    bx0 = __builtin_amdgcn_readfirstlane(bb.x);
    bx1 = __builtin_amdgcn_readfirstlane(bb.y);
    by0 = __builtin_amdgcn_readfirstlane(bb.z);
    by1 = __builtin_amdgcn_readfirstlane(bb.w);
    care_ok = __builtin_amdgcn_readfirstlane(pl.bfs_box[8 * inst + 4]);
  }
  if (!(bx1 >= bx0 && by1 >= by0)) {  // (uniform over the workgroup) a whole-grid search: the region is the map, nothing is ever "settled"
    bx0 = 0;
    by0 = 0;
    bx1 = (int)pl.nx - 1;
    by1 = (int)pl.ny - 1;
    care_ok = 0;
  }
  extern __shared__ __align__(16) uint32_t sm[];
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_flag[3];  // rotating by exchange: something new was reached since the last one
  __shared__ uint32_t s_open[3];  //                       something of the robot's box is still open
  uint32_t tid_ = threadIdx.x, nx_ = pl.nx, ny_ = pl.ny;
  asm volatile("" : "+v"(tid_), "+s"(nx_), "+s"(ny_));  // opaque per item, as in bfsWaveGrid
  const uint32_t tid = tid_;
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8] = wall_clock64();
  const Geom g = geomOf(pl, inst);
  const uint32_t nx = nx_, ny = ny_, Wr = (nx + 31) >> 5;  // Wr <= W words really exist
  const uint32_t nw = blockDim.x >> 6;
  const uint32_t lane = tid & 63u, wave_id = tid >> 6;
  // lanes D .. 63 - D own rows wave * 50 .. wave * 50 + 49; the D lanes on either side copy the neighbouring waves' rows
  const int row_i = (int)(wave_id * kRowsPerWave + lane) - D;
  const bool real = row_i >= 0 && row_i < (int)ny;           // the lane's row exists (own or halo)
  const bool owner = real && lane >= (uint32_t)D && lane < 64u - D;
  const uint32_t row = real ? (uint32_t)row_i : 0u;
  const uint32_t rows_p = nw * kRowsPerWave;
  const uint32_t seed_words = (rows_p * Wr + 3u) & ~3u;
  uint32_t* seedm = sm;                                   // [rows_p][Wr], padded to whole 16 bytes
  uint32_t* care_l = sm + seed_words;                     // [kCareRows][kCareWords]
  uint32_t* edge = care_l + kCareRows * kCareWords;       // [nw + 2][top | bottom][D rows][frontier WP | blocked WP]; slot = wave + 1
  const uint32_t edge_words = (nw + 2) * 2 * D * 2 * WP;
  const uint8_t* master = pl.master + (size_t)inst * pl.cells_padded;
  const uint32_t* freew = pl.bfs_free + (size_t)inst * ny * Wr;
  uint32_t* dist = (which == 0 ? pl.path : (which == 1 ? pl.goal : pl.goal_front)) + (size_t)inst * pl.cells;
  const uint32_t N_obst = pl.cells, N_unreach = pl.cells + 1;
  const uint32_t last_mask = (nx & 31) ? ((1u << (nx & 31)) - 1u) : 0xFFFFFFFFu;
  const bool aligned4 = (nx & 3) == 0;

  for (uint32_t i = tid; i < seed_words + kCareRows * kCareWords + edge_words; i += blockDim.x) sm[i] = 0;
  if (tid < 3) s_flag[tid] = s_open[tid] = 0;
  __syncthreads();
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 4] = wall_clock64();
  if (care_ok) {  // the pocket mask of the robot's box: by region row, four words from the region's first (k_samples)
    const uint32_t* care = pl.bfs_care + (size_t)inst * kCareRows * kCareWords;
    for (uint32_t i = tid; i < (uint32_t)(kCareRows * kCareWords); i += blockDim.x) care_l[i] = care[i];
  }
  // --- seeds from the plan
  rowsPlanSeeds(pl, inst, which, g, master, nx, tid, s_wave, [&](uint32_t mx, uint32_t my) {
    atomicOr(&seedm[my * Wr + (mx >> 5)], 1u << (mx & 31));  // a few hundred seeds, once
  });
  __syncthreads();
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 5] = wall_clock64();

  uint32_t blocked[WP], fr[WP];
#pragma unroll
  for (int j = 0; j < WP; ++j) {
    blocked[j] = 0xFFFFFFFFu;  // rows beyond the grid and words beyond the row never produce cells
    fr[j] = 0;
    if (real && (uint32_t)j < Wr) {  // (halo lanes too: exact copies of the neighbours' rows)
      fr[j] = seedm[row * Wr + j];  // seeds expand whatever their cost (map_grid.cpp:160-187)
      blocked[j] = ~(freew[row * Wr + j] & ((uint32_t)j + 1 == Wr ? last_mask : 0xFFFFFFFFu)) | fr[j];
    }
  }
  // the region in this lane's terms
  const bool wave_in_box = (int)(wave_id * kRowsPerWave) <= by1 && (int)((wave_id + 1) * kRowsPerWave) > by0;  // wave-uniform
  const bool row_in_box = owner && row_i >= by0 && row_i <= by1;
  const int w0 = bx0 >> 5, w1 = bx1 >> 5;
  uint32_t region_groups = 0;  // groups that hold words of the region
  for (int jr = w0; jr <= w1; ++jr) region_groups |= 1u << (jr >> 2);
  uint32_t* drow = dist + (size_t)row * nx;
  // distances of the cells `cells` of word j of this lane's row.  Two plain bit loops (every lane runs the longest one, so
  // their bodies are kept to a find-first-bit, an address and a store): whole aligned groups of four first - fronts that
  // run along a row reach 32 cells of a word at once - then what is left, cell by cell
  auto storeCells = [&](int j, uint32_t cells, uint32_t value) {
    uint32_t* dw = drow + j * 32;
    if (aligned4) {
      uint32_t full = cells & (cells >> 1) & (cells >> 2) & (cells >> 3) & 0x11111111u;
      cells &= ~(full * 15u);
      const uint4 v4 = make_uint4(value, value, value, value);
      while (full) {
        const uint32_t bpos = (uint32_t)__ffs(full) - 1u;
        *reinterpret_cast<uint4*>(dw + bpos) = v4;
        full &= full - 1;
      }
    }
    while (cells) {
      const uint32_t bpos = (uint32_t)__ffs(cells) - 1u;
      dw[bpos] = value;
      cells &= cells - 1;
    }
  };
  // LDS offsets (words) of the 2 * WP-word row record this lane publishes / takes in at an exchange:
  //   own rows 0 .. D-1 (lanes D .. 2D-1) -> this wave's TOP record, read by the wave above into its lanes 64-D .. 63;
  //   own rows 50-D .. 49 (lanes 64-2D .. 63-D) -> BOTTOM record, read by the wave below into its lanes 0 .. D-1
  const bool pub_top = lane >= (uint32_t)D && lane < 2u * D, pub_bot = lane >= 64u - 2 * D && lane < 64u - D;
  const uint32_t pub_wr = (((wave_id + 1) * 2 + (pub_top ? 0u : 1u)) * D + (pub_top ? lane - D : lane - (64u - 2 * D))) * 2 * WP;
  const bool halo_top = lane < (uint32_t)D, halo_bot = lane >= 64u - D;
  const uint32_t halo_rd = ((halo_top ? (wave_id * 2 + 1) : ((wave_id + 2) * 2)) * D + (halo_top ? lane : lane - (64u - D))) * 2 * WP;
  constexpr uint32_t gmask = (1u << NG) - 1u;

  // which groups hold or border a frontier cell of this wave's 64 rows
  auto activity = [&]() -> uint32_t {
    uint32_t nz = 0, lo = 0, hi = 0;
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const uint32_t t = fr[4 * q] | fr[4 * q + 1] | fr[4 * q + 2] | fr[4 * q + 3];
      if (__builtin_amdgcn_ballot_w64(t != 0) != 0) {
        nz |= 1u << q;
        if (__builtin_amdgcn_ballot_w64(fr[4 * q] != 0) != 0) lo |= 1u << q;
        if (__builtin_amdgcn_ballot_w64(fr[4 * q + 3] != 0) != 0) hi |= 1u << q;
      }
    }
    return (nz | (lo >> 1) | (hi << 1)) & gmask;
  };
  if (wave_in_box) {
#pragma unroll
    for (int j = 0; j < W; ++j)
      if (j >= w0 && j <= w1 && row_in_box) storeCells(j, fr[j], 0u);  // the seeds: distance 0
  }
  uint32_t a_own = activity();
  uint32_t level = 0, xch = 0, any_blk = 0;
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 2] = wall_clock64();
  bool done = false;
#ifdef NAVGPU_BFS_STATS
  unsigned long long bst[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  while (!done) {
    // ---- D levels on the wave's own: registers and DPP only
    for (int k = 0; k < D; ++k) {
      BFS_STAMP(ts1);
      BFS_ACC(5, 1);
      const uint32_t aw = __builtin_amdgcn_readfirstlane(a_own);  // (provably uniform, but the "s" operands below need the compiler to know it)
      BFS_ACC(7, aw != 0 ? 1 : 0);
      BFS_ACC(8, __builtin_popcount(aw));
      if (aw != 0) {
        uint32_t nz = 0, lo = 0, hi = 0;
        // One group = one asm statement that carries its own wave-uniform skip, so the compiler sees straight-line code
        // with in-place (tied) updates of `blocked` and `fr`.  (Written as C++ branches the same loop made it rename both
        // arrays per word: register copies in the path of every SKIPPED word and a dozen more at the loop's back edge.)
        // A group's new frontier waits in h[] until the NEXT group has read the old words (its left neighbour), then goes back.
        uint32_t h[2][4];
        const uint32_t zero = 0;
#pragma unroll
        for (int q = 0; q < NG; ++q) {
          rowsGroup4(q, aw, nz, lo, hi, blocked[4 * q], blocked[4 * q + 1], blocked[4 * q + 2], blocked[4 * q + 3],
                     q > 0 ? fr[q > 0 ? 4 * q - 1 : 0] : zero, fr[4 * q], fr[4 * q + 1], fr[4 * q + 2], fr[4 * q + 3],
                     q + 1 < NG ? fr[q + 1 < NG ? 4 * q + 4 : 0] : zero, h[q & 1][0], h[q & 1][1], h[q & 1][2], h[q & 1][3]);
          if (q > 0) {
            const int p = q > 0 ? q - 1 : 0;
            rowsCommit4(p, aw, fr[4 * p], fr[4 * p + 1], fr[4 * p + 2], fr[4 * p + 3], h[p & 1][0], h[p & 1][1], h[p & 1][2], h[p & 1][3]);
          }
        }
        rowsCommit4(NG - 1, aw, fr[4 * (NG - 1)], fr[4 * (NG - 1) + 1], fr[4 * (NG - 1) + 2], fr[4 * (NG - 1) + 3], h[(NG - 1) & 1][0], h[(NG - 1) & 1][1],
                    h[(NG - 1) & 1][2], h[(NG - 1) & 1][3]);
        BFS_STAMP(ts2);
        BFS_ACC(1, ts2 - ts1);
        // the new cells of the robot's region get their distance now, from the lane that owns the row
        if (wave_in_box && (nz & region_groups) != 0) {
#pragma unroll
          for (int q = 0; q < NG; ++q) {
            if (((nz & region_groups) >> q) & 1u) {  // wave-uniform
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const int j = 4 * q + c;
                if (j < W && j >= w0 && j <= w1) {
                  if (row_in_box && fr[j < W ? j : 0] != 0) storeCells(j, fr[j < W ? j : 0], level + 1);
                }
              }
            }
          }
        }
        BFS_STAMP(ts3);
        BFS_ACC(2, ts3 - ts2);
        any_blk |= nz;
        a_own = (nz | (lo >> 1) | (hi << 1)) & gmask;
      }
      ++level;
    }
    // ---- exchange: the outer D own rows go to the neighbours, theirs come into the halo lanes; stop flags
    BFS_STAMP(ts4);
    const uint32_t slot = xch % 3u;
    if (pub_top || pub_bot) {
#pragma unroll
      for (int q = 0; q < WP; q += 4) {
        *reinterpret_cast<uint4*>(edge + pub_wr + q) = make_uint4(fr[q], fr[q + 1], fr[q + 2], fr[q + 3]);
        *reinterpret_cast<uint4*>(edge + pub_wr + WP + q) = make_uint4(blocked[q], blocked[q + 1], blocked[q + 2], blocked[q + 3]);
      }
    }
    if (any_blk) s_flag[slot] = 1;
    if (wave_in_box) {  // wave-uniform: is anything of the robot's box still open, or a frontier cell inside the region?
      uint32_t open_any = 0;
      const uint32_t rr = (uint32_t)(row_i - by0);
#pragma unroll
      for (int j = 0; j < W; ++j) {
        if (j >= w0 && j <= w1) {  // wave-uniform
          const uint32_t cw_i = (uint32_t)(j - w0);
          if (row_in_box) {
            const uint32_t care = (care_ok != 0 && cw_i < (uint32_t)kCareWords && rr < (uint32_t)kCareRows) ? care_l[rr * kCareWords + cw_i] : (care_ok ? 0u : 0xFFFFFFFFu);
            const int c_lo = max(bx0 - j * 32, 0), c_hi = min(bx1 - j * 32, 31);
            const uint32_t open = (~blocked[j] & care) | fr[j];
            if (c_hi >= c_lo) open_any |= open & (0xFFFFFFFFu >> (31 - c_hi)) & (0xFFFFFFFFu << c_lo);
          }
        }
      }
      if (open_any != 0) s_open[slot] = 1;
    }
    if (tid == 0) {
      s_flag[(xch + 1) % 3u] = 0;
      s_open[(xch + 1) % 3u] = 0;
    }
    BFS_STAMP(ts5);
    BFS_ACC(3, ts5 - ts4);
    __syncthreads();
    done = !s_flag[slot] || !s_open[slot];  // nothing new in D levels, or nothing open in the box: the search is over
    if (!done) {
      if ((halo_top || halo_bot) && real) {
#pragma unroll
        for (int q = 0; q < WP; q += 4) {
          const uint4 v = *reinterpret_cast<const uint4*>(edge + halo_rd + q);
          const uint4 b = *reinterpret_cast<const uint4*>(edge + halo_rd + WP + q);
          fr[q] = v.x;
          fr[q + 1] = v.y;
          fr[q + 2] = v.z;
          fr[q + 3] = v.w;
          blocked[q] = b.x;
          blocked[q + 1] = b.y;
          blocked[q + 2] = b.z;
          blocked[q + 3] = b.w;
        }
      }
      a_own = activity();
      __syncthreads();  // the records are free for the next exchange
    }
    ++xch;
    any_blk = 0;
    BFS_STAMP(ts6);
    BFS_ACC(4, ts6 - ts5);
  }
#ifdef NAVGPU_BFS_STATS
  if (lane == 0)
    for (int k = 0; k < 9; ++k) atomicAdd(&g_bfs_stats[k], bst[k]);
#endif
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 3] = wall_clock64();

  // --- the rest of the region: obstacle cells an expanded cell touched -> obstacleCosts(), everything else that was not
  // reached -> unreachableCellCosts().  Expanded = reached free cells + seeds.  (The halo lanes next to the own rows are
  // exact copies as of the last exchange; the own rows have moved on since, so their neighbours are exchanged once more.)
  __syncthreads();
  {
    uint32_t ex[WP], fb[WP];
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      const bool in = owner && (uint32_t)j < Wr;
      fb[j] = in ? (freew[row * Wr + j] & ((uint32_t)j + 1 == Wr ? last_mask : 0xFFFFFFFFu)) : 0u;
      ex[j] = in ? ((blocked[j] & fb[j]) | seedm[row * Wr + j]) : 0u;
    }
    if (pub_top || pub_bot) {
#pragma unroll
      for (int j = 0; j < WP; ++j) edge[pub_wr + j] = ex[j];
    }
    __syncthreads();
    if ((halo_top || halo_bot) && real) {
#pragma unroll
      for (int j = 0; j < WP; ++j) ex[j] = edge[halo_rd + j];
    }
    if (wave_in_box) {
#pragma unroll
      for (int j = 0; j < W; ++j) {
        if (j >= w0 && j <= w1) {  // wave-uniform
          const uint32_t fc = ex[j];
          const uint32_t lw = j > 0 ? ex[j > 0 ? j - 1 : 0] : 0u, rw = j + 1 < WP ? ex[j + 1 < WP ? j + 1 : 0] : 0u;
          const uint32_t nbr = __builtin_amdgcn_alignbit(fc, lw, 31) | __builtin_amdgcn_alignbit(rw, fc, 1) | fromLaneBelow(fc) | fromLaneAbove(fc);
          const uint32_t cm = (uint32_t)j + 1 == Wr ? last_mask : ((uint32_t)j < Wr ? 0xFFFFFFFFu : 0u);
          const uint32_t touched = nbr & ~fb[j] & ~fc & cm;
          if (row_in_box) {
            storeCells(j, touched, N_obst);
            storeCells(j, ~fc & ~touched & cm, N_unreach);
          }
        }
      }
    }
  }
  if (pl.bfs_trace && tid == 0) {
    pl.bfs_trace[(size_t)item * 8 + 6] = wall_clock64();
    pl.bfs_trace[(size_t)item * 8 + 1] = wall_clock64() | ((unsigned long long)level << 48);
  }
  if (tid == 0) pl.bfs_levels[(size_t)inst * 3 + which] = level;  // next cycle's dispatch order
}
template <int W>
__global__ __launch_bounds__(1024, 6) void k_bfs_rows(PlannerDev pl, uint32_t first, uint32_t count, uint32_t* next_item, const uint32_t* order) {
  __shared__ uint32_t s_item;
  const uint32_t total = count * 3u;
  for (;;) {
    if (threadIdx.x == 0) s_item = atomicAdd(next_item, 1u);
    __syncthreads();
    const uint32_t slot = s_item;
    if (slot >= total) break;  // (every workgroup gets here: the counter only grows)
    const uint32_t item = order ? order[slot] : slot;  // longest searches first
    const uint32_t g = item / count;
    bfsRowsGrid<W>(pl, first + (item - g * count), 2 - (int)g, item);
    __syncthreads();
  }
}
// ------------------------------------------------------------------------------------------------
// k_bfs_rows2: the row sweep of k_bfs_rows for maps up to 1024 cells wide and 1344 rows (configs[4]'s 1000 x 1000), TWO rows
// per lane.  A lane keeps rows (A, B) = (2k, 2k + 1) of its wave's block: row A's upper neighbour is row B of the lane
// above (DPP), its lower one the lane's own row B (a register); row B's upper neighbour is the lane's own row A, its lower
// one row A of the lane below.  A wave owns 112 rows (56 lanes) and copies 8 rows (4 lanes) of either neighbour, so the
// waves meet every 8 levels; 1000 rows are 9 waves.  2 x 2 x 32 words of state per lane: 168 registers, three waves per
// SIMD, one search per CU.  The seed bitmap does not fit LDS next to the exchange records and lives in global scratch
// (one per workgroup), read with plain loads behind an agent-scope acquire.  Everything else is k_bfs_rows.
// ------------------------------------------------------------------------------------------------
constexpr int kRows2HaloLanes = 4;                                // lanes a wave copies from either neighbour
constexpr int kRows2Levels = 2 * kRows2HaloLanes;                 // = rows copied = levels between two exchanges
constexpr int kRows2PerWave = 2 * (64 - 2 * kRows2HaloLanes);     // rows a wave owns
constexpr int kRows2Words = 32;
__host__ __device__ inline uint32_t bfs_rows2_waves(uint32_t ny) { return (ny + kRows2PerWave - 1) / kRows2PerWave; }
__host__ __device__ inline size_t bfs_rows2_lds_words(uint32_t ny) {
  return (size_t)kCareRows * kCareWords + (size_t)(bfs_rows2_waves(ny) + 2) * 2 * kRows2HaloLanes * 4 * kRows2Words;
}
// One group of four words of BOTH rows of a lane, one level, in place; skipped as a whole when bit g of the wave's active mask
// is clear.  Per word (rowsGroup4's arithmetic with the vertical neighbours of a row pair):
//   row A: x = left | right | (row B of the lane above) | own row B;   row B: x = left | right | own row A | (row A of the lane below)
//   new frontier = x & ~blocked;   blocked |= x
// The words are updated where they stand, so the old last word of the group is kept in pA / pB for the next group's left
// neighbour.  They are valid only if this group was live; a group that is not holds no frontier cell (it either never had one
// or wrote its empty result back the level it went quiet), so its neighbour shifts in zeros instead (the G > 0 prologue).
// rA / rB: the first word of the next group (still old).  nz / lo / hi as rowsGroup4.
template <int G>
__device__ __forceinline__ void rows2Group(const uint32_t aw, uint32_t& nz, uint32_t& lo, uint32_t& hi, uint32_t* __restrict__ bA, uint32_t* __restrict__ bB,
                                           uint32_t* __restrict__ fA, uint32_t* __restrict__ fB, uint32_t& pA, uint32_t& pB, const uint32_t rA,
                                           const uint32_t rB) {
  uint32_t u0, u1, u2, u3, v0, v1, v2, v3, h0, h1, h2, h3, st;
#define NAVGPU_ROWS2_HEAD_FIRST                \
  "s_bitcmp1_b32 %[aw], %[g]\n\t"              \
  "s_cbranch_scc0 1f\n\t"                      \
  "v_lshlrev_b32 %[u0], 1, %[fA0]\n\t"         \
  "v_lshlrev_b32 %[v0], 1, %[fB0]\n\t"
#define NAVGPU_ROWS2_HEAD_NEXT                 \
  "s_bitcmp1_b32 %[aw], %[g]\n\t"              \
  "s_cbranch_scc0 1f\n\t"                      \
  "s_bitcmp1_b32 %[aw], %[gp]\n\t"             \
  "s_cbranch_scc1 3f\n\t"                      \
  "v_lshlrev_b32 %[u0], 1, %[fA0]\n\t"         \
  "v_lshlrev_b32 %[v0], 1, %[fB0]\n\t"         \
  "s_branch 4f\n\t"                            \
  "3:\n\t"                                     \
  "v_alignbit_b32 %[u0], %[fA0], %[pA], 31\n\t" \
  "v_alignbit_b32 %[v0], %[fB0], %[pB], 31\n\t" \
  "4:\n\t"
#define NAVGPU_ROWS2_BODY(HEAD)                                                                        \
  asm volatile(                                                                                        \
      HEAD                                                                                             \
      "v_alignbit_b32 %[u1], %[fA1], %[fA0], 31\n\t"                                                   \
      "v_alignbit_b32 %[u2], %[fA2], %[fA1], 31\n\t"                                                   \
      "v_alignbit_b32 %[u3], %[fA3], %[fA2], 31\n\t"                                                   \
      "v_alignbit_b32 %[v1], %[fB1], %[fB0], 31\n\t"                                                   \
      "v_alignbit_b32 %[v2], %[fB2], %[fB1], 31\n\t"                                                   \
      "v_alignbit_b32 %[v3], %[fB3], %[fB2], 31\n\t"                                                   \
      "v_mov_b32 %[pA], %[fA3]\n\t"                                                                    \
      "v_mov_b32 %[pB], %[fB3]\n\t"                                                                    \
      "v_alignbit_b32 %[h0], %[fA1], %[fA0], 1\n\t"                                                    \
      "v_alignbit_b32 %[h1], %[fA2], %[fA1], 1\n\t"                                                    \
      "v_alignbit_b32 %[h2], %[fA3], %[fA2], 1\n\t"                                                    \
      "v_alignbit_b32 %[h3], %[rA], %[fA3], 1\n\t"                                                     \
      "v_or_b32_dpp %[u0], %[fB0], %[u0] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
      "v_or_b32_dpp %[u1], %[fB1], %[u1] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
      "v_or_b32_dpp %[u2], %[fB2], %[u2] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
      "v_or_b32_dpp %[u3], %[fB3], %[u3] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
      "v_or3_b32 %[u0], %[u0], %[h0], %[fB0]\n\t"                                                      \
      "v_or3_b32 %[u1], %[u1], %[h1], %[fB1]\n\t"                                                      \
      "v_or3_b32 %[u2], %[u2], %[h2], %[fB2]\n\t"                                                      \
      "v_or3_b32 %[u3], %[u3], %[h3], %[fB3]\n\t"                                                      \
      "v_alignbit_b32 %[h0], %[fB1], %[fB0], 1\n\t"                                                    \
      "v_alignbit_b32 %[h1], %[fB2], %[fB1], 1\n\t"                                                    \
      "v_alignbit_b32 %[h2], %[fB3], %[fB2], 1\n\t"                                                    \
      "v_alignbit_b32 %[h3], %[rB], %[fB3], 1\n\t"                                                     \
      "v_or_b32_dpp %[v0], %[fA0], %[v0] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
      "v_or_b32_dpp %[v1], %[fA1], %[v1] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
      "v_or_b32_dpp %[v2], %[fA2], %[v2] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
      "v_or_b32_dpp %[v3], %[fA3], %[v3] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
      "v_or3_b32 %[v0], %[v0], %[h0], %[fA0]\n\t"                                                      \
      "v_or3_b32 %[v1], %[v1], %[h1], %[fA1]\n\t"                                                      \
      "v_or3_b32 %[v2], %[v2], %[h2], %[fA2]\n\t"                                                      \
      "v_or3_b32 %[v3], %[v3], %[h3], %[fA3]\n\t"                                                      \
      "v_bitop3_b32 %[fA0], %[u0], %[bA0], %[u0] bitop3:0x30\n\t"                                      \
      "v_bitop3_b32 %[fA1], %[u1], %[bA1], %[u1] bitop3:0x30\n\t"                                      \
      "v_bitop3_b32 %[fA2], %[u2], %[bA2], %[u2] bitop3:0x30\n\t"                                      \
      "v_bitop3_b32 %[fA3], %[u3], %[bA3], %[u3] bitop3:0x30\n\t"                                      \
      "v_bitop3_b32 %[fB0], %[v0], %[bB0], %[v0] bitop3:0x30\n\t"                                      \
      "v_bitop3_b32 %[fB1], %[v1], %[bB1], %[v1] bitop3:0x30\n\t"                                      \
      "v_bitop3_b32 %[fB2], %[v2], %[bB2], %[v2] bitop3:0x30\n\t"                                      \
      "v_bitop3_b32 %[fB3], %[v3], %[bB3], %[v3] bitop3:0x30\n\t"                                      \
      "v_or_b32_e32 %[bA0], %[bA0], %[u0]\n\t"                                                         \
      "v_or_b32_e32 %[bA1], %[bA1], %[u1]\n\t"                                                         \
      "v_or_b32_e32 %[bA2], %[bA2], %[u2]\n\t"                                                         \
      "v_or_b32_e32 %[bA3], %[bA3], %[u3]\n\t"                                                         \
      "v_or_b32_e32 %[bB0], %[bB0], %[v0]\n\t"                                                         \
      "v_or_b32_e32 %[bB1], %[bB1], %[v1]\n\t"                                                         \
      "v_or_b32_e32 %[bB2], %[bB2], %[v2]\n\t"                                                         \
      "v_or_b32_e32 %[bB3], %[bB3], %[v3]\n\t"                                                         \
      "v_or_b32_e32 %[h0], %[fA0], %[fB0]\n\t"                                                         \
      "v_or_b32_e32 %[h3], %[fA3], %[fB3]\n\t"                                                         \
      "v_or3_b32 %[h1], %[fA1], %[fA2], %[fB1]\n\t"                                                    \
      "v_or3_b32 %[h2], %[h0], %[h3], %[fB2]\n\t"                                                      \
      "v_or_b32_e32 %[h1], %[h1], %[h2]\n\t"                                                           \
      "v_cmp_ne_u32_e32 vcc, 0, %[h1]\n\t"                                                             \
      "s_cbranch_vccz 1f\n\t"                                                                          \
      "s_bitset1_b32 %[nz], %[g]\n\t"                                                                  \
      "v_cmp_ne_u32_e32 vcc, 0, %[h0]\n\t"                                                             \
      "s_nop 0\n\t"                                                                                    \
      "s_cmp_lg_u64 vcc, 0\n\t"                                                                        \
      "s_cselect_b32 %[st], 1, 0\n\t"                                                                  \
      "s_lshl_b32 %[st], %[st], %[g]\n\t"                                                              \
      "s_or_b32 %[lo], %[lo], %[st]\n\t"                                                               \
      "v_cmp_ne_u32_e32 vcc, 0, %[h3]\n\t"                                                             \
      "s_nop 0\n\t"                                                                                    \
      "s_cmp_lg_u64 vcc, 0\n\t"                                                                        \
      "s_cselect_b32 %[st], 1, 0\n\t"                                                                  \
      "s_lshl_b32 %[st], %[st], %[g]\n\t"                                                              \
      "s_or_b32 %[hi], %[hi], %[st]\n\t"                                                               \
      "1:\n\t"                                                                                         \
      : [bA0] "+v"(bA[0]), [bA1] "+v"(bA[1]), [bA2] "+v"(bA[2]), [bA3] "+v"(bA[3]), [bB0] "+v"(bB[0]), [bB1] "+v"(bB[1]), [bB2] "+v"(bB[2]),  \
        [bB3] "+v"(bB[3]), [fA0] "+v"(fA[0]), [fA1] "+v"(fA[1]), [fA2] "+v"(fA[2]), [fA3] "+v"(fA[3]), [fB0] "+v"(fB[0]), [fB1] "+v"(fB[1]),   \
        [fB2] "+v"(fB[2]), [fB3] "+v"(fB[3]), [pA] "+v"(pA), [pB] "+v"(pB), [nz] "+s"(nz), [lo] "+s"(lo), [hi] "+s"(hi), [u0] "=&v"(u0),       \
        [u1] "=&v"(u1), [u2] "=&v"(u2), [u3] "=&v"(u3), [v0] "=&v"(v0), [v1] "=&v"(v1), [v2] "=&v"(v2), [v3] "=&v"(v3), [h0] "=&v"(h0),        \
        [h1] "=&v"(h1), [h2] "=&v"(h2), [h3] "=&v"(h3), [st] "=&s"(st)                                                                         \
      : [aw] "s"(aw), [g] "n"(G), [gp] "n"(G > 0 ? G - 1 : 0), [rA] "v"(rA), [rB] "v"(rB)                                                      \
      : "vcc", "scc")
  if constexpr (G == 0) NAVGPU_ROWS2_BODY(NAVGPU_ROWS2_HEAD_FIRST);
  else NAVGPU_ROWS2_BODY(NAVGPU_ROWS2_HEAD_NEXT);
#undef NAVGPU_ROWS2_BODY
#undef NAVGPU_ROWS2_HEAD_FIRST
#undef NAVGPU_ROWS2_HEAD_NEXT
}
// the NG groups of a level, first to last (a compile-time recursion: the group number is an immediate of the asm block)
template <int G>
__device__ __forceinline__ void rows2Level(const uint32_t aw, uint32_t& nz, uint32_t& lo, uint32_t& hi, uint32_t* __restrict__ blA, uint32_t* __restrict__ blB,
                                           uint32_t* __restrict__ frA, uint32_t* __restrict__ frB, uint32_t& pA, uint32_t& pB, const uint32_t zero) {
  constexpr int NG = kRows2Words / 4;
  if constexpr (G < NG) {
    rows2Group<G>(aw, nz, lo, hi, blA + 4 * G, blB + 4 * G, frA + 4 * G, frB + 4 * G, pA, pB, G + 1 < NG ? frA[G + 1 < NG ? 4 * G + 4 : 0] : zero,
                  G + 1 < NG ? frB[G + 1 < NG ? 4 * G + 4 : 0] : zero);
    rows2Level<G + 1>(aw, nz, lo, hi, blA, blB, frA, frB, pA, pB, zero);
  }
}
__device__ __forceinline__ void bfsRows2Grid(const PlannerDev& pl, const uint32_t inst, const int which, const uint32_t item, uint32_t* seedw) {
  constexpr int W = kRows2Words, NG = W / 4, D = kRows2Levels, HL = kRows2HaloLanes;
  int bx0 = 0, bx1 = -1, by0 = 0, by1 = -1, care_ok = 0;  // the robot's region (box + 2 cells) and whether its pockets are known
  if (pl.bfs_bounded) {
    const int4 bb = reinterpret_cast<const int4*>(pl.bfs_box)[2 * inst];
    bx0 = __builtin_amdgcn_readfirstlane(bb.x);
    bx1 = __builtin_amdgcn_readfirstlane(bb.y);
    by0 = __builtin_amdgcn_readfirstlane(bb.z);
    by1 = __builtin_amdgcn_readfirstlane(bb.w);
    care_ok = __builtin_amdgcn_readfirstlane(pl.bfs_box[8 * inst + 4]);
  }
  if (!(bx1 >= bx0 && by1 >= by0)) {  // (uniform over the workgroup) a whole-grid search: the region is the map
    bx0 = 0;
    by0 = 0;
    bx1 = (int)pl.nx - 1;
    by1 = (int)pl.ny - 1;
    care_ok = 0;
  }
  extern __shared__ __align__(16) uint32_t sm[];
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_flag[3];
  __shared__ uint32_t s_open[3];
  uint32_t tid_ = threadIdx.x, nx_ = pl.nx, ny_ = pl.ny;
  asm volatile("" : "+v"(tid_), "+s"(nx_), "+s"(ny_));  // opaque per item, as in bfsWaveGrid
  const uint32_t tid = tid_;
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8] = wall_clock64();
  const Geom g = geomOf(pl, inst);
  const uint32_t nx = nx_, ny = ny_, Wr = (nx + 31) >> 5;  // Wr <= 32 words really exist
  const uint32_t nw = blockDim.x >> 6;
  const uint32_t lane = tid & 63u, wave_id = tid >> 6;
  // lanes HL .. 63 - HL own rows wave * 112 + 2 (lane - HL) and the one after; the HL lanes on either side copy the neighbours'
  const int rowA_i = (int)(wave_id * kRows2PerWave) + 2 * ((int)lane - HL), rowB_i = rowA_i + 1;
  const bool own_lane = lane >= (uint32_t)HL && lane < 64u - HL;
  const bool realA = rowA_i >= 0 && rowA_i < (int)ny, realB = rowB_i >= 0 && rowB_i < (int)ny;
  const bool ownerA = realA && own_lane, ownerB = realB && own_lane;
  const uint32_t rowA = realA ? (uint32_t)rowA_i : 0u, rowB = realB ? (uint32_t)rowB_i : 0u;
  uint32_t* care_l = sm;                             // [kCareRows][kCareWords]
  uint32_t* edge = care_l + kCareRows * kCareWords;  // [nw + 2][top | bottom][HL lanes][A frontier | A blocked | B frontier | B blocked][W]; slot = wave + 1
  const uint32_t edge_words = (nw + 2) * 2 * HL * 4 * W;
  const uint8_t* master = pl.master + (size_t)inst * pl.cells_padded;
  const uint32_t* freew = pl.bfs_free + (size_t)inst * ny * Wr;
  uint32_t* dist = (which == 0 ? pl.path : (which == 1 ? pl.goal : pl.goal_front)) + (size_t)inst * pl.cells;
  const uint32_t N_obst = pl.cells, N_unreach = pl.cells + 1;
  const uint32_t last_mask = (nx & 31) ? ((1u << (nx & 31)) - 1u) : 0xFFFFFFFFu;
  const bool aligned4 = (nx & 3) == 0;

  for (uint32_t i = tid; i < kCareRows * kCareWords + edge_words; i += blockDim.x) sm[i] = 0;
  for (uint32_t i = tid; i < ny * Wr; i += blockDim.x) seedw[i] = 0;
  if (tid < 3) s_flag[tid] = s_open[tid] = 0;
  __syncthreads();
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 4] = wall_clock64();
  if (care_ok) {  // the pocket mask of the robot's box: by region row, four words from the region's first (k_samples)
    const uint32_t* care = pl.bfs_care + (size_t)inst * kCareRows * kCareWords;
    for (uint32_t i = tid; i < (uint32_t)(kCareRows * kCareWords); i += blockDim.x) care_l[i] = care[i];
  }
  rowsPlanSeeds(pl, inst, which, g, master, nx, tid, s_wave, [&](uint32_t mx, uint32_t my) {
    atomicOr(&seedw[my * Wr + (mx >> 5)], 1u << (mx & 31));
  });
  __syncthreads();
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 5] = wall_clock64();
  // The seed words were zeroed and set by other lanes of this workgroup and read by an earlier item: drop this CU's stale L1
  // lines, then plain loads see what the atomics left in L2.
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  auto colMask = [&](int j) { return (uint32_t)j + 1 == Wr ? last_mask : 0xFFFFFFFFu; };
  // words 4q .. 4q + 3 of a row of a [ny][Wr] bitmap: one 16-byte load where rows are whole 16-byte units (1000 cells: 32
  // words).  A lane's two rows are 256 contiguous bytes; word by word a wave's load touched 64 cache lines for 256 bytes
  // and the 128 loads of a lane took 150 us per search (tools/trace_bfs_configs4.py)
  const bool rows16 = (Wr & 3u) == 0;
  auto rowWords = [&](const uint32_t* base, uint32_t row, int q) -> uint4 {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (rows16) {
      if ((uint32_t)(4 * q) < Wr) v = *reinterpret_cast<const uint4*>(base + (size_t)row * Wr + 4 * q);
    } else {
      const uint32_t* r = base + (size_t)row * Wr;
      if ((uint32_t)(4 * q) < Wr) v.x = r[4 * q];
      if ((uint32_t)(4 * q + 1) < Wr) v.y = r[4 * q + 1];
      if ((uint32_t)(4 * q + 2) < Wr) v.z = r[4 * q + 2];
      if ((uint32_t)(4 * q + 3) < Wr) v.w = r[4 * q + 3];
    }
    return v;
  };

  uint32_t blA[W], frA[W], blB[W], frB[W];
#pragma unroll
  for (int q = 0; q < NG; ++q) {
    // rows beyond the grid and words beyond the row never produce cells; halo lanes hold exact copies of the neighbours' rows;
    // seeds expand whatever their cost (map_grid.cpp:160-187)
    const uint4 sA = realA ? rowWords(seedw, rowA, q) : make_uint4(0, 0, 0, 0), sB = realB ? rowWords(seedw, rowB, q) : make_uint4(0, 0, 0, 0);
    const uint4 fA = realA ? rowWords(freew, rowA, q) : make_uint4(0, 0, 0, 0), fB = realB ? rowWords(freew, rowB, q) : make_uint4(0, 0, 0, 0);
    const uint32_t sa[4] = {sA.x, sA.y, sA.z, sA.w}, sb[4] = {sB.x, sB.y, sB.z, sB.w}, fa[4] = {fA.x, fA.y, fA.z, fA.w}, fb[4] = {fB.x, fB.y, fB.z, fB.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = 4 * q + c;
      const bool in = (uint32_t)j < Wr;
      // (a real move: as plain copies the frontier words stay tied to the 4-register tuples of the loads for the whole sweep,
      // and the allocator spills whole tuples inside the level loop)
      asm volatile("v_mov_b32 %0, %1" : "=v"(frA[j]) : "v"(sa[c]));
      asm volatile("v_mov_b32 %0, %1" : "=v"(frB[j]) : "v"(sb[c]));
      blA[j] = (in && realA) ? (~(fa[c] & colMask(j)) | sa[c]) : 0xFFFFFFFFu;
      blB[j] = (in && realB) ? (~(fb[c] & colMask(j)) | sb[c]) : 0xFFFFFFFFu;
    }
  }
  const bool wave_in_box = (int)(wave_id * kRows2PerWave) <= by1 && (int)((wave_id + 1) * kRows2PerWave) > by0;  // wave-uniform
  const bool inA = ownerA && rowA_i >= by0 && rowA_i <= by1, inB = ownerB && rowB_i >= by0 && rowB_i <= by1;
  const int w0 = bx0 >> 5, w1 = bx1 >> 5;
  uint32_t region_groups = 0;
  for (int jr = w0; jr <= w1; ++jr) region_groups |= 1u << (jr >> 2);
  auto storeCells = [&](const uint32_t row, int j, uint32_t cells, uint32_t value) {  // as bfsRowsGrid's (the row's address is worked out here: registers)
    uint32_t* dw = dist + (size_t)row * nx + j * 32;
    if (aligned4) {
      uint32_t full = cells & (cells >> 1) & (cells >> 2) & (cells >> 3) & 0x11111111u;
      cells &= ~(full * 15u);
      const uint4 v4 = make_uint4(value, value, value, value);
      while (full) {
        const uint32_t bpos = (uint32_t)__ffs(full) - 1u;
        *reinterpret_cast<uint4*>(dw + bpos) = v4;
        full &= full - 1;
      }
    }
    while (cells) {
      const uint32_t bpos = (uint32_t)__ffs(cells) - 1u;
      dw[bpos] = value;
      cells &= cells - 1;
    }
  };
  // LDS offsets (words) of the 4 * W-word record (both rows) this lane publishes / takes in at an exchange:
  //   lanes HL .. 2 HL - 1 -> this wave's TOP record, read by the wave above into its lanes 64 - HL .. 63;
  //   lanes 64 - 2 HL .. 63 - HL -> BOTTOM record, read by the wave below into its lanes 0 .. HL - 1
  const bool pub_top = lane >= (uint32_t)HL && lane < 2u * HL, pub_bot = lane >= 64u - 2 * HL && lane < 64u - HL;
  const uint32_t pub_wr = (((wave_id + 1) * 2 + (pub_top ? 0u : 1u)) * HL + (pub_top ? lane - HL : lane - (64u - 2 * HL))) * 4 * W;
  const bool halo_top = lane < (uint32_t)HL, halo_bot = lane >= 64u - HL;
  const uint32_t halo_rd = ((halo_top ? (wave_id * 2 + 1) : ((wave_id + 2) * 2)) * HL + (halo_top ? lane : lane - (64u - HL))) * 4 * W;
  constexpr uint32_t gmask = (1u << NG) - 1u;

  auto activity = [&]() -> uint32_t {
    uint32_t nz = 0, lo = 0, hi = 0;
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const uint32_t t = frA[4 * q] | frA[4 * q + 1] | frA[4 * q + 2] | frA[4 * q + 3] | frB[4 * q] | frB[4 * q + 1] | frB[4 * q + 2] | frB[4 * q + 3];
      if (__builtin_amdgcn_ballot_w64(t != 0) != 0) {
        nz |= 1u << q;
        if (__builtin_amdgcn_ballot_w64((frA[4 * q] | frB[4 * q]) != 0) != 0) lo |= 1u << q;
        if (__builtin_amdgcn_ballot_w64((frA[4 * q + 3] | frB[4 * q + 3]) != 0) != 0) hi |= 1u << q;
      }
    }
    return (nz | (lo >> 1) | (hi << 1)) & gmask;
  };
  if (wave_in_box) {
#pragma unroll
    for (int j = 0; j < W; ++j)
      if (j >= w0 && j <= w1) {
        if (inA) storeCells(rowA, j, frA[j], 0u);  // the seeds: distance 0
        if (inB) storeCells(rowB, j, frB[j], 0u);
      }
  }
  uint32_t a_own = activity();
  uint32_t level = 0, xch = 0, any_blk = 0;
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 2] = wall_clock64();
  bool done = false;
#ifdef NAVGPU_BFS_STATS
  unsigned long long bst[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  while (!done) {
    // ---- D levels on the wave's own: registers and DPP only
    for (int k = 0; k < D; ++k) {
      BFS_STAMP(ts1);
      BFS_ACC(5, 1);
      const uint32_t aw = __builtin_amdgcn_readfirstlane(a_own);
      BFS_ACC(7, aw != 0 ? 1 : 0);
      BFS_ACC(8, __builtin_popcount(aw));
      if (aw != 0) {
        uint32_t nz = 0, lo = 0, hi = 0;
        uint32_t pA = 0, pB = 0;  // the old last words of the previous live group
        const uint32_t zero = 0;
        rows2Level<0>(aw, nz, lo, hi, blA, blB, frA, frB, pA, pB, zero);
        BFS_STAMP(ts2);
        BFS_ACC(1, ts2 - ts1);
        // the new cells of the robot's region get their distance now, from the lane that owns the row
        if (wave_in_box && (nz & region_groups) != 0) {
#pragma unroll
          for (int q = 0; q < NG; ++q) {
            if (((nz & region_groups) >> q) & 1u) {  // wave-uniform
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const int j = 4 * q + c;
                if (j >= w0 && j <= w1) {
                  if (inA && frA[j] != 0) storeCells(rowA, j, frA[j], level + 1);
                  if (inB && frB[j] != 0) storeCells(rowB, j, frB[j], level + 1);
                }
              }
            }
          }
        }
        BFS_STAMP(ts3);
        BFS_ACC(2, ts3 - ts2);
        any_blk |= nz;
        a_own = (nz | (lo >> 1) | (hi << 1)) & gmask;
      }
      ++level;
    }
    // ---- exchange: the outer D own rows go to the neighbours, theirs come into the halo lanes; stop flags
    BFS_STAMP(ts4);
    const uint32_t slot = xch % 3u;
    // (the words pass through real moves on their way to and from the 16-byte LDS accesses: tied to those 4-register tuples
    // the allocator keeps the state in tuples for the whole sweep and spills them inside the level loop)
    auto quad = [&](const uint32_t* w4) {
      uint32_t a, b, c, d;
      asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7" : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(w4[0]), "v"(w4[1]), "v"(w4[2]), "v"(w4[3]));
      return make_uint4(a, b, c, d);
    };
    auto unquad = [&](const uint4 v, uint32_t* w4) {
      asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7" : "=&v"(w4[0]), "=&v"(w4[1]), "=&v"(w4[2]), "=&v"(w4[3]) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
    };
    if (pub_top || pub_bot) {
#pragma unroll
      for (int q = 0; q < W; q += 4) {
        *reinterpret_cast<uint4*>(edge + pub_wr + q) = quad(frA + q);
        *reinterpret_cast<uint4*>(edge + pub_wr + W + q) = quad(blA + q);
        *reinterpret_cast<uint4*>(edge + pub_wr + 2 * W + q) = quad(frB + q);
        *reinterpret_cast<uint4*>(edge + pub_wr + 3 * W + q) = quad(blB + q);
      }
    }
    if (any_blk) s_flag[slot] = 1;
    if (wave_in_box) {  // wave-uniform: is anything of the robot's box still open, or a frontier cell inside the region?
      uint32_t open_any = 0;
      const uint32_t rrA = (uint32_t)(rowA_i - by0), rrB = (uint32_t)(rowB_i - by0);
#pragma unroll
      for (int j = 0; j < W; ++j) {
        if (j >= w0 && j <= w1) {  // wave-uniform
          const uint32_t cw_i = (uint32_t)(j - w0);
          const int c_lo = max(bx0 - j * 32, 0), c_hi = min(bx1 - j * 32, 31);
          const uint32_t cm = c_hi >= c_lo ? ((0xFFFFFFFFu >> (31 - c_hi)) & (0xFFFFFFFFu << c_lo)) : 0u;
          if (inA) {
            const uint32_t care = (care_ok != 0 && cw_i < (uint32_t)kCareWords && rrA < (uint32_t)kCareRows) ? care_l[rrA * kCareWords + cw_i] : (care_ok ? 0u : 0xFFFFFFFFu);
            open_any |= ((~blA[j] & care) | frA[j]) & cm;
          }
          if (inB) {
            const uint32_t care = (care_ok != 0 && cw_i < (uint32_t)kCareWords && rrB < (uint32_t)kCareRows) ? care_l[rrB * kCareWords + cw_i] : (care_ok ? 0u : 0xFFFFFFFFu);
            open_any |= ((~blB[j] & care) | frB[j]) & cm;
          }
        }
      }
      if (open_any != 0) s_open[slot] = 1;
    }
    if (tid == 0) {
      s_flag[(xch + 1) % 3u] = 0;
      s_open[(xch + 1) % 3u] = 0;
    }
    BFS_STAMP(ts5);
    BFS_ACC(3, ts5 - ts4);
    __syncthreads();
    done = !s_flag[slot] || !s_open[slot];  // nothing new in D levels, or nothing open in the box: the search is over
    if (!done) {
      if (halo_top || halo_bot) {
#pragma unroll
        for (int q = 0; q < W; q += 4) {
          if (realA) {
            unquad(*reinterpret_cast<const uint4*>(edge + halo_rd + q), frA + q);
            unquad(*reinterpret_cast<const uint4*>(edge + halo_rd + W + q), blA + q);
          }
          if (realB) {
            unquad(*reinterpret_cast<const uint4*>(edge + halo_rd + 2 * W + q), frB + q);
            unquad(*reinterpret_cast<const uint4*>(edge + halo_rd + 3 * W + q), blB + q);
          }
        }
      }
      a_own = activity();
      __syncthreads();  // the records are free for the next exchange
    }
    ++xch;
    any_blk = 0;
    BFS_STAMP(ts6);
    BFS_ACC(4, ts6 - ts5);
  }
#ifdef NAVGPU_BFS_STATS
  if (lane == 0)
    for (int k = 0; k < 9; ++k) atomicAdd(&g_bfs_stats[k], bst[k]);
#endif
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 3] = wall_clock64();

  // --- the rest of the region: obstacle cells an expanded cell touched -> obstacleCosts(), everything else that was not
  // reached -> unreachableCellCosts().  Expanded = reached free cells + seeds.  (The own rows have moved on since the last
  // exchange, so the rows next to a wave's block are exchanged once more.)
  __syncthreads();
  {
    // (in place: blocked -> expanded)
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const uint4 sA = ownerA ? rowWords(seedw, rowA, q) : make_uint4(0, 0, 0, 0), sB = ownerB ? rowWords(seedw, rowB, q) : make_uint4(0, 0, 0, 0);
      const uint4 fA = ownerA ? rowWords(freew, rowA, q) : make_uint4(0, 0, 0, 0), fB = ownerB ? rowWords(freew, rowB, q) : make_uint4(0, 0, 0, 0);
      const uint32_t sa[4] = {sA.x, sA.y, sA.z, sA.w}, sb[4] = {sB.x, sB.y, sB.z, sB.w}, fa[4] = {fA.x, fA.y, fA.z, fA.w}, fb[4] = {fB.x, fB.y, fB.z, fB.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int j = 4 * q + c;
        const bool in = (uint32_t)j < Wr;
        blA[j] = (in && ownerA) ? ((blA[j] & fa[c] & colMask(j)) | sa[c]) : 0u;
        blB[j] = (in && ownerB) ? ((blB[j] & fb[c] & colMask(j)) | sb[c]) : 0u;
      }
    }
    // only the row next to the neighbouring wave's block is needed: row B of its last own lane / row A of its first
    if (lane == (uint32_t)HL || lane == 63u - HL) {
      const bool top = lane == (uint32_t)HL;
      uint32_t* rec = edge + ((wave_id + 1) * 2 + (top ? 0u : 1u)) * HL * 4 * W;
#pragma unroll
      for (int j = 0; j < W; ++j) rec[j] = top ? blA[j] : blB[j];
    }
    __syncthreads();
    if (lane == (uint32_t)HL - 1u && realB) {  // the last halo lane above: its row B is the row over this wave's first
      const uint32_t* rec = edge + (wave_id * 2 + 1) * HL * 4 * W;
#pragma unroll
      for (int j = 0; j < W; ++j) blB[j] = rec[j];
    }
    if (lane == 64u - HL && realA) {  // the first halo lane below: its row A is the row under this wave's last
      const uint32_t* rec = edge + ((wave_id + 2) * 2) * HL * 4 * W;
#pragma unroll
      for (int j = 0; j < W; ++j) blA[j] = rec[j];
    }
    if (wave_in_box) {
#pragma unroll
      for (int j = 0; j < W; ++j) {
        if (j >= w0 && j <= w1) {  // wave-uniform
          const uint32_t cm = (uint32_t)j + 1 == Wr ? last_mask : ((uint32_t)j < Wr ? 0xFFFFFFFFu : 0u);
          const uint32_t eA = blA[j], eB = blB[j];
          const uint32_t lA = j > 0 ? blA[j > 0 ? j - 1 : 0] : 0u, rA = j + 1 < W ? blA[j + 1 < W ? j + 1 : 0] : 0u;
          const uint32_t lB = j > 0 ? blB[j > 0 ? j - 1 : 0] : 0u, rB = j + 1 < W ? blB[j + 1 < W ? j + 1 : 0] : 0u;
          const uint32_t upA = fromLaneBelow(eB), dnB = fromLaneAbove(eA);
          if (inA) {
            const uint32_t nbr = __builtin_amdgcn_alignbit(eA, lA, 31) | __builtin_amdgcn_alignbit(rA, eA, 1) | upA | eB;
            const uint32_t fb = freew[rowA * Wr + j] & cm;
            const uint32_t touched = nbr & ~fb & ~eA & cm;
            storeCells(rowA, j, touched, N_obst);
            storeCells(rowA, j, ~eA & ~touched & cm, N_unreach);
          }
          if (inB) {
            const uint32_t nbr = __builtin_amdgcn_alignbit(eB, lB, 31) | __builtin_amdgcn_alignbit(rB, eB, 1) | eA | dnB;
            const uint32_t fb = freew[rowB * Wr + j] & cm;
            const uint32_t touched = nbr & ~fb & ~eB & cm;
            storeCells(rowB, j, touched, N_obst);
            storeCells(rowB, j, ~eB & ~touched & cm, N_unreach);
          }
        }
      }
    }
  }
  if (pl.bfs_trace && tid == 0) {
    pl.bfs_trace[(size_t)item * 8 + 6] = wall_clock64();
    pl.bfs_trace[(size_t)item * 8 + 1] = wall_clock64() | ((unsigned long long)level << 48);
  }
  if (tid == 0) pl.bfs_levels[(size_t)inst * 3 + which] = level;  // next cycle's dispatch order
}
__global__ __launch_bounds__(768, 3) void k_bfs_rows2(PlannerDev pl, uint32_t first, uint32_t count, uint32_t* next_item, const uint32_t* order, uint32_t* scratch) {
  __shared__ uint32_t s_item;
  const uint32_t total = count * 3u;
  uint32_t* seedw = scratch + (size_t)blockIdx.x * pl.ny * ((pl.nx + 31) >> 5);  // this workgroup's seed bitmap
  for (;;) {
    if (threadIdx.x == 0) s_item = atomicAdd(next_item, 1u);
    __syncthreads();
    const uint32_t slot = s_item;
    if (slot >= total) break;  // (every workgroup gets here: the counter only grows)
    const uint32_t item = order ? order[slot] : slot;  // longest searches first
    const uint32_t g = item / count;
    bfsRows2Grid(pl, first + (item - g * count), 2 - (int)g, item, seedw);
    __syncthreads();
  }
}
static bool bfs_rows2_fits(uint32_t nx, uint32_t ny) {
  static const bool off = getenv("NAVGPU_DEBUG_BFS_NO_ROWS") != nullptr;  // A/B timing only (then: k_bfs_global)
  return !off && (nx + 31) / 32 <= (uint32_t)kRows2Words && bfs_rows2_waves(ny) <= 12;  // 12 waves of 168 registers
}
// words per row the register-resident row sweep is instantiated for; 0 = not this map's kernel
static int bfs_rows_words(uint32_t nx, uint32_t ny) {
  static const bool off = getenv("NAVGPU_DEBUG_BFS_NO_ROWS") != nullptr;  // A/B timing only
  const uint32_t Wr = (nx + 31) / 32;
  if (off || bfs_rows_waves(ny) > 16) return 0;
  return Wr <= 7 ? 7 : (Wr <= 13 ? 13 : (Wr <= 20 ? 20 : 0));
}

// k_bfs_wave applies when all strips fit the 16 waves of one workgroup
static bool bfs_wave_fits(uint32_t nx, uint32_t ny, int rpt) {
  const uint32_t W = (nx + 31) / 32;
  if (W + 1 > 64) return false;
  const uint32_t spw = 64u / (W + 1), strips = (ny + rpt - 1) / rpt;
  return strips <= 16u * spw;
}
static size_t bfs_wave_lds(uint32_t nx, uint32_t ny, int rpt) {
  const uint32_t W = (nx + 31) / 32, strips = (ny + rpt - 1) / rpt;
  return ((size_t)2 * strips * rpt * W + (size_t)2 * (strips + 2) * 2 * W) * 4;
}

// ------------------------------------------------------------------------------------------------
// k_bfs_global: maps too large for the register / LDS resident kernels (beyond ~640 x 624, e.g. 1000x1000):
// level-synchronous bit-parallel wavefront with the four bitmaps in a global scratch buffer (4 x words x
// 4 B per grid) and direct distance stores, one workgroup per grid.  Levels are activity-driven: only
// the 128 x 16-cell tiles that hold or border new frontier cells are expanded (see below), the words of
// the next tile are fetched while the current one is processed.  1000 x 1000: 11.8 ms per wavefront
// (35 ms for the dense sweep it replaces).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kMaxTiles = 8192;  // 128 x 16-cell tiles of the largest map k_bfs_global accepts (32 KB of flags)
__global__ __launch_bounds__(1024) void k_bfs_global(PlannerDev pl, uint32_t first, uint32_t* scratch) {
  __shared__ uint32_t s_wave[16];
  __shared__ uint8_t s_act[4 * kMaxTiles];
  const int which = (int)pl.bfs_grids - 1 - (int)blockIdx.y;  // longest searches (goal grids) are dispatched first
  const uint32_t inst = first + blockIdx.x;
  const uint32_t tid = threadIdx.x;
  const Geom g = geomOf(pl, inst);
  const uint32_t nx = pl.nx, ny = pl.ny, W = (nx + 31) >> 5, words = ny * W;
  uint32_t* base = scratch + ((size_t)(blockIdx.x * 3 + which)) * 4 * words;
  uint32_t* vis = base;
  uint32_t* fre = base + words;
  uint32_t* cur = base + 2 * words;
  uint32_t* nxt = base + 3 * words;
  const uint8_t* master = pl.master + (size_t)inst * pl.cells_padded;
  uint32_t* dist = (which == 0 ? pl.path : (which == 1 ? pl.goal : pl.goal_front)) + (size_t)inst * pl.cells;
  const uint32_t N_obst = pl.cells, N_unreach = pl.cells + 1;
  const uint32_t last_mask = (nx & 31) ? ((1u << (nx & 31)) - 1u) : 0xFFFFFFFFu;
  const uint32_t* freew = pl.bfs_free + (size_t)inst * words;  // k_free_bits (or the extra blocks of k_samples)
  for (uint32_t w = tid; w < words; w += blockDim.x) {
    const uint32_t row = w / W, wi = w - row * W;
    const uint32_t bits = freew[w] & ((wi + 1 == W) ? last_mask : 0xFFFFFFFFu);
    fre[w] = bits | (bfsWithinWord(pl, which, inst, row, W, wi) & ((wi + 1 == W) ? last_mask : 0xFFFFFFFFu));
    vis[w] = (wi + 1 == W) ? ~last_mask : 0u;
    cur[w] = 0;
    nxt[w] = 0;
  }
  __syncthreads();
  {
    const uint32_t n = pl.plan_count[inst];
    const double* P = pl.plan + (size_t)inst * pl.max_plan * 2;
    const bool ovr = which == 2;
    const double lx = pl.front_last[2 * inst], ly = pl.front_last[2 * inst + 1];
    const uint32_t chunk = (n + blockDim.x - 1) / blockDim.x;
    const uint32_t i0 = min(n, tid * chunk), i1 = min(n, i0 + chunk);
    uint32_t mine = 0;
    for (uint32_t i = i0; i < i1; ++i) mine += adjustedPoints(P, i, lx, ly, ovr, n, g.res, true, [](uint32_t, double, double) {});
    uint32_t total;
    const uint32_t bs = blockExclusiveScan1024(mine, s_wave, &total);
    auto valid = [&](double x, double y, uint32_t& cell) {
      uint32_t mx, my;
      if (!worldToMap(g, x, y, mx, my)) return false;
      cell = my * nx + mx;
      return master[cell] != kNoInfo;
    };
    uint32_t fmin_ = 0xFFFFFFFFu, b = bs;
    for (uint32_t i = i0; i < i1; ++i)
      b += adjustedPoints(P, i, lx, ly, ovr, n, g.res, false, [&](uint32_t k, double x, double y) {
        uint32_t cell;
        if (valid(x, y, cell)) fmin_ = min(fmin_, b + k);
      });
    const uint32_t f = blockMin1024(fmin_, s_wave);
    if (f != 0xFFFFFFFFu) {
      uint32_t emin = total;
      b = bs;
      for (uint32_t i = i0; i < i1; ++i)
        b += adjustedPoints(P, i, lx, ly, ovr, n, g.res, false, [&](uint32_t k, double x, double y) {
          uint32_t cell;
          if (b + k > f && !valid(x, y, cell)) emin = min(emin, b + k);
        });
      const uint32_t e = blockMin1024(emin, s_wave);
      b = bs;
      for (uint32_t i = i0; i < i1; ++i)
        b += adjustedPoints(P, i, lx, ly, ovr, n, g.res, false, [&](uint32_t k, double x, double y) {
          const uint32_t idx = b + k;
          const bool seed = (which == 0) ? (idx >= f && idx < e) : (idx == e - 1);
          if (!seed) return;
          uint32_t cell;
          if (!valid(x, y, cell)) return;
          const uint32_t my = cell / nx, mx = cell - my * nx;
          atomicOr(&cur[my * W + (mx >> 5)], 1u << (mx & 31));
          dist[cell] = 0;
        });
    }
  }
  __syncthreads();
  for (uint32_t w = tid; w < words; w += blockDim.x) vis[w] |= cur[w];
  // Activity-driven levels: the map is cut into tiles of 4 words x 16 rows (128 x 16 cells, one wave each; tile
  // t -> wave t % 16).  A tile is expanded at a level only when it, or the tile across one of its edges, reached
  // cells the level before (flags raised by plain LDS stores); a wavefront ring crosses such a tile for ~150
  // of the ~1500 levels of a 1000 x 1000 map.  A tile that is left out must not leave an old frontier behind in
  // the buffer that becomes `cur` next: `dirty` remembers which tiles wrote a non-empty frontier into which buffer.
  const uint32_t tiles_x = (W + 3) >> 2, tiles_y = (ny + 15) >> 4, T = tiles_x * tiles_y;
  uint8_t* act = s_act;                    // [2][kMaxTiles]
  uint8_t* dirty = s_act + 2 * kMaxTiles;  // [2][kMaxTiles]
  for (uint32_t i = tid; i < 4 * kMaxTiles; i += blockDim.x) s_act[i] = 0;
  __syncthreads();
  for (uint32_t i = tid; i < T; i += blockDim.x) {
    act[i] = 1;    // first level: every tile
    dirty[i] = 1;  // `cur` (buffer 0) holds the seeds
  }
  __syncthreads();
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  const uint32_t lr = lane >> 2, lc = lane & 3u;
  uint32_t level = 0, buf = 0;  // buf: which flag set belongs to `cur`
  // bounded search (see k_bfs_wave): the robot's region, its pocket mask, and the words of the bitmaps that cover it
  int gx0 = 0, gx1 = -1, gy0 = 0, gy1 = -1, care_ok = 0;
  if (pl.bfs_bounded && pl.bfs_grids == 3) {
    const int* bb = pl.bfs_box + (size_t)inst * 8;
    gx0 = bb[0];
    gx1 = bb[1];
    gy0 = bb[2];
    gy1 = bb[3];
    care_ok = bb[4];
  }
  const bool bounded = gx1 >= gx0 && gy1 >= gy0;
  const uint32_t rg_w0 = (uint32_t)(gx0 >> 5), rg_nw = bounded ? (uint32_t)(gx1 >> 5) - rg_w0 + 1 : 0u;
  const uint32_t rg_words = bounded ? (uint32_t)(gy1 - gy0 + 1) * rg_nw : 0u;
  const uint32_t* care = pl.bfs_care + (size_t)inst * kCareRows * kCareWords;
  while (true) {
    int any = 0;
    uint8_t* act_cur = act + buf * kMaxTiles;
    uint8_t* act_nxt = act + (buf ^ 1u) * kMaxTiles;
    uint8_t* dirty_nxt = dirty + (buf ^ 1u) * kMaxTiles;
    for (uint32_t t0 = wave; t0 < T; t0 += 16u * 64u) {
      // this wave's next (up to) 64 tiles: lane i looks at tile t0 + 16 i
      const uint32_t ti = t0 + 16u * lane;
      const bool a_ = ti < T && act_cur[ti] != 0;
      const bool d_ = ti < T && dirty_nxt[ti] != 0;
      if (ti < T) act_cur[ti] = 0;  // consumed; raised again by the tiles that reach cells this level
      uint64_t amask = __builtin_amdgcn_ballot_w64(a_);
      uint64_t todo = amask | __builtin_amdgcn_ballot_w64(d_);
      // software pipeline: the loads of the next tile are issued before the current one is expanded
      struct TileIn {
        uint32_t t, ty, tx, row, wi, w;
        uint32_t fc, l, r, u, d, v, fb;
        bool in, expand;
      };
      auto fetch = [&](uint32_t i) {
        TileIn q;
        q.t = t0 + 16u * i;
        q.ty = q.t / tiles_x;
        q.tx = q.t - q.ty * tiles_x;
        q.row = q.ty * 16 + lr;
        q.wi = q.tx * 4 + lc;
        q.in = q.row < ny && q.wi < W;
        q.w = q.row * W + q.wi;
        q.expand = (amask >> i) & 1u;
        q.fc = q.l = q.r = q.u = q.d = q.v = q.fb = 0;
        if (q.in && q.expand) {
          q.fc = cur[q.w];
          q.l = q.wi > 0 ? cur[q.w - 1] : 0u;
          q.r = q.wi + 1 < W ? cur[q.w + 1] : 0u;
          q.u = q.row > 0 ? cur[q.w - W] : 0u;
          q.d = q.row + 1 < ny ? cur[q.w + W] : 0u;
          q.v = vis[q.w];
          q.fb = fre[q.w];
        }
        return q;
      };
      TileIn nextq{};
      if (todo) nextq = fetch((uint32_t)__builtin_ctzll(todo));
      while (todo) {
        todo &= todo - 1;
        const TileIn q = nextq;
        if (todo) nextq = fetch((uint32_t)__builtin_ctzll(todo));
        const uint32_t t = q.t;
        if (!q.expand) {  // not expanded: only wipe the frontier it wrote two levels ago
          if (q.in) nxt[q.w] = 0;
          if (lane == 0) dirty_nxt[t] = 0;
          continue;
        }
        uint32_t nf = 0;
        if (q.in) {
          const uint32_t cand = ((q.fc << 1) | (q.l >> 31) | (q.fc >> 1) | (q.r << 31) | q.u | q.d) & ~q.v;
          nf = cand & q.fb;
          uint32_t no = cand & ~q.fb;
          nxt[q.w] = nf;
          if (cand) {
            vis[q.w] = q.v | cand;
            uint32_t* drow = dist + q.row * nx + q.wi * 32;
            uint32_t qq = nf;
            while (qq) {
              const int bpos = __ffs(qq) - 1;
              qq &= qq - 1;
              drow[bpos] = level + 1;
            }
            while (no) {
              const int bpos = __ffs(no) - 1;
              no &= no - 1;
              drow[bpos] = N_obst;
            }
          }
        }
        const uint64_t nz = __builtin_amdgcn_ballot_w64(nf != 0);
        if (nz != 0) {  // wave-uniform: wake this tile and the tiles across the edges the new cells lie on
          any = 1;
          const bool up = (nz & 0xFull) != 0, down = (nz >> 60) != 0;
          const bool left = __builtin_amdgcn_ballot_w64(lc == 0 && (nf & 1u)) != 0;
          const bool right = __builtin_amdgcn_ballot_w64(lc == 3 && (nf >> 31)) != 0;
          if (lane == 0) {
            act_nxt[t] = 1;
            dirty_nxt[t] = 1;
            if (up && q.ty > 0) act_nxt[t - tiles_x] = 1;
            if (down && q.ty + 1 < tiles_y) act_nxt[t + tiles_x] = 1;
            if (left && q.tx > 0) act_nxt[t - 1] = 1;
            if (right && q.tx + 1 < tiles_x) act_nxt[t + 1] = 1;
          }
        } else if (lane == 0) {
          dirty_nxt[t] = 0;
        }
      }
    }
    if (!__syncthreads_or(any)) break;
    uint32_t* t = cur;
    cur = nxt;
    nxt = t;
    buf ^= 1u;
    ++level;
    if (bounded && (level & 7u) == 0) {  // stop once no cell of the box is open and no frontier cell is in the region
      int open = 0;
      for (uint32_t i = tid; i < rg_words; i += blockDim.x) {
        const uint32_t rr = i / rg_nw, ww = i - rr * rg_nw, wi = rg_w0 + ww, w = ((uint32_t)gy0 + rr) * W + wi;
        const int c_lo = max(gx0 - (int)(wi * 32), 0), c_hi = min(gx1 - (int)(wi * 32), 31);
        const uint32_t cm = (0xFFFFFFFFu >> (31 - c_hi)) & (0xFFFFFFFFu << c_lo);
        const uint32_t cw = care_ok ? (ww < (uint32_t)kCareWords ? care[rr * kCareWords + ww] : 0u) : 0xFFFFFFFFu;
        if ((((~vis[w] & fre[w] & cw) | cur[w]) & cm) != 0) open = 1;
      }
      if (!__syncthreads_or(open)) break;
    }
  }
  if (tid == 0 && pl.bfs_grids == 3) pl.bfs_levels[(size_t)inst * 3 + which] = level;
  // (a bounded search is only ever read inside its region: the rest of the grid is left as it is)
  for (uint32_t i = tid; i < (bounded ? rg_words : words); i += blockDim.x) {
    const uint32_t w = bounded ? ((uint32_t)gy0 + i / rg_nw) * W + rg_w0 + (i - (i / rg_nw) * rg_nw) : i;
    uint32_t t = ~vis[w];
    if (t) {
      const uint32_t row = w / W, wi = w - row * W;
      uint32_t* drow = dist + row * nx + wi * 32;
      while (t) {
        const int bpos = __ffs(t) - 1;
        t &= t - 1;
        drow[bpos] = N_unreach;
      }
    }
  }
}

// rows per thread needed so that ceil(ny/RPT) * W strips fit one 1024-thread workgroup
static int bfs_rows_per_thread(uint32_t nx, uint32_t ny) {
  const uint32_t W = (nx + 31) / 32;
  for (int rpt : {6, 12, 24}) {
    if (((ny + rpt - 1) / rpt) * W <= 1024) return rpt;
  }
  return 0;
}

static int bfs_rows_per_thread(uint32_t nx, uint32_t ny);
size_t bfs_lds_bytes(uint32_t nx, uint32_t ny) {
  const uint32_t W = (nx + 31) / 32;
  const int rpt = bfs_rows_per_thread(nx, ny);
  if (!rpt) return ~(size_t)0;
  const uint32_t strips = (ny + rpt - 1) / rpt;
  return ((size_t)2 * (strips * rpt + 2) * (W + 2) + (size_t)ny * W) * 4;
}

bool bfs_lds_resident(uint32_t nx, uint32_t ny) {
  return bfs_rows_per_thread(nx, ny) != 0 && bfs_lds_bytes(nx, ny) <= 156u * 1024u;
}
size_t bfs_scratch_words(uint32_t nx, uint32_t ny) {  // per instance, for k_bfs_global
  return bfs_lds_resident(nx, ny) ? 0 : (size_t)3 * 4 * ny * ((nx + 31) / 32);
}

static uint32_t bfs_cu_count() {
  static const uint32_t n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return (uint32_t)cus;
  }();
  return n;
}
bool bfs_bounded_applies(const PlannerDev& pl) {
  static const bool force_lds_kernel = getenv("NAVGPU_DEBUG_BFS_LDS") != nullptr;
  if (!bfs_lds_resident(pl.nx, pl.ny)) return true;  // k_bfs_global
  if (force_lds_kernel) return false;
  return bfs_wave_fits(pl.nx, pl.ny, 7) || (bfs_wave_fits(pl.nx, pl.ny, 13) && bfs_wave_lds(pl.nx, pl.ny, 13) <= 156u * 1024u);
}
// persistent DIRECT workgroups per CU (2 fill every wave slot; 1 leaves half of them to another stream's kernels)
static uint32_t bfs_direct_wgs_per_cu() {
  static const uint32_t v = [] {
    const char* e = getenv("NAVGPU_BFS_WGS_PER_CU");
    return e ? (uint32_t)std::max(1, atoi(e)) : 2u;
  }();
  return v;
}
// the bounded searches of a DWA launch on the row sweep; false = not this map's kernel
template <int W>
static void launch_bfs_rows_w(const PlannerDev& pl, uint32_t first, uint32_t count, hipStream_t s, const uint32_t* order) {
  const uint32_t nw = bfs_rows_waves(pl.ny);
  const size_t lds = bfs_rows_lds_words<W>(pl.nx, pl.ny) * 4;
  if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k_bfs_rows<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const uint32_t per_cu = std::max<uint32_t>(1u, std::min<uint32_t>(std::min<uint32_t>(24u / nw, (uint32_t)((156u * 1024u) / (lds + 1024))), 2u * bfs_direct_wgs_per_cu()));
  hipLaunchKernelGGL(k_bfs_rows<W>, dim3(std::min(count * 3u, per_cu * bfs_cu_count())), dim3(nw * 64), lds, s, pl, first, count, pl.bfs_next_item + 1, order);
}
static bool launch_bfs_rows(const PlannerDev& pl, uint32_t first, uint32_t count, hipStream_t s, const uint32_t* order) {
  switch (bfs_rows_words(pl.nx, pl.ny)) {
    case 7: launch_bfs_rows_w<7>(pl, first, count, s, order); return true;
    case 13: launch_bfs_rows_w<13>(pl, first, count, s, order); return true;
    case 20: launch_bfs_rows_w<20>(pl, first, count, s, order); return true;
    default: return false;
  }
}
void launch_bfs(const PlannerDev& pl, uint32_t first, uint32_t count, hipStream_t s, const uint32_t* order, bool free_ready, int n_whole) {
  // n_whole: how many of the robots search their whole grid this cycle (< 0 = unknown / all): a bounded launch runs the
  // DIRECT variant for the bounded searches and, only if there are any, the plane variant for the others
  dim3 grid(count, pl.bfs_grids);
  const size_t lds = bfs_lds_bytes(pl.nx, pl.ny);  // dense bit-parallel sweep (no LDS atomics in the loop)
  const int rpt = bfs_rows_per_thread(pl.nx, pl.ny);
  if (rpt != 0 && lds <= 156u * 1024u) {
#define NAVGPU_BFS(R)                                                                                           \
  {                                                                                                             \
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k_bfs<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(k_bfs<R>, grid, dim3(1024), lds, s, pl, first);                                           \
  }
    static const bool force_lds_kernel = getenv("NAVGPU_DEBUG_BFS_LDS") != nullptr;  // A/B timing only
#define NAVGPU_BFS_WAVE(R, LEG, P)                                                                                            \
  {                                                                                                                           \
    const size_t lds_w = bfs_wave_lds(pl.nx, pl.ny, R);                                                                       \
    if (lds_w > 48 * 1024) hipFuncSetAttribute((const void*)k_bfs_wave<R, LEG, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w); \
    if (!free_ready) hipLaunchKernelGGL(k_free_bits, dim3((pl.ny * ((pl.nx + 31) / 32) + 255) / 256, count), dim3(256), 0, s, pl, first); \
    if (!free_ready) hipMemsetAsync(pl.bfs_next_item, 0, 2 * sizeof(uint32_t), s);  /* (k_samples zeroes them too) */         \
    if constexpr (!LEG) {                                                                                                     \
      if (launch_bfs_rows(pl, first, count, s, order)) return;  /* bounded and whole-grid searches alike */                    \
    }                                                                                                                         \
    const bool direct = !LEG && pl.bfs_bounded && n_whole >= 0 && (uint32_t)n_whole < count;                                  \
    if (!direct || n_whole > 0)                                                                                               \
      hipLaunchKernelGGL((k_bfs_wave<R, LEG, P>), dim3(std::min(count * pl.bfs_grids, bfs_cu_count())), dim3(1024), lds_w, s, pl, first, count, pl.bfs_next_item, order, direct ? 1 : 0); \
    if constexpr (!LEG) {                                                                                                     \
      if (direct) {                                                                                                           \
        if (lds_w > 48 * 1024) hipFuncSetAttribute((const void*)k_bfs_wave<R, false, P, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w); \
        hipLaunchKernelGGL((k_bfs_wave<R, false, P, true>), dim3(std::min(count * pl.bfs_grids, std::min(bfs_direct_wgs_per_cu(), 2 * lds_w <= 156u * 1024u ? 2u : 1u) * bfs_cu_count())), dim3(1024), lds_w, s, pl, first, count, pl.bfs_next_item + 1, order, 0); \
      }                                                                                                                       \
    }                                                                                                                         \
    return;                                                                                                                   \
  }
    if (!force_lds_kernel && bfs_wave_fits(pl.nx, pl.ny, 7)) {
      if (pl.bfs_grids == 2) NAVGPU_BFS_WAVE(7, true, 10)
      else NAVGPU_BFS_WAVE(7, false, 10)
    }
    if (!force_lds_kernel && bfs_wave_fits(pl.nx, pl.ny, 13) && bfs_wave_lds(pl.nx, pl.ny, 13) <= 156u * 1024u) {
      if (pl.bfs_grids == 2) NAVGPU_BFS_WAVE(13, true, 3)
      else NAVGPU_BFS_WAVE(13, false, 3)
    }
#undef NAVGPU_BFS_WAVE
    if (rpt == 6) {
      NAVGPU_BFS(6)
    } else if (rpt == 12) NAVGPU_BFS(12)
    else NAVGPU_BFS(24)
#undef NAVGPU_BFS
    return;
  }
  if (!free_ready) hipLaunchKernelGGL(k_free_bits, dim3((pl.ny * ((pl.nx + 31) / 32) + 255) / 256, count), dim3(256), 0, s, pl, first);
  static const bool force_global_kernel = getenv("NAVGPU_DEBUG_BFS_GLOBAL") != nullptr;  // A/B timing only
  if (pl.bfs_grids == 3 && !force_global_kernel && bfs_rows2_fits(pl.nx, pl.ny)) {  // two rows per lane, one search per CU
    // (the scratch holds 12 bitmaps per robot of the fleet; a workgroup uses one)
    if (!free_ready) hipMemsetAsync(pl.bfs_next_item, 0, 2 * sizeof(uint32_t), s);
    const size_t lds2 = bfs_rows2_lds_words(pl.ny) * 4;
    if (lds2 > 48 * 1024) hipFuncSetAttribute((const void*)k_bfs_rows2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    hipLaunchKernelGGL(k_bfs_rows2, dim3(std::min(count * 3u, bfs_cu_count())), dim3(bfs_rows2_waves(pl.ny) * 64), lds2, s, pl, first, count, pl.bfs_next_item, order, pl.bfs_scratch);
    return;
  }
  hipLaunchKernelGGL(k_bfs_global, grid, dim3(1024), 0, s, pl, first, pl.bfs_scratch);
}

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
struct ScoreOut {
  double total;
  int status;
  int n_points;
};

// TABLES (use_dwa && discretize_by_time): the heading sequence theta_k of a sample depends only on its
// v_theta and the step (theta += v_theta*dt, rounded to float each step), so sincos(theta_k),
// sincos(pi/2+theta_k), the rotated footprint vertices and the forward-point offset are computed once
// per (v_theta sample, step) by the workgroup into LDS and shared by all (vx, vy) samples, and lanes
// are mapped so that a wave shares one v_theta: identical edge shapes => convergent Bresenham loops.
// The arithmetic per value is unchanged (same operations, same rounding), only deduplicated.
__host__ __device__ inline size_t score_bits_bytes(int win) {  // [win][nw][4] words (part of the LDS image), 16-byte aligned
  return (((size_t)4 * win * ((win + 31) >> 5) * 4) + 15) & ~(size_t)15;
}
// the two [win][nw][2] word arrays the dilation passes work in: behind the image, only where the image is BUILT (PREP != 2)
__host__ __device__ inline size_t score_scratch_bytes(int win) { return score_bits_bytes(win); }
// PREP: 0 = build the LDS image (window, bitmaps, tables) in this workgroup; 1 = build it and store it to
// pl.prep (k_score_prep*, one workgroup per robot); 2 = load the stored image (the scoring workgroups of
// a robot all use the same one: 74 of them in the 32x32x16 configuration)
// CHUNK: cells of a footprint edge fetched per LDS round trip; the launcher picks the smallest of 6 / 9 / 12 / 16 that
// covers the longest edge (a 0.4 m square at 0.05 m: 9), longer edges take several chunks
// AGG: the MapGridCostFunction options DWAPlanner itself never sets - aggregation Sum / Product and a sideways shift
// (map_grid_cost_function.cpp:75-129) - as navgpu_planner_set_map_grid_options configures them: every live critic looks
// its own cell up at every point (no screen, no shared cell); the product kernels are compiled without it.
// n / d for 0 <= n < 2^22, 1 <= d < 2^22: float quotient + one correction step either way (the generic 32-bit division is
// ~40 vector instructions, and every lane of a scoring workgroup makes two of them)
__device__ __forceinline__ int divSmall(int n, int d) {
  int q = (int)((float)n * __builtin_amdgcn_rcpf((float)d));
  int r = n - q * d;
  if (r < 0) {
    --q;
    r += d;
  }
  if (r >= d) ++q;
  return q;
}
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
  uint32_t* s_bb = s_ba + 2 * win * nw;
  const int rc = (int)pl.fp_rcells;
  const uint8_t fail_span_w = (pl.cfg.allow_unknown != 0) ? 0 : 1;
  if (PREP != 2) __syncthreads();
#ifdef NAVGPU_SCORE_TIMING
  const unsigned long long ts0a = wall_clock64();
#endif
  if (PREP != 2) {
  for (int it = tid; it < win * nw; it += blockDim.x) {
    const int y = it / nw, j = it - y * nw;
    uint32_t nf = 0, fl = 0;
    for (int b = 0; b < 32; ++b) {
      const int lx = 32 * j + b;
      uint32_t o = 1, f = 1;
      // cells off the MAP count as set like cells outside the window: a footprint vertex there fails
      // worldToMap, i.e. footprintCost = -1 (costmap_model.cpp:77-99), which only the polygon walk reports
      const int gx = wx0 + lx, gy = wy0 + y;
      if (lx < win && gx >= 0 && gy >= 0 && gx < (int)g.nx && gy < (int)g.ny) {
        const uint8_t cc = s_win[y * win + lx];
        o = cc != 0 ? 1u : 0u;
        f = (uint8_t)(cc - kLethal) <= fail_span_w ? 1u : 0u;
      }
      nf |= o << b;
      fl |= f << b;
    }
    if (pl.scale_obstacle == 0) nf = fl = 0;  // obstacle critic off (scale 0: skipped, simple_scored_sampling_planner.cpp:55-57): nothing to screen
    s_ba[2 * it] = nf;
    s_ba[2 * it + 1] = fl;
  }
  __syncthreads();
  for (int it = tid; it < 2 * win * nw; it += blockDim.x) {
    const int cell = it >> 1, f = it & 1;
    const int y = cell / nw, j = cell - y * nw;
    const uint32_t cur = s_ba[it];
    const uint32_t left = j > 0 ? s_ba[it - 2] : 0xFFFFFFFFu, right = j + 1 < nw ? s_ba[it + 2] : 0xFFFFFFFFu;
    uint32_t m = cur;
    if (rc > 31) m = 0xFFFFFFFFu;  // reach beyond the neighbouring words: no shortcut
    for (int d = 1; d <= rc && d < 32; ++d)
      m |= (cur << d) | (left >> (32 - d)) | (cur >> d) | (right << (32 - d));
    (void)f;
    s_bb[it] = m;
  }
  __syncthreads();
  for (int it = tid; it < 2 * win * nw; it += blockDim.x) {
    const int cell = it >> 1;
    const int y = cell / nw;
    uint32_t m = (y - rc < 0 || y + rc >= win) ? 0xFFFFFFFFu : 0u;
    if (!m)
      for (int d = -rc; d <= rc; ++d) m |= s_bb[it + 2 * d * nw];
    s_fb[4 * cell + (it & 1)] = m;
  }
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
#ifdef NAVGPU_SCORE_TIMING
  const unsigned long long ts0b = wall_clock64();
#endif
  if (PREP != 0) {  // the LDS image as 16-byte words: [0, prep_bytes)
    uint4* img = reinterpret_cast<uint4*>(pl.prep + (size_t)inst * pl.prep_stride);
    uint4* lds = reinterpret_cast<uint4*>(s_dyn);
    const uint32_t n16 = pl.prep_bytes >> 4;
    if (PREP == 1) {
      for (uint32_t i = tid; i < n16; i += blockDim.x) img[i] = lds[i];
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
          const float nxp = (float)(px + (lv[0] * cs + lv[1] * cs2) * dt);
          const float nyp = (float)(py + (lv[0] * sn + lv[1] * sn2) * dt);
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
uint32_t launch_score(const PlannerDev& pl_in, uint32_t first, uint32_t count, const float* explicit_sample, hipStream_t s) {
  PlannerDev pl = pl_in;
  // A/B switch for the per-(v_theta, step) tables (tools/probe_score.py); read once, off in product use
  static const bool dbg_no_tables = getenv("NAVGPU_DEBUG_NO_TABLES") && atoi(getenv("NAVGPU_DEBUG_NO_TABLES"));
  if (dbg_no_tables) pl.use_tables = 0;
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
        const float nxp = (float)(px + (lx * s_sc[step][0] + ly * s_sc[step][2]) * dt);
        const float nyp = (float)(py + (lx * s_sc[step][1] + ly * s_sc[step][3]) * dt);
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
        const float nxp = (float)(px + (lv[0] * cs + lv[1] * cs2) * dt);
        const float nyp = (float)(py + (lv[0] * sn + lv[1] * sn2) * dt);
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

#ifdef NAVGPU_BFS_STATS
extern "C" int navgpu_debug_bfs_stats(unsigned long long* out16, int reset) {
  if (out16) hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_bfs_stats), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_bfs_stats), z, sizeof(z));
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
