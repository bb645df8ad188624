// k_samples (gfx950): VelocityIterator + SimpleTrajectoryGenerator::initialise, and the per-robot preparation of the
// MapGrid wavefronts (region, pockets, dispatch ranks, traversable-cell bitmaps) that shares its launch.
#include "planner_common.h"

namespace navgpu {

// ------------------------------------------------------------------------------------------------
// k_samples: base_local_planner/include/base_local_planner/velocity_iterator.h:49-74 and
// SimpleTrajectoryGenerator::initialise (src/simple_trajectory_generator.cpp:60-135).
// One lane per axis: `next += step_size` is a sequential fp64 accumulation and must stay one.
// ------------------------------------------------------------------------------------------------
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

}  // namespace navgpu
