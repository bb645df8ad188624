// MapGrid wavefronts (gfx950): k_free_bits, k_bfs_global (maps beyond the register-resident sweeps of planner_bfs_rows.hip)
// and launch_bfs, which picks the kernel for a map.
#include "planner_common.h"

namespace navgpu {

// k_free_bits: the traversable-cell bitmap of every robot's costmap, [ny][W] words, once per launch for the two or three
// wavefronts of a robot (each reads its rows twice).  Inside k_bfs_wave the same 160 KB of cost bytes took 14 wide loads
// per lane that the register budget of the sweep serialises: 16 us per read, against 7 dword loads now.
__global__ __launch_bounds__(256) void k_free_bits(PlannerDev pl, uint32_t first) {
  const uint32_t W = (pl.nx + 31) >> 5, words = pl.ny * W;
  const uint32_t inst = first + blockIdx.y;
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= words) return;
  const uint32_t row = i / W, wi = i - row * W;
  const uint32_t fw = bfsFreeWord(pl.master + (size_t)inst * pl.cells_padded, row, pl.nx, wi, pl.cfg.allow_unknown != 0 ? 0u : 1u);
  pl.bfs_free[(size_t)inst * words + i] = fw;
  if (pl.within) pl.within[(size_t)inst * words + i] |= fw;  // legacy planner: path_map_'s bitmap = free | within_robot (bfsFreeBitmap)
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
  const uint32_t* freew = bfsFreeBitmap(pl, which, inst, words);  // k_free_bits (or the extra blocks of k_samples)
  for (uint32_t w = tid; w < words; w += blockDim.x) {
    const uint32_t row = w / W, wi = w - row * W;
    fre[w] = freew[w] & ((wi + 1 == W) ? last_mask : 0xFFFFFFFFu);
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
// per-instance scratch words of the wavefront launches: none for maps the one-row-per-lane sweep takes, else twelve bitmaps
// (k_bfs_global: four per grid; k_bfs_rows2: one seed bitmap per workgroup, at most three workgroups per robot)
size_t bfs_scratch_words(uint32_t nx, uint32_t ny) { return bfs_rows_fits(nx, ny) ? 0 : (size_t)3 * 4 * ny * ((nx + 31) / 32); }

uint32_t bfs_cu_count() {
  static const uint32_t n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return (uint32_t)cus;
  }();
  return n;
}
// Every wavefront kernel stops a bounded search at its robot's region (pl.bfs_bounded) and runs the legacy
// planner's two-grid launches (pl.bfs_grids == 2, pl.within) as well: k_bfs_rows for maps up to 640 x 800, k_bfs_rows2 up to
// 1024 x 1344, k_bfs_global beyond.  order: the launch's items longest first (k_samples), or null.
void launch_bfs(const PlannerDev& pl, uint32_t first, uint32_t count, hipStream_t s, const uint32_t* order, bool free_ready) {
  if (!free_ready) {  // (a planner cycle has both done by its k_samples launch)
    hipLaunchKernelGGL(k_free_bits, dim3((pl.ny * ((pl.nx + 31) / 32) + 255) / 256, count), dim3(256), 0, s, pl, first);
    hipMemsetAsync(pl.bfs_next_item, 0, 2 * sizeof(uint32_t), s);
  }
  if (launch_bfs_rows(pl, first, count, s, order)) return;
  if (launch_bfs_rows2(pl, first, count, s, order)) return;
  hipLaunchKernelGGL(k_bfs_global, dim3(count, pl.bfs_grids), dim3(1024), 0, s, pl, first, pl.bfs_scratch);
}

}  // namespace navgpu
