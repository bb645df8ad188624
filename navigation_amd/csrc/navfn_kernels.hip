// navfn::NavFn on the device: the potential-field expansion behind the navfn / global_planner global
// planners and the gradient-descent path extraction, for a batch of independent plans.
//   NavFn::setCostmap (navfn/src/navfn.cpp:222-283)           k_navfn_costmap   one thread per cell
//   NavFn::setupNavFn / initCost (:379-453)                    k_navfn_plan, all lanes
//   NavFn::propNavFnDijkstra / updateCell (:466-535, 633-701)  k_navfn_plan, lane 0
//   NavFn::propNavFnAstar / updateCellAstar (:548-620, 714-791)
//   NavFn::calcPath / gradCell (:811-1056)
// The expansion is not a Dijkstra proper: cells are relaxed out of three priority buffers in buffer order, a cell sees
// the potentials its predecessors IN THE SAME BLOCK have just written, the buffers hold 10 000 entries and drop what does
// not fit, and the search stops the moment the start cell has a potential - the array it leaves is a snapshot of a
// sequential process, final near the path and provisional elsewhere.  A wavefront that relaxes a block in
// parallel computes a different array.  What is reproduced here is therefore the process itself: one lane per plan walks
// the buffers in the reference's order with the reference's float/double arithmetic, bit for bit, while the other lanes of
// its wave only initialise the arrays; the batch dimension (one plan per robot of a fleet, 256 at a time on 256 CUs, the
// costmaps already resident in HBM) is where the device is used.
#include <hip/hip_runtime.h>

#include "navgpu_device.h"

namespace navgpu {

namespace {
constexpr int kCostUnknownRos = 255, kCostObs = 254, kCostObsRos = 253, kCostNeutral = 50;  // navfn.h:49-67
constexpr float kPotHigh = 1.0e10f;                                                          // navfn.h:77
// `int minp = potarr[stc]` (navfn.cpp:895, gradient_path.cpp:119) with potarr[stc] == POT_HIGH is out of int range; the
// amd64 builds of the reference get cvttss2si's 0x80000000 there (so no neighbour is lower and the trace ends with "high
// potential"); v_cvt_i32_f32 would saturate to INT_MAX and walk on, so the amd64 value is restated.
__device__ __forceinline__ int truncX86(float v) { return (v >= -2147483648.f && v < 2147483648.f) ? (int)v : (int)0x80000000; }
constexpr int kPriorityBufSize = 10000;                                                      // navfn.h:80
}  // namespace

// NavFn::setCostmap: cost_mode 0 = the bytes ARE costarr, 1 = isROS, 2 = plain PGM (borders of 7 cells stay obstacles)
__global__ __launch_bounds__(256) void k_navfn_costmap(NavfnDev nv, uint32_t first, const uint8_t* cmap, size_t cmap_stride, int cost_mode,
                                                       int allow_unknown) {
  const uint32_t plan = first + blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= nv.ns) return;
  const uint8_t in = cmap[(size_t)blockIdx.y * cmap_stride + n];
  uint8_t out = (uint8_t)kCostObs;
  if (cost_mode == 0) {
    out = in;
  } else {
    const int i = n / nv.nx, j = n - i * nv.nx;
    const bool border = cost_mode == 2 && (i < 7 || i > nv.ny - 8 || j < 7 || j > nv.nx - 8);
    if (!border) {
      int v = in;
      if (v < kCostObsRos) {
        v = (int)(kCostNeutral + 0.8 * v);  // COST_NEUTRAL + COST_FACTOR * v, in double, truncated (:238)
        if (v >= kCostObs) v = kCostObs - 1;
        out = (uint8_t)v;
      } else if (v == kCostUnknownRos && (allow_unknown || cost_mode == 2)) {
        out = (uint8_t)(kCostObs - 1);
      }
    }
  }
  nv.costarr[(size_t)plan * nv.ns_padded + n] = out;
}

// NavFn::calcPath / gradCell (:811-1056) over one plan's potential array, one lane: the interpolated gradient descent from the
// start cell, its result record included.  Shared by the two expansions (reference order: k_navfn_plan; tiled wavefront:
// k_navfn_wf_path).
__device__ void navfnCalcPath(const NavfnDev& nv, uint32_t plan, const float* potarr, int goal0, int goal1, int start0, int start1, int n_max,
                              int cycle) {
  const int nx = nv.nx, ny = nv.ny, ns = nv.ns;
  float* gradx = nv.gradx + (size_t)plan * nv.ns_padded;
  float* grady = nv.grady + (size_t)plan * nv.ns_padded;
  float* pathx = nv.path + (size_t)plan * 2 * nv.path_cap;
  float* pathy = pathx + nv.path_cap;
  const int startCell = start1 * nx + start0;
  // ---- gradCell (:1001-1056)
  auto gradCell = [&](int n) {
    if (gradx[n] + grady[n] > 0.0) return;
    if (n < nx || n > ns - nx) return;
    const float cv = potarr[n];
    float dx = 0.0f, dy = 0.0f;
    if (cv >= kPotHigh) {
      if (potarr[n - 1] < kPotHigh)
        dx = -kCostObs;
      else if (potarr[n + 1] < kPotHigh)
        dx = kCostObs;
      if (potarr[n - nx] < kPotHigh)
        dy = -kCostObs;
      else if (potarr[nx + 1] < kPotHigh)  // as written in the reference (:1020)
        dy = kCostObs;
    } else {
      if (potarr[n - 1] < kPotHigh) dx += potarr[n - 1] - cv;
      if (potarr[n + 1] < kPotHigh) dx += cv - potarr[n + 1];
      if (potarr[n - nx] < kPotHigh) dy += potarr[n - nx] - cv;
      if (potarr[n + nx] < kPotHigh) dy += cv - potarr[n + nx];
    }
    float norm = (float)hypot((double)dx, (double)dy);
    if (norm > 0) {
      norm = (float)(1.0 / norm);
      gradx[n] = norm * dx;
      grady[n] = norm * dy;
    }
  };
  // ---- calcPath (:811-985)
  const float pathStep = 0.5f;
  int stc = startCell, npath = 0, found = 0;
  float dx = 0, dy = 0;
  for (int i = 0; i < n_max && i < (int)nv.path_cap; i++) {
    const int nearest_point = max(0, min(nx * ny - 1, stc + (int)round((double)dx) + (int)(nx * round((double)dy))));
    if (potarr[nearest_point] < (float)kCostNeutral) {
      pathx[npath] = (float)goal0;
      pathy[npath] = (float)goal1;
      ++npath;
      found = 1;
      break;
    }
    if (stc < nx || stc > ns - nx) break;  // would be out of bounds
    pathx[npath] = (float)(stc % nx) + dx;
    pathy[npath] = (float)(stc / nx) + dy;
    npath++;
    bool oscillation_detected = false;
    if (npath > 2 && pathx[npath - 1] == pathx[npath - 3] && pathy[npath - 1] == pathy[npath - 3]) oscillation_detected = true;
    const int stcnx = stc + nx, stcpx = stc - nx;
    if (potarr[stc] >= kPotHigh || potarr[stc + 1] >= kPotHigh || potarr[stc - 1] >= kPotHigh || potarr[stcnx] >= kPotHigh ||
        potarr[stcnx + 1] >= kPotHigh || potarr[stcnx - 1] >= kPotHigh || potarr[stcpx] >= kPotHigh || potarr[stcpx + 1] >= kPotHigh ||
        potarr[stcpx - 1] >= kPotHigh || oscillation_detected) {
      // potential-function boundary: follow the grid to the lowest of the eight neighbours (:893-925; minp is an int there)
      int minc = stc;
      int minp = truncX86(potarr[stc]);
      const int nb[8] = {stcpx - 1, stcpx, stcpx + 1, stc - 1, stc + 1, stcnx - 1, stcnx, stcnx + 1};
      for (int q = 0; q < 8; ++q)
        if (potarr[nb[q]] < (float)minp) {
          minp = (int)potarr[nb[q]];
          minc = nb[q];
        }
      stc = minc;
      dx = 0;
      dy = 0;
      if (potarr[stc] >= kPotHigh) break;
    } else {
      gradCell(stc);
      gradCell(stc + 1);
      gradCell(stcnx);
      gradCell(stcnx + 1);
      const float x1 = (float)((1.0 - dx) * gradx[stc] + dx * gradx[stc + 1]);
      const float x2 = (float)((1.0 - dx) * gradx[stcnx] + dx * gradx[stcnx + 1]);
      const float x = (float)((1.0 - dy) * x1 + dy * x2);
      const float y1 = (float)((1.0 - dx) * grady[stc] + dx * grady[stc + 1]);
      const float y2 = (float)((1.0 - dx) * grady[stcnx] + dx * grady[stcnx + 1]);
      const float y = (float)((1.0 - dy) * y1 + dy * y2);
      if (x == 0.0 && y == 0.0) break;  // zero gradient
      const float ss = (float)(pathStep / hypot((double)x, (double)y));
      dx += x * ss;
      dy += y * ss;
      if (dx > 1.0) { stc++; dx = (float)(dx - 1.0); }
      if (dx < -1.0) { stc--; dx = (float)(dx + 1.0); }
      if (dy > 1.0) { stc += nx; dy = (float)(dy - 1.0); }
      if (dy < -1.0) { stc -= nx; dy = (float)(dy + 1.0); }
    }
  }
  navgpu_navfn_result r;
  r.found = found;
  r.path_length = found ? npath : 0;
  r.cycles = cycle;
  r.start_potential = potarr[startCell];
  nv.results[plan] = r;
}

__global__ __launch_bounds__(256) void k_navfn_plan(NavfnDev nv, uint32_t first, const int32_t* goals, const int32_t* starts, int astar,
                                                    int at_start) {
  const uint32_t plan = first + blockIdx.x;
  const int nx = nv.nx, ny = nv.ny, ns = nv.ns;
  uint8_t* costarr = nv.costarr + (size_t)plan * nv.ns_padded;
  uint8_t* pending = nv.pending + (size_t)plan * nv.ns_padded;
  float* potarr = nv.potarr + (size_t)plan * nv.ns_padded;
  float* gradx = nv.gradx + (size_t)plan * nv.ns_padded;
  float* grady = nv.grady + (size_t)plan * nv.ns_padded;
  const int goal0 = goals[2 * blockIdx.x], goal1 = goals[2 * blockIdx.x + 1];
  const int start0 = starts[2 * blockIdx.x], start1 = starts[2 * blockIdx.x + 1];
  // ---- setupNavFn(keepit = true) (:379-440), all lanes
  for (int i = threadIdx.x; i < ns; i += blockDim.x) {
    potarr[i] = kPotHigh;
    gradx[i] = 0.0f;
    grady[i] = 0.0f;
    pending[i] = 0;
    const int y = i / nx, x = i - y * nx;
    if (y == 0 || y == ny - 1 || x == 0 || x == nx - 1) costarr[i] = (uint8_t)kCostObs;  // outer bounds of the cost array
  }
  __syncthreads();
  if (threadIdx.x != 0) return;

  int* curP = nv.pb + (size_t)plan * 3 * kPriorityBufSize;
  int* nextP = curP + kPriorityBufSize;
  int* overP = nextP + kPriorityBufSize;
  int curPe = 0, nextPe = 0, overPe = 0;
  float curT = (float)kCostObs;
  const float priInc = 2 * kCostNeutral;
  auto pushable = [&](int n) { return n >= 0 && n < ns && !pending[n] && costarr[n] < kCostObs; };
  auto push_cur = [&](int n) {  // :367-375
    if (pushable(n) && curPe < kPriorityBufSize) {
      curP[curPe++] = n;
      pending[n] = 1;
    }
  };
  auto push_next = [&](int n) {
    if (pushable(n) && nextPe < kPriorityBufSize) {
      nextP[nextPe++] = n;
      pending[n] = 1;
    }
  };
  auto push_over = [&](int n) {
    if (pushable(n) && overPe < kPriorityBufSize) {
      overP[overPe++] = n;
      pending[n] = 1;
    }
  };
  {  // initCost(goal, 0) (:445-453)
    const int k = goal0 + goal1 * nx;
    potarr[k] = 0.0f;
    push_cur(k + 1);
    push_cur(k - 1);
    push_cur(k - nx);
    push_cur(k + nx);
  }
  // ---- updateCell / updateCellAstar (:466-620)
  auto updateCell = [&](int n) {
    const float l = potarr[n - 1], r = potarr[n + 1], u = potarr[n - nx], d = potarr[n + nx];
    float ta, tc;
    if (l < r) tc = l; else tc = r;
    if (u < d) ta = u; else ta = d;
    if (costarr[n] < kCostObs) {  // don't propagate into obstacles
      const float hf = (float)costarr[n];
      float dc = tc - ta;
      if (dc < 0) {
        dc = -dc;
        ta = tc;
      }
      float pot;
      if (dc >= hf)
        pot = ta + hf;
      else {  // two-neighbour interpolation; the polynomial's literals are doubles
        const float dd = dc / hf;
        const float v = (float)(-0.2301 * dd * dd + 0.5307 * dd + 0.7040);
        pot = ta + hf * v;
      }
      if (pot < potarr[n]) {
        const float le = (float)(0.707106781 * (float)costarr[n - 1]);
        const float re = (float)(0.707106781 * (float)costarr[n + 1]);
        const float ue = (float)(0.707106781 * (float)costarr[n - nx]);
        const float de = (float)(0.707106781 * (float)costarr[n + nx]);
        potarr[n] = pot;
        if (astar) {
          const int x = n % nx, y = n / nx;
          const float dist = (float)(hypot((double)(x - start0), (double)(y - start1)) * (float)kCostNeutral);
          pot += dist;
        }
        if (pot < curT) {  // low-cost buffer block
          if (l > pot + le) push_next(n - 1);
          if (r > pot + re) push_next(n + 1);
          if (u > pot + ue) push_next(n - nx);
          if (d > pot + de) push_next(n + nx);
        } else {  // overflow block
          if (l > pot + le) push_over(n - 1);
          if (r > pot + re) push_over(n + 1);
          if (u > pot + ue) push_over(n - nx);
          if (d > pot + de) push_over(n + nx);
        }
      }
    }
  };
  // ---- propNavFnDijkstra / propNavFnAstar (:633-791)
  const int cycles = max(nx * ny / 20, nx + ny);
  int cycle = 0;
  if (astar) {
    const float dist = (float)(hypot((double)(goal0 - start0), (double)(goal1 - start1)) * (float)kCostNeutral);
    curT = dist + curT;
  }
  const int startCell = start1 * nx + start0;
  for (; cycle < cycles; cycle++) {
    if (curPe == 0 && nextPe == 0) break;
    for (int i = 0; i < curPe; i++) pending[curP[i]] = 0;
    for (int i = 0; i < curPe; i++) updateCell(curP[i]);
    curPe = nextPe;
    nextPe = 0;
    int* pb = curP;
    curP = nextP;
    nextP = pb;
    if (curPe == 0) {
      curT += priInc;
      curPe = overPe;
      overPe = 0;
      pb = curP;
      curP = overP;
      overP = pb;
    }
    if (astar || at_start)
      if (potarr[startCell] < kPotHigh) break;
  }
  navfnCalcPath(nv, plan, potarr, goal0, goal1, start0, start1, astar ? nx * 4 : nx * ny / 2, cycle);
}

// ------------------------------------------------------------------------------------------------
// The expansion as a device algorithm (navgpu_navfn_plan_wavefront): the same update rule - NavFn::updateCell's
// two-neighbour interpolation (navfn.cpp:466-535), float / double arithmetic as written there - relaxed to its FIXED POINT by
// 32 x 32 tiles instead of walked through three priority buffers on one lane.
//   * A tile in LDS (34 x 34 potentials with its halo, 32 x 32 costs) is swept red / black - the four-neighbour stencil is
//     bipartite, so a half-sweep reads only cells of the other colour: race-free, and the same result whatever the waves'
//     timing - until nothing in it changes; then its interior goes back to HBM and the tiles across every edge whose border
//     row changed are marked for the next round.  One launch per round; a launch's workgroups look their tile's mark up and
//     leave if there is none.
//   * Rounds are Jacobi across tiles: round r reads the array round r - 1 wrote (P[(r - 1) & 1]) and writes P[r & 1]; a tile
//     that changed in round r - 1 and is not marked in round r copies itself across, so both arrays stay complete.  Nothing a
//     round reads is written in that round: results do not depend on how the workgroups are scheduled.
//   * Early stop, the counterpart of `if (atStart) if (potarr[startCell] < POT_HIGH) break` (:692-694): an update writes a
//     value above every value it was computed from, so once a round's smallest new value is >= the start cell's potential
//     no later round can write below it - the start cell and everything below its potential are final.  Each round
//     records its number of changed tiles and its smallest new value; the next one reads them before anything else.
// The array this leaves is the update rule's fixed point wherever the potential is below the start cell's, which the
// reference only approaches (its buffers drop entries beyond 10 000, its push tests skip some updates, its early stop leaves
// the last block half done): potentials here are <= the reference's, and the path differs from the reference's by a
// fraction of a cell (tests/test_navfn.py).  The reference-order mode stays the bit-exact one.
// ------------------------------------------------------------------------------------------------
constexpr int kWfTile = 32, kWfThreads = 256;
constexpr uint32_t kWfCopy = 1u, kWfCompute = 2u;

// hf of a cell under a rule: its cost as the update sees it, < 0 = never updated.  navfn: costarr itself, obstacles from COST_OBS
// (navfn.cpp:483); global_planner: DijkstraExpansion::getCost (dijkstra.h:78-87) narrowed to unsigned char as updateCell passes it
// (dijkstra.cpp:178-185)
__device__ __forceinline__ float wfCellCost(const NavfnWfRule& rule, uint8_t cost) {
  if (!rule.global_planner) return cost < kCostObs ? (float)cost : -1.0f;
  float c = cost;
  if (c < rule.lethal_cost - 1 || (rule.allow_unknown && c == 255)) {
    c = c * rule.cost_factor + rule.neutral_cost;
    if (c >= rule.lethal_cost) c = rule.lethal_cost - 1;
    return (float)(uint8_t)c;
  }
  return -1.0f;
}

// the arrays a search starts from: POT_HIGH everywhere but the seeds (navfn: the goal, 0 - setupNavFn + initCost :379-453;
// global_planner: the start cell, or setPreciseStart's four cells - dijkstra.cpp:88-110), zero gradients, the outline; the
// tiles that hold a seed or one of its four neighbours are marked for round 0
__global__ __launch_bounds__(256) void k_navfn_wf_init(NavfnDev nv, uint32_t first, NavfnWfRule rule, const int32_t* seed_cells, const float* seed_vals) {
  const uint32_t plan = first + blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= nv.ns) return;
  const int nx = nv.nx, ny = nv.ny;
  const size_t base = (size_t)plan * nv.ns_padded;
  float p0 = kPotHigh;
  bool mark = false;
  for (int k = 0; k < 4; ++k) {
    const int c = seed_cells[4 * blockIdx.y + k];
    if (c < 0) continue;
    if (n == c) p0 = seed_vals[4 * blockIdx.y + k];
    mark = mark || n == c || n == c - 1 || n == c + 1 || n == c - nx || n == c + nx;
  }
  nv.potarr[base + n] = p0;
  nv.potalt[base + n] = p0;
  nv.gradx[base + n] = 0.0f;
  nv.grady[base + n] = 0.0f;
  const int y = n / nx, x = n - y * nx;
  if (rule.outline && (y == 0 || y == ny - 1 || x == 0 || x == nx - 1)) nv.costarr[base + n] = (uint8_t)kCostObs;  // outer bounds of the cost array (254 = LETHAL_OBSTACLE too)
  if (mark) {
    const int tiles = nv.wf_tiles_x * nv.wf_tiles_y;
    atomicOr(&nv.wf_act[((size_t)plan * 2 + 0) * tiles + (y / kWfTile) * nv.wf_tiles_x + x / kWfTile], kWfCompute);
  }
}

__global__ __launch_bounds__(kWfThreads) void k_navfn_wf_round(NavfnDev nv, uint32_t first, NavfnWfRule rule, const int32_t* stop_cells, int at_start, int round) {
  __shared__ float sP[kWfTile + 2][kWfTile + 4];
  __shared__ float sH[kWfTile][kWfTile];
  __shared__ uint32_t s_sides, s_min, s_chg[2];
  const uint32_t plan = first + blockIdx.y;
  const int nx = nv.nx, ny = nv.ny;
  const int tiles = nv.wf_tiles_x * nv.wf_tiles_y;
  const int tile = blockIdx.x, ty = tile / nv.wf_tiles_x, tx = tile - ty * nv.wf_tiles_x;
  const int tid = threadIdx.x;
  NavfnWfStatus* st = nv.wf_status + plan;
  if (st->done) return;
  const size_t base = (size_t)plan * nv.ns_padded;
  const float* Pin = ((round & 1) ? nv.potarr : nv.potalt) + base;  // what round - 1 wrote
  float* Pout = ((round & 1) ? nv.potalt : nv.potarr) + base;
  uint32_t* nchg = nv.wf_nchg + (size_t)plan * nv.wf_max_rounds;
  uint32_t* minv = nv.wf_min + (size_t)plan * nv.wf_max_rounds;
  if (round > 0) {
    const uint32_t changed_tiles = nchg[round - 1];
    const float low = __uint_as_float(minv[round - 1]);  // (0xFFFFFFFF = a NaN when nothing changed: not read then)
    const float ps = Pin[stop_cells[blockIdx.y]];
    if (changed_tiles == 0 || (at_start && ps < kPotHigh && low >= ps)) {
      if (tile == 0 && tid == 0) {
        st->final_array = (round & 1) ^ 1;  // 0: potarr, 1: potalt
        st->rounds = round;
        st->done = 1;
      }
      return;
    }
  }
  uint32_t* act_cur = nv.wf_act + ((size_t)plan * 2 + (round & 1)) * tiles;
  uint32_t* act_nxt = nv.wf_act + ((size_t)plan * 2 + ((round & 1) ^ 1)) * tiles;
  const uint32_t a = act_cur[tile];
  if (a == 0) return;
  if (tid == 0) {
    s_sides = 0;
    s_min = 0xFFFFFFFFu;
    s_chg[0] = s_chg[1] = 0u;
  }
  const int x0 = tx * kWfTile, y0 = ty * kWfTile;
  // ---- load: interior (4 cells per thread, rows of 32 floats), then the halo ring; off the map = an unreached obstacle
  const uint8_t* cost = nv.costarr + base;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = tid + kWfThreads * k, row = i >> 5, col = i & 31;
    const int gx = x0 + col, gy = y0 + row;
    const bool in = gx < nx && gy < ny;
    sP[row + 1][col + 1] = in ? Pin[gy * nx + gx] : kPotHigh;
    const uint8_t c = in ? cost[gy * nx + gx] : (uint8_t)kCostObs;
    // < 0 = not updated ("don't propagate into obstacles", :483).  global_planner: rows 0 and ny - 1 are never updated either - the
    // reference's updateCell would read potential[n - nx] / [n + nx] outside its arrays there (its outlineMap makes those rows lethal by
    // default; with outline_map off the checker, GlobalPlannerOracle::dijkstraFixedPoint, leaves them out the same way)
    sH[row][col] = (rule.global_planner && (gy == 0 || gy == ny - 1)) ? -1.0f : wfCellCost(rule, c);
  }
  if (tid < 128) {
    const int side = tid >> 5, j = tid & 31;  // 0: row above, 1: row below, 2: column left, 3: column right
    const int gx = side == 0 || side == 1 ? x0 + j : (side == 2 ? x0 - 1 : x0 + kWfTile);
    const int gy = side == 0 ? y0 - 1 : (side == 1 ? y0 + kWfTile : y0 + j);
    const bool in = gx >= 0 && gy >= 0 && gx < nx && gy < ny;
    const float v = in ? Pin[gy * nx + gx] : kPotHigh;
    if (side == 0) sP[0][j + 1] = v;
    else if (side == 1) sP[kWfTile + 1][j + 1] = v;
    else if (side == 2) sP[j + 1][0] = v;
    else sP[j + 1][kWfTile + 1] = v;
  }
  __syncthreads();
  bool any_change = false, capped = false;
  if (a & kWfCompute) {
    uint32_t sides = 0;
    float low = kPotHigh * 4.0f;
    int sweeps = 0;
    // a lane's four cells (two per colour): their costs, and the two neighbour minima their last update was computed from - the
    // update is a function of those and the cost alone, so a cell whose minima have not moved is skipped (most cells, most
    // sweeps: the front inside a tile is a cell or two wide)
    float hf4[4], seen_h[4], seen_v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = tid + kWfThreads * (q & 1), row = i >> 4, col = 2 * (i & 15) + ((row + (q >> 1)) & 1);
      hf4[q] = sH[row][col];
      seen_h[q] = seen_v[q] = -1.0f;  // (no potential is negative)
    }
    // "did anything change this sweep" through two alternating LDS flags (any lane that lowers a cell sets this sweep's flag; the
    // other flag is cleared between the two barriers of the sweep, when nobody reads or sets it): two barriers per sweep, where
    // __syncthreads_or costs three of its own
    for (;;) {
      int changed = 0;
      volatile uint32_t* flag = &s_chg[sweeps & 1];
#pragma unroll
      for (int colour = 0; colour < 2; ++colour) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int i = tid + kWfThreads * k, row = i >> 4, col = 2 * (i & 15) + ((row + colour) & 1);
          const int q = 2 * colour + k;
          const float hf = hf4[q];
          const float l = sP[row + 1][col], r = sP[row + 1][col + 2], u = sP[row][col + 1], d = sP[row + 2][col + 1];
          float ta, tc;
          if (l < r) tc = l; else tc = r;
          if (u < d) ta = u; else ta = d;
          if (tc == seen_h[q] && ta == seen_v[q]) continue;
          seen_h[q] = tc;
          seen_v[q] = ta;
          float dc = tc - ta;
          if (dc < 0) {
            dc = -dc;
            ta = tc;
          }
          float pot;
          if (!rule.quadratic)  // PotentialCalculator::calculatePotential (potential_calculator.h:50-59)
            pot = fminf(fminf(l, r), fminf(u, d)) + hf;
          else if (dc >= hf)
            pot = ta + hf;
          else {  // (hf > 0 here: 0 <= dc < hf)
            const float dd = dc / hf;
            const float v = (float)(-0.2301 * dd * dd + 0.5307 * dd + 0.7040);
            pot = ta + hf * v;
          }
          if (hf >= 0.0f && pot < sP[row + 1][col + 1]) {
            sP[row + 1][col + 1] = pot;
            changed = 1;
            low = fminf(low, pot);
            sides |= (row == 0 ? 1u : 0u) | (row == kWfTile - 1 ? 2u : 0u) | (col == 0 ? 4u : 0u) | (col == kWfTile - 1 ? 8u : 0u);
          }
        }
        if (colour == 0) {
          if (changed) *flag = 1u;
          changed = 0;
          __syncthreads();
          if (tid == 0) s_chg[(sweeps + 1) & 1] = 0u;
        }
      }
      if (changed) *flag = 1u;
      __syncthreads();
      const uint32_t any = *flag;
      if (!any) break;
      any_change = true;
      if (++sweeps >= rule.max_sweeps) {  // go on next round, with the neighbours' news
        capped = true;
        break;
      }
    }
    if (any_change) {
      // (positive floats order like their bit patterns)
      for (int off = 32; off > 0; off >>= 1) {
        low = fminf(low, __shfl_down(low, off));
        sides |= __shfl_down(sides, off);
      }
      if ((tid & 63) == 0) {
        atomicMin(&s_min, __float_as_uint(low));
        atomicOr(&s_sides, sides);
      }
      __syncthreads();
    }
  }
  if (any_change || (a & kWfCopy)) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + kWfThreads * k, row = i >> 5, col = i & 31;
      const int gx = x0 + col, gy = y0 + row;
      if (gx < nx && gy < ny) Pout[gy * nx + gx] = sP[row + 1][col + 1];
    }
  }
  if (tid == 0) {
    act_cur[tile] = 0;  // (read again two rounds on; set only by round + 1)
    if (any_change) {
      const uint32_t sd = s_sides;
      atomicOr(&act_nxt[tile], kWfCopy | (capped ? kWfCompute : 0u));
      if ((sd & 1u) && ty > 0) atomicOr(&act_nxt[tile - nv.wf_tiles_x], kWfCompute);
      if ((sd & 2u) && ty + 1 < nv.wf_tiles_y) atomicOr(&act_nxt[tile + nv.wf_tiles_x], kWfCompute);
      if ((sd & 4u) && tx > 0) atomicOr(&act_nxt[tile - 1], kWfCompute);
      if ((sd & 8u) && tx + 1 < nv.wf_tiles_x) atomicOr(&act_nxt[tile + 1], kWfCompute);
      atomicAdd(&nchg[round], 1u);
      atomicMin(&minv[round], s_min);
    }
  }
}

// NavFn::calcPath / gradCell (:811-1056) by one WAVE: the walk itself is sequential (a step needs the cell and offset the step
// before it left), but a step alone on one lane costs ~3 us of dependent L2 round trips (the 3 x 3 neighbourhood, then four
// gradCell calls of five reads each, memoised through two more arrays).  Here a 64 x 64 window of the potential array around
// the walker sits in LDS (reloaded when the walker comes within two cells of its rim; addressed by FLAT index, so a read that
// runs off a row's end sees what the reference's flat array holds there), the nine neighbourhood reads are nine lanes and a
// ballot, the four gradCell calls four lanes, and every lane carries the walker's state.  gradCell is a pure function of the
// potential array (its gradx / grady memo only saves recomputation), so the arithmetic per value is that of navfnCalcPath,
// expression for expression.
constexpr int kPathWin = 64;
__global__ __launch_bounds__(64) void k_navfn_wf_path(NavfnDev nv, uint32_t first, const int32_t* goals, const int32_t* starts) {
  __shared__ float sW[kPathWin * kPathWin];
  const uint32_t plan = first + blockIdx.x;
  const int lane = threadIdx.x;
  const NavfnWfStatus st = nv.wf_status[plan];
  const float* potarr = (st.final_array ? nv.potalt : nv.potarr) + (size_t)plan * nv.ns_padded;
  const int nx = nv.nx, ny = nv.ny, ns = nv.ns;
  float* pathx = nv.path + (size_t)plan * 2 * nv.path_cap;
  float* pathy = pathx + nv.path_cap;
  const int goal0 = goals[2 * blockIdx.x], goal1 = goals[2 * blockIdx.x + 1];
  const int startCell = starts[2 * blockIdx.x + 1] * nx + starts[2 * blockIdx.x];
  const int n_max = nx * ny / 2;
  const float pot_nx1 = potarr[nx + 1];  // gradCell's `potarr[nx + 1]` (:1020, as written in the reference)
  int wx0 = 0, wy0 = 0;
  bool have_win = false;
  const float pathStep = 0.5f;
  int stc = startCell, npath = 0, found = 0;
  float dx = 0, dy = 0;
  float px1 = 0, py1 = 0, px2 = 0, py2 = 0;  // the two points before the last one (oscillation test)
  for (int i = 0; i < n_max && i < (int)nv.path_cap; i++) {
    const int sy = stc / nx, sx = stc - sy * nx;
    if (!have_win || sx - wx0 < 2 || sx - wx0 > kPathWin - 4 || sy - wy0 < 2 || sy - wy0 > kPathWin - 4) {
      __syncthreads();
      wx0 = sx - kPathWin / 2;
      wy0 = sy - kPathWin / 2;
      for (int k = lane; k < kPathWin * kPathWin; k += 64) {
        const int j = k / kPathWin, c = k - j * kPathWin;
        const long n = (long)(wy0 + j) * nx + (wx0 + c);
        sW[k] = (n >= 0 && n < ns) ? potarr[n] : kPotHigh;
      }
      have_win = true;
      __syncthreads();
    }
    auto P = [&](int ox, int oy) -> float { return sW[(sy - wy0 + oy) * kPathWin + (sx - wx0 + ox)]; };  // potarr[stc + ox + oy * nx]
    {
      const int want = stc + (int)round((double)dx) + (int)(nx * round((double)dy));
      const int nearest_point = max(0, min(nx * ny - 1, want));
      const float pn = nearest_point == want ? P((int)round((double)dx), (int)round((double)dy)) : potarr[nearest_point];
      if (pn < (float)kCostNeutral) {
        if (lane == 0) {
          pathx[npath] = (float)goal0;
          pathy[npath] = (float)goal1;
        }
        ++npath;
        found = 1;
        break;
      }
    }
    if (stc < nx || stc > ns - nx) break;  // would be out of bounds
    const float cx = (float)(stc % nx) + dx, cy = (float)(stc / nx) + dy;
    if (lane == 0) {
      pathx[npath] = cx;
      pathy[npath] = cy;
    }
    npath++;
    const bool oscillation_detected = npath > 2 && cx == px2 && cy == py2;
    px2 = px1;
    py2 = py1;
    px1 = cx;
    py1 = cy;
    const int l9 = lane < 9 ? lane : 0;
    const bool high9 = lane < 9 && P(l9 % 3 - 1, l9 / 3 - 1) >= kPotHigh;
    if (__ballot(high9) != 0ull || oscillation_detected) {
      // potential-function boundary: follow the grid to the lowest of the eight neighbours (:893-925; minp is an int there)
      int mox = 0, moy = 0;
      int minp = truncX86(P(0, 0));
      const int ox[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, oy[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
      for (int q = 0; q < 8; ++q) {
        const float v = P(ox[q], oy[q]);
        if (v < (float)minp) {
          minp = (int)v;
          mox = ox[q];
          moy = oy[q];
        }
      }
      const float pm = P(mox, moy);
      stc += mox + moy * nx;
      dx = 0;
      dy = 0;
      if (pm >= kPotHigh) break;
    } else {
      // gradCell (:1001-1056) of stc, stc + 1, stc + nx, stc + nx + 1 on lanes 0..3
      const int q = lane & 3, qx = q & 1, qy = q >> 1;
      const int n = stc + qx + qy * nx;
      float gx = 0.0f, gy = 0.0f;
      if (!(n < nx || n > ns - nx)) {
        const float cv = P(qx, qy);
        float ddx = 0.0f, ddy = 0.0f;
        if (cv >= kPotHigh) {
          if (P(qx - 1, qy) < kPotHigh)
            ddx = -kCostObs;
          else if (P(qx + 1, qy) < kPotHigh)
            ddx = kCostObs;
          if (P(qx, qy - 1) < kPotHigh)
            ddy = -kCostObs;
          else if (pot_nx1 < kPotHigh)
            ddy = kCostObs;
        } else {
          if (P(qx - 1, qy) < kPotHigh) ddx += P(qx - 1, qy) - cv;
          if (P(qx + 1, qy) < kPotHigh) ddx += cv - P(qx + 1, qy);
          if (P(qx, qy - 1) < kPotHigh) ddy += P(qx, qy - 1) - cv;
          if (P(qx, qy + 1) < kPotHigh) ddy += cv - P(qx, qy + 1);
        }
        float norm = (float)hypot((double)ddx, (double)ddy);
        if (norm > 0) {
          norm = (float)(1.0 / norm);
          gx = norm * ddx;
          gy = norm * ddy;
        }
      }
      const float gx0 = __shfl(gx, 0), gx1 = __shfl(gx, 1), gx2 = __shfl(gx, 2), gx3 = __shfl(gx, 3);
      const float gy0 = __shfl(gy, 0), gy1 = __shfl(gy, 1), gy2 = __shfl(gy, 2), gy3 = __shfl(gy, 3);
      const float x1 = (float)((1.0 - dx) * gx0 + dx * gx1);
      const float x2 = (float)((1.0 - dx) * gx2 + dx * gx3);
      const float x = (float)((1.0 - dy) * x1 + dy * x2);
      const float y1 = (float)((1.0 - dx) * gy0 + dx * gy1);
      const float y2 = (float)((1.0 - dx) * gy2 + dx * gy3);
      const float y = (float)((1.0 - dy) * y1 + dy * y2);
      if (x == 0.0 && y == 0.0) break;  // zero gradient
      const float ss = (float)(pathStep / hypot((double)x, (double)y));
      dx += x * ss;
      dy += y * ss;
      if (dx > 1.0) { stc++; dx = (float)(dx - 1.0); }
      if (dx < -1.0) { stc--; dx = (float)(dx + 1.0); }
      if (dy > 1.0) { stc += nx; dy = (float)(dy - 1.0); }
      if (dy < -1.0) { stc -= nx; dy = (float)(dy + 1.0); }
    }
  }
  if (lane == 0) {
    navgpu_navfn_result r;
    r.found = found;
    r.path_length = found ? npath : 0;
    r.cycles = st.rounds;
    r.start_potential = potarr[startCell];
    nv.results[plan] = r;
  }
}

// ------------------------------------------------------------------------------------------------
// global_planner (the other half of SURVEY 8 f-4): what GlobalPlanner::makePlan runs between worldToMap and the plan
// assembly (global_planner/src/planner_core.cpp:250-306) -
//   outlineMap (:62-76), DijkstraExpansion::calculatePotentials / updateCell / getCost (dijkstra.cpp:71-229, dijkstra.h:78-87)
//   or AStarExpansion::calculatePotentials / add (astar.cpp:46-95: std::push_heap / pop_heap with greater1),
//   PotentialCalculator / QuadraticCalculator::calculatePotential (potential_calculator.h:50-59, quadratic_calculator.cpp:41-77),
//   Expander::clearEndpoint (expander.h:76-89), GradientPath::getPath / gradCell (gradient_path.cpp:68-313) or
//   GridPath::getPath (grid_path.cpp:44-82).
// Same kind of process as NavFn's (priority buffers in buffer order, early stop at the goal cell), same treatment: one
// lane per plan, bit for bit.  costarr holds the raw costmap bytes (navgpu_navfn_set_costmap with cost_mode 0).
// ------------------------------------------------------------------------------------------------
struct GpHeapEntry {  // astar.h:47-55 Index
  int i;
  float cost;
};
// WAVEFRONT: the expansion has already run as a tiled wavefront (k_navfn_wf_round with the global_planner rule) and left its
// potentials in `wf_potential`, its round count in `wf_rounds`; what remains is makePlan's tail - found_legal, clearEndpoint,
// the traceback - on one lane, the same code as the reference-order kernel's.
template <bool WAVEFRONT>
__device__ __forceinline__ void gpPlanBody(const NavfnDev& nv, uint32_t first, const navgpu_global_planner_params& gp, const double* starts,
                                           const double* goals, const int32_t* goal_cells, GpHeapEntry* heaps, float* wf_potential, int wf_rounds) {
  const uint32_t plan = first + blockIdx.x;
  const int nx = nv.nx, ny = nv.ny, ns = nv.ns;
  uint8_t* costs = nv.costarr + (size_t)plan * nv.ns_padded;
  uint8_t* pending = nv.pending + (size_t)plan * nv.ns_padded;
  float* potential = WAVEFRONT ? wf_potential : nv.potarr + (size_t)plan * nv.ns_padded;
  float* gradx = nv.gradx + (size_t)plan * nv.ns_padded;
  float* grady = nv.grady + (size_t)plan * nv.ns_padded;
  float* pathx = nv.path + (size_t)plan * 2 * nv.path_cap;
  float* pathy = pathx + nv.path_cap;
  const double start_x = starts[2 * blockIdx.x], start_y = starts[2 * blockIdx.x + 1];
  const double goal_x = goals[2 * blockIdx.x], goal_y = goals[2 * blockIdx.x + 1];
  const int lethal = gp.lethal_cost, neutral = gp.neutral_cost;
  const float factor = gp.cost_factor;
  const bool unknown = gp.allow_unknown != 0, quadratic = gp.use_quadratic != 0;
  constexpr float kHigh = 1.0e10f;
  // all lanes: the arrays every expansion starts from, and GlobalPlanner::outlineMap
  if (!WAVEFRONT) {
    for (int i = threadIdx.x; i < ns; i += blockDim.x) {
      potential[i] = kHigh;
      gradx[i] = 0.0f;
      grady[i] = 0.0f;
      pending[i] = 0;
      const int y = i / nx, x = i - y * nx;
      if (gp.outline_map && (y == 0 || y == ny - 1 || x == 0 || x == nx - 1)) costs[i] = 254;  // costmap_2d::LETHAL_OBSTACLE
    }
    __syncthreads();
  }
  (void)ny;
  (void)pending;
  if (threadIdx.x != 0) return;

  // A border cell can enter the expansion when nothing outlines the map (gp.outline_map == 0) or when the outline's 254 is
  // below lethal_cost (A* with lethal_cost 255).  The reference then reads potential[n - nx] / costs[n + nx] outside its
  // arrays (harmless garbage on the CPU heap); here every neighbour access goes through potAt() / getCost(), which give
  // an off-array cell the values of an unreached lethal one.  In-array reads are unchanged, bit for bit.
  auto potAt = [&](int n) -> float { return (n >= 0 && n < ns) ? potential[n] : kHigh; };
  auto calculatePotential = [&](uint8_t cost, int n, float prev_potential) -> float {
    if (!quadratic) {
      if (prev_potential < 0) {
        const float min_h = fminf(potAt(n - 1), potAt(n + 1)), min_v = fminf(potAt(n - nx), potAt(n + nx));
        prev_potential = fminf(min_h, min_v);
      }
      return prev_potential + cost;
    }
    const float l = potAt(n - 1), r = potAt(n + 1), u = potAt(n - nx), d = potAt(n + nx);
    float ta, tc;
    if (l < r) tc = l; else tc = r;
    if (u < d) ta = u; else ta = d;
    const float hf = cost;
    float dc = tc - ta;
    if (dc < 0) {
      dc = -dc;
      ta = tc;
    }
    if (dc >= hf) return ta + hf;
    const float dd = dc / hf;
    const float v = (float)(-0.2301 * dd * dd + 0.5307 * dd + 0.7040);
    return ta + hf * v;
  };
  auto getCost = [&](int n) -> float {
    if (n < 0 || n >= ns) return lethal;
    float c = costs[n];
    if (c < lethal - 1 || (unknown && c == 255)) {
      c = c * factor + neutral;
      if (c >= lethal) c = lethal - 1;
      return c;
    }
    return lethal;
  };
  const int cycles = nx * ny * 2;
  int cycle = 0;
  bool found_legal = false;
  const int endCell = (int)goal_x + nx * (int)goal_y;
  if (WAVEFRONT) {
    cycle = wf_rounds;
    found_legal = potential[endCell] < kHigh;  // the reference leaves its loop through `break` exactly when the goal cell has a potential
  } else if (gp.use_dijkstra) {
    int* cur = nv.pb + (size_t)plan * 3 * kPriorityBufSize;
    int* nxt = cur + kPriorityBufSize;
    int* ovr = nxt + kPriorityBufSize;
    int curE = 0, nxtE = 0, ovrE = 0;
    float threshold = lethal;
    const float priorityIncrement = 2 * neutral;
    auto push = [&](int* buf, int& end, int n) {
      if (n >= 0 && n < ns && !pending[n] && getCost(n) < lethal && end < kPriorityBufSize) {
        buf[end++] = n;
        pending[n] = 1;
      }
    };
    const int k = (int)start_x + nx * (int)start_y;
    if (!gp.old_navfn_behavior) {  // setPreciseStart(true) (planner_core.cpp:124-127)
      double dx = start_x - (int)start_x, dy = start_y - (int)start_y;
      dx = floorf((float)(dx * 100 + 0.5)) / 100;
      dy = floorf((float)(dy * 100 + 0.5)) / 100;
      potential[k] = (float)(neutral * 2 * dx * dy);
      potential[k + 1] = (float)(neutral * 2 * (1 - dx) * dy);
      potential[k + nx] = (float)(neutral * 2 * dx * (1 - dy));
      potential[k + nx + 1] = (float)(neutral * 2 * (1 - dx) * (1 - dy));
      push(cur, curE, k + 2);
      push(cur, curE, k - 1);
      push(cur, curE, k + nx - 1);
      push(cur, curE, k + nx + 2);
      push(cur, curE, k - nx);
      push(cur, curE, k - nx + 1);
      push(cur, curE, k + nx * 2);
      push(cur, curE, k + nx * 2 + 1);
    } else {
      potential[k] = 0;
      push(cur, curE, k + 1);
      push(cur, curE, k - 1);
      push(cur, curE, k - nx);
      push(cur, curE, k + nx);
    }
    bool ran_dry = false;
    for (; cycle < cycles; cycle++) {
      if (curE == 0 && nxtE == 0) {
        ran_dry = true;
        break;
      }
      for (int i = 0; i < curE; i++) pending[cur[i]] = 0;
      for (int i = 0; i < curE; i++) {  // updateCell
        const int n = cur[i];
        const float c = getCost(n);
        if (c >= lethal) continue;
        const float pot = calculatePotential((uint8_t)c, n, -1.0f);
        if (pot < potential[n]) {
          const float le = (float)(0.707106781 * (float)getCost(n - 1));
          const float re = (float)(0.707106781 * (float)getCost(n + 1));
          const float ue = (float)(0.707106781 * (float)getCost(n - nx));
          const float de = (float)(0.707106781 * (float)getCost(n + nx));
          potential[n] = pot;
          if (pot < threshold) {
            if (potAt(n - 1) > pot + le) push(nxt, nxtE, n - 1);
            if (potAt(n + 1) > pot + re) push(nxt, nxtE, n + 1);
            if (potAt(n - nx) > pot + ue) push(nxt, nxtE, n - nx);
            if (potAt(n + nx) > pot + de) push(nxt, nxtE, n + nx);
          } else {
            if (potAt(n - 1) > pot + le) push(ovr, ovrE, n - 1);
            if (potAt(n + 1) > pot + re) push(ovr, ovrE, n + 1);
            if (potAt(n - nx) > pot + ue) push(ovr, ovrE, n - nx);
            if (potAt(n + nx) > pot + de) push(ovr, ovrE, n + nx);
          }
        }
      }
      curE = nxtE;
      nxtE = 0;
      int* t = cur;
      cur = nxt;
      nxt = t;
      if (curE == 0) {
        threshold += priorityIncrement;
        curE = ovrE;
        ovrE = 0;
        t = cur;
        cur = ovr;
        ovr = t;
      }
      if (potential[endCell] < kHigh) break;
    }
    found_legal = !ran_dry && cycle < cycles;
  } else {
    GpHeapEntry* heap = heaps + (size_t)plan * nv.ns_padded;
    long len = 0;
    auto pushHeap = [&](long hole, const GpHeapEntry value) {  // std::__push_heap with greater1: parent.cost > value.cost moves down
      long parent = (hole - 1) / 2;
      while (hole > 0 && heap[parent].cost > value.cost) {
        heap[hole] = heap[parent];
        hole = parent;
        parent = (hole - 1) / 2;
      }
      heap[hole] = value;
    };
    const int start_i = (int)start_x + nx * (int)start_y;
    heap[len++] = GpHeapEntry{start_i, 0.0f};  // queue_.push_back(Index(start_i, 0)) - no push_heap on the first element
    potential[start_i] = 0;
    const int ex = (int)goal_x, ey = (int)goal_y;
    auto add = [&](float prev_potential, int next_i) {
      if (next_i < 0 || next_i >= ns) return;
      if (potential[next_i] < kHigh) return;
      if (costs[next_i] >= lethal && !(unknown && costs[next_i] == 255)) return;
      potential[next_i] = calculatePotential((uint8_t)(costs[next_i] + neutral), next_i, prev_potential);
      const int x = next_i % nx, y = next_i / nx;
      const float distance = (float)(abs(ex - x) + abs(ey - y));
      pushHeap(len, GpHeapEntry{next_i, potential[next_i] + distance * neutral});
      ++len;
    };
    while (len > 0 && cycle < cycles) {
      const GpHeapEntry top = heap[0];
      if (len > 1) {  // std::pop_heap: __adjust_heap(first, 0, len - 1, value = last element)
        const GpHeapEntry value = heap[len - 1];
        const long l = len - 1;
        long hole = 0, child = 0;
        while (child < (l - 1) / 2) {
          child = 2 * (child + 1);
          if (heap[child].cost > heap[child - 1].cost) child--;
          heap[hole] = heap[child];
          hole = child;
        }
        if ((l & 1) == 0 && child == (l - 2) / 2) {
          child = 2 * (child + 1);
          heap[hole] = heap[child - 1];
          hole = child - 1;
        }
        pushHeap(hole, value);
      }
      --len;
      const int i = top.i;
      if (i == endCell) {
        found_legal = true;
        break;
      }
      add(potential[i], i + 1);
      add(potential[i], i - 1);
      add(potential[i], i + nx);
      add(potential[i], i - nx);
      cycle++;
    }
  }
  if (!gp.old_navfn_behavior) {  // Expander::clearEndpoint(costs, potential, goal_x_i, goal_y_i, 2) (expander.h:76-89)
    const int startCell = goal_cells[2 * blockIdx.x] + nx * goal_cells[2 * blockIdx.x + 1];
    for (int i = -2; i <= 2; i++)
      for (int j = -2; j <= 2; j++) {
        const int n = startCell + i + nx * j;
        if (n < nx + 1 || n >= ns - nx - 1) continue;  // (the reference reads outside its arrays for a goal this close to the border)
        if (potential[n] < kHigh) continue;
        const float c = (float)(costs[n] + neutral);
        potential[n] = calculatePotential((uint8_t)c, n, -1.0f);
      }
  }
  // ---- traceback (only when a legal potential was found, planner_core.cpp:303-311)
  int npath = 0, found = 0;
  float last3x[3] = {0, 0, 0}, last3y[3] = {0, 0, 0};  // the path's last three points (all of it may not fit the buffer)
  auto pushPoint = [&](float x, float y) {
    if (npath < (int)nv.path_cap) {
      pathx[npath] = x;
      pathy[npath] = y;
    }
    last3x[0] = last3x[1];
    last3y[0] = last3y[1];
    last3x[1] = last3x[2];
    last3y[1] = last3y[2];
    last3x[2] = x;
    last3y[2] = y;
    ++npath;
  };
  if (found_legal && !gp.use_grid_path) {  // GradientPath::getPath (gradient_path.cpp:68-248)
    auto gradCell = [&](int n) {
      if (n < nx || n > nx * ny - nx) return;  // (the reference tests this second: its gradx_[n] read may lie past the array)
      if (gradx[n] + grady[n] > 0.0) return;
      const float cv = potential[n];
      float dx = 0.0f, dy = 0.0f;
      if (cv >= kHigh) {
        if (potential[n - 1] < kHigh)
          dx = -lethal;
        else if (potential[n + 1] < kHigh)
          dx = lethal;
        if (potential[n - nx] < kHigh)
          dy = -lethal;
        else if (potential[nx + 1] < kHigh)  // as written in the reference (:287)
          dy = lethal;
      } else {
        if (potential[n - 1] < kHigh) dx += potential[n - 1] - cv;
        if (potential[n + 1] < kHigh) dx += cv - potential[n + 1];
        if (potential[n - nx] < kHigh) dy += potential[n - nx] - cv;
        if (potAt(n + nx) < kHigh) dy += cv - potAt(n + nx);  // (n = ns - nx passes the test above)
      }
      float norm = (float)hypot((double)dx, (double)dy);
      if (norm > 0) {
        norm = (float)(1.0 / norm);
        gradx[n] = norm * dx;
        grady[n] = norm * dy;
      }
    };
    int stc = (int)goal_x + nx * (int)goal_y;
    float dx = (float)(goal_x - (int)goal_x), dy = (float)(goal_y - (int)goal_y);
    const long lim = (long)ns * 4;
    long c = 0;
    while (c++ < lim) {
      const double px = stc % nx + dx, py = stc / nx + dy;
      if (fabs(px - start_x) < .5 && fabs(py - start_y) < .5) {
        pushPoint((float)start_x, (float)start_y);
        found = 1;
        break;
      }
      if (stc < nx || stc > nx * ny - nx) break;
      pushPoint((float)px, (float)py);
      const bool oscillation_detected = npath > 2 && last3x[2] == last3x[0] && last3y[2] == last3y[0];
      const int stcnx = stc + nx, stcpx = stc - nx;
      if (potential[stc] >= kHigh || potAt(stc + 1) >= kHigh || potAt(stc - 1) >= kHigh || potAt(stcnx) >= kHigh ||
          potAt(stcnx + 1) >= kHigh || potAt(stcnx - 1) >= kHigh || potAt(stcpx) >= kHigh || potAt(stcpx + 1) >= kHigh ||
          potAt(stcpx - 1) >= kHigh || oscillation_detected) {
        int minc = stc;
        int minp = truncX86(potential[stc]);
        const int nb[8] = {stcpx - 1, stcpx, stcpx + 1, stc - 1, stc + 1, stcnx - 1, stcnx, stcnx + 1};
        for (int q = 0; q < 8; ++q)
          if (potAt(nb[q]) < (float)minp) {
            minp = (int)potAt(nb[q]);
            minc = nb[q];
          }
        stc = minc;
        dx = 0;
        dy = 0;
        if (potential[stc] >= kHigh) break;
      } else {
        gradCell(stc);
        gradCell(stc + 1);
        gradCell(stcnx);
        gradCell(stcnx + 1);
        auto gAt = [&](const float* g, int n) -> float { return n < ns ? g[n] : 0.0f; };  // (stc on the last row: the reference reads past its arrays)
        const float x1 = (float)((1.0 - dx) * gradx[stc] + dx * gAt(gradx, stc + 1));
        const float x2 = (float)((1.0 - dx) * gAt(gradx, stcnx) + dx * gAt(gradx, stcnx + 1));
        const float x = (float)((1.0 - dy) * x1 + dy * x2);
        const float y1 = (float)((1.0 - dx) * grady[stc] + dx * gAt(grady, stc + 1));
        const float y2 = (float)((1.0 - dx) * gAt(grady, stcnx) + dx * gAt(grady, stcnx + 1));
        const float y = (float)((1.0 - dy) * y1 + dy * y2);
        if (x == 0.0 && y == 0.0) break;
        const float ss = (float)(0.5f / hypot((double)x, (double)y));  // pathStep_ = 0.5 (:47)
        dx += x * ss;
        dy += y * ss;
        if (dx > 1.0) { stc++; dx = (float)(dx - 1.0); }
        if (dx < -1.0) { stc--; dx = (float)(dx + 1.0); }
        if (dy > 1.0) { stc += nx; dy = (float)(dy - 1.0); }
        if (dy < -1.0) { stc -= nx; dy = (float)(dy + 1.0); }
      }
    }
  } else if (found_legal) {  // GridPath::getPath (grid_path.cpp:44-82)
    float cx = (float)goal_x, cy = (float)goal_y;
    const int start_index = (int)start_x + nx * (int)start_y;
    pushPoint(cx, cy);
    long c = 0;
    found = 1;
    while ((int)cx + nx * (int)cy != start_index) {
      float min_val = 1e10f;
      int min_x = 0, min_y = 0;
      for (int xd = -1; xd <= 1; xd++)
        for (int yd = -1; yd <= 1; yd++) {
          if (xd == 0 && yd == 0) continue;
          const int x = (int)(cx + xd), y = (int)(cy + yd);
          const int index = x + nx * y;
          if (index < 0 || index >= ns) continue;  // (the reference reads outside its array here)
          if (potential[index] < min_val) {
            min_val = potential[index];
            min_x = x;
            min_y = y;
          }
        }
      if (min_x == 0 && min_y == 0) {
        found = 0;
        break;
      }
      cx = (float)min_x;
      cy = (float)min_y;
      pushPoint(cx, cy);
      if (c++ > (long)ns * 4) {
        found = 0;
        break;
      }
    }
  }
  navgpu_navfn_result r;
  r.found = (found && npath <= (int)nv.path_cap) ? 1 : 0;
  r.path_length = r.found ? npath : 0;
  r.cycles = cycle;
  r.start_potential = potential[endCell];
  nv.results[plan] = r;
}
__global__ __launch_bounds__(256) void k_gp_plan(NavfnDev nv, uint32_t first, navgpu_global_planner_params gp, const double* starts, const double* goals,
                                                 const int32_t* goal_cells, GpHeapEntry* heaps) {
  gpPlanBody<false>(nv, first, gp, starts, goals, goal_cells, heaps, nullptr, 0);
}
__global__ __launch_bounds__(64) void k_gp_wf_finish(NavfnDev nv, uint32_t first, navgpu_global_planner_params gp, const double* starts, const double* goals,
                                                     const int32_t* goal_cells) {
  const uint32_t plan = first + blockIdx.x;
  const NavfnWfStatus st = nv.wf_status[plan];
  gpPlanBody<true>(nv, first, gp, starts, goals, goal_cells, nullptr, (st.final_array ? nv.potalt : nv.potarr) + (size_t)plan * nv.ns_padded, st.rounds);
}

void launch_navfn_costmap(const NavfnDev& nv, uint32_t first, uint32_t count, const uint8_t* cmap, size_t stride, int cost_mode, int allow_unknown,
                          hipStream_t s) {
  hipLaunchKernelGGL(k_navfn_costmap, dim3((nv.ns + 255) / 256, count), dim3(256), 0, s, nv, first, cmap, stride, cost_mode, allow_unknown);
}
void launch_navfn_plan(const NavfnDev& nv, uint32_t first, uint32_t count, const int32_t* goals, const int32_t* starts, int astar, int at_start,
                       hipStream_t s) {
  hipLaunchKernelGGL(k_navfn_plan, dim3(count), dim3(256), 0, s, nv, first, goals, starts, astar, at_start);
}

void launch_navfn_wf_init(const NavfnDev& nv, uint32_t first, uint32_t count, const NavfnWfRule& rule, const int32_t* seed_cells, const float* seed_vals,
                          hipStream_t s) {
  hipLaunchKernelGGL(k_navfn_wf_init, dim3((nv.ns + 255) / 256, count), dim3(256), 0, s, nv, first, rule, seed_cells, seed_vals);
}
void launch_navfn_wf_round(const NavfnDev& nv, uint32_t first, uint32_t count, const NavfnWfRule& rule, const int32_t* stop_cells, int at_start, int round,
                           hipStream_t s) {
  hipLaunchKernelGGL(k_navfn_wf_round, dim3(nv.wf_tiles_x * nv.wf_tiles_y, count), dim3(kWfThreads), 0, s, nv, first, rule, stop_cells, at_start, round);
}
void launch_gp_wf_finish(const NavfnDev& nv, uint32_t first, uint32_t count, const navgpu_global_planner_params& gp, const double* starts,
                         const double* goals, const int32_t* goal_cells, hipStream_t s) {
  hipLaunchKernelGGL(k_gp_wf_finish, dim3(count), dim3(64), 0, s, nv, first, gp, starts, goals, goal_cells);
}
void launch_navfn_wf_path(const NavfnDev& nv, uint32_t first, uint32_t count, const int32_t* goals, const int32_t* starts, hipStream_t s) {
  hipLaunchKernelGGL(k_navfn_wf_path, dim3(count), dim3(64), 0, s, nv, first, goals, starts);
}

void launch_gp_plan(const NavfnDev& nv, uint32_t first, uint32_t count, const navgpu_global_planner_params& gp, const double* starts,
                    const double* goals, const int32_t* goal_cells, void* heaps, hipStream_t s) {
  hipLaunchKernelGGL(k_gp_plan, dim3(count), dim3(256), 0, s, nv, first, gp, starts, goals, goal_cells, static_cast<GpHeapEntry*>(heaps));
}

}  // namespace navgpu
