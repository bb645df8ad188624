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

__global__ __launch_bounds__(256) void k_navfn_plan(NavfnDev nv, uint32_t first, const int32_t* goals, const int32_t* starts, int astar,
                                                    int at_start) {
  const uint32_t plan = first + blockIdx.x;
  const int nx = nv.nx, ny = nv.ny, ns = nv.ns;
  uint8_t* costarr = nv.costarr + (size_t)plan * nv.ns_padded;
  uint8_t* pending = nv.pending + (size_t)plan * nv.ns_padded;
  float* potarr = nv.potarr + (size_t)plan * nv.ns_padded;
  float* gradx = nv.gradx + (size_t)plan * nv.ns_padded;
  float* grady = nv.grady + (size_t)plan * nv.ns_padded;
  float* pathx = nv.path + (size_t)plan * 2 * nv.path_cap;
  float* pathy = pathx + nv.path_cap;
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
  const int n_max = astar ? nx * 4 : nx * ny / 2;
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
      int minp = (int)potarr[stc];
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

void launch_navfn_costmap(const NavfnDev& nv, uint32_t first, uint32_t count, const uint8_t* cmap, size_t stride, int cost_mode, int allow_unknown,
                          hipStream_t s) {
  hipLaunchKernelGGL(k_navfn_costmap, dim3((nv.ns + 255) / 256, count), dim3(256), 0, s, nv, first, cmap, stride, cost_mode, allow_unknown);
}
void launch_navfn_plan(const NavfnDev& nv, uint32_t first, uint32_t count, const int32_t* goals, const int32_t* starts, int astar, int at_start,
                       hipStream_t s) {
  hipLaunchKernelGGL(k_navfn_plan, dim3(count), dim3(256), 0, s, nv, first, goals, starts, astar, at_start);
}

}  // namespace navgpu
