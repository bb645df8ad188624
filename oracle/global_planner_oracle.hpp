// ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the global_planner package's potential expansion and traceback
// (second half): Expander / DijkstraExpansion / AStarExpansion, PotentialCalculator /
// QuadraticCalculator, GradientPath / GridPath, and the steps of GlobalPlanner::makePlan between worldToMap and
// getPlanFromPotential (outlineMap, calculatePotentials with nx * ny * 2 cycles, clearEndpoint).  Every function cites the
// reference lines it follows (global_planner/src, global_planner/include/global_planner).  The reference sources need
// ROS headers (planner_core.h pulls in costmap_2d_ros, nav_core, dynamic_reconfigure) and are not compiled here; the
// reference holds no test for this package, so this restatement is pinned by reading only ("parity unpinned" for the
// global_planner half of f-4; the navfn half is pinned by navfn/test/path_calc_test.cpp, see navfn_oracle.hpp).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

namespace oracle {
#ifndef ORACLE_TRUNC_X86
#define ORACLE_TRUNC_X86
// `int minp = potarr[stc]` with potarr[stc] == POT_HIGH (1e10) is out of int range: undefined in C++, and on every amd64
// build of the reference it is cvttss2si's "integer indefinite" 0x80000000.  Stated explicitly so the oracle does not
// depend on how this compiler folds the conversion; the HIP path restates the same value (its own cvt saturates).
static inline int truncX86(float v) { return (v >= -2147483648.f && v < 2147483648.f) ? (int)v : (int)0x80000000; }
#endif

struct GlobalPlannerParams {  // planner_core.cpp:105-152 defaults
  int use_dijkstra = 1, use_quadratic = 1, use_grid_path = 0, old_navfn_behavior = 0;
  int allow_unknown = 1;
  int lethal_cost = 253, neutral_cost = 50;
  float cost_factor = 3.0f;
  int outline_map = 1;  // makePlan always outlines (planner_core.cpp:296)
};

struct GlobalPlannerOracle {
  static constexpr float kPotHigh = 1.0e10f;  // planner_core.h:40
  static constexpr int kPriorityBufSize = 10000;
  int nx, ny, ns;
  GlobalPlannerParams p;
  std::vector<uint8_t> costs;
  std::vector<float> potential, gradx, grady;
  std::vector<uint8_t> pending;
  std::vector<std::pair<float, float>> path;
  int cycles_used = 0;
  GlobalPlannerOracle(int xs, int ys, const GlobalPlannerParams& par) : nx(xs), ny(ys), ns(xs * ys), p(par) {
    costs.assign(ns, 0);
    potential.assign(ns, 0.f);
    gradx.assign(ns, 0.f);
    grady.assign(ns, 0.f);
    pending.assign(ns, 0);
  }
  float convertOffset() const { return p.old_navfn_behavior ? 0.0f : 0.5f; }  // planner_core.cpp:108-111
  // GlobalPlanner::outlineMap (planner_core.cpp:62-76)
  void outlineMap(uint8_t value) {
    for (int i = 0; i < nx; i++) costs[i] = value;
    for (int i = 0; i < nx; i++) costs[(ny - 1) * nx + i] = value;
    for (int i = 0; i < ny; i++) costs[i * nx] = value;
    for (int i = 0; i < ny; i++) costs[i * nx + nx - 1] = value;
  }
  // PotentialCalculator::calculatePotential (potential_calculator.h:50-59) / QuadraticCalculator (quadratic_calculator.cpp:41-77);
  // the cost parameter is an unsigned char in both
  float calculatePotential(uint8_t cost, int n, float prev_potential = -1) const {
    if (!p.use_quadratic) {
      if (prev_potential < 0) {
        const float min_h = std::min(potential[n - 1], potential[n + 1]), min_v = std::min(potential[n - nx], potential[n + nx]);
        prev_potential = std::min(min_h, min_v);
      }
      return prev_potential + cost;
    }
    const float l = potential[n - 1], r = potential[n + 1], u = potential[n - nx], d = potential[n + nx];
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
    const float v = -0.2301 * dd * dd + 0.5307 * dd + 0.7040;
    return ta + hf * v;
  }
  // DijkstraExpansion::getCost (dijkstra.h:78-87)
  float getCost(int n) const {
    float c = costs[n];
    if (c < p.lethal_cost - 1 || (p.allow_unknown && c == 255)) {
      c = c * p.cost_factor + p.neutral_cost;
      if (c >= p.lethal_cost) c = p.lethal_cost - 1;
      return c;
    }
    return p.lethal_cost;
  }
  // DijkstraExpansion::calculatePotentials + updateCell (dijkstra.cpp:71-229)
  bool dijkstra(double start_x, double start_y, double end_x, double end_y, int cycles, bool precise) {
    std::vector<int> b1(kPriorityBufSize), b2(kPriorityBufSize), b3(kPriorityBufSize);
    int *cur = b1.data(), *nxt = b2.data(), *ovr = b3.data();
    int curE = 0, nxtE = 0, ovrE = 0;
    float threshold = p.lethal_cost;
    const float priorityIncrement = 2 * p.neutral_cost;
    std::fill(pending.begin(), pending.end(), 0);
    std::fill(potential.begin(), potential.end(), kPotHigh);
    auto push = [&](int* buf, int& end, int n) {
      if (n >= 0 && n < ns && !pending[n] && getCost(n) < p.lethal_cost && end < kPriorityBufSize) {
        buf[end++] = n;
        pending[n] = 1;
      }
    };
    const int k = (int)start_x + nx * (int)start_y;  // toIndex(double, double): int conversion of the arguments
    if (precise) {
      double dx = start_x - (int)start_x, dy = start_y - (int)start_y;
      dx = floorf(dx * 100 + 0.5) / 100;
      dy = floorf(dy * 100 + 0.5) / 100;
      potential[k] = p.neutral_cost * 2 * dx * dy;
      potential[k + 1] = p.neutral_cost * 2 * (1 - dx) * dy;
      potential[k + nx] = p.neutral_cost * 2 * dx * (1 - dy);
      potential[k + nx + 1] = p.neutral_cost * 2 * (1 - dx) * (1 - dy);
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
    int cycle = 0;
    const int startCell = (int)end_x + nx * (int)end_y;
    for (; cycle < cycles; cycle++) {
      if (curE == 0 && nxtE == 0) {
        cycles_used = cycle;
        return false;
      }
      for (int i = 0; i < curE; i++) pending[cur[i]] = 0;
      for (int i = 0; i < curE; i++) {  // updateCell
        const int n = cur[i];
        const float c = getCost(n);
        if (c >= p.lethal_cost) continue;
        const float pot = calculatePotential((uint8_t)c, n);
        if (pot < potential[n]) {
          const float le = 0.707106781 * (float)getCost(n - 1);
          const float re = 0.707106781 * (float)getCost(n + 1);
          const float ue = 0.707106781 * (float)getCost(n - nx);
          const float de = 0.707106781 * (float)getCost(n + nx);
          potential[n] = pot;
          if (pot < threshold) {
            if (potential[n - 1] > pot + le) push(nxt, nxtE, n - 1);
            if (potential[n + 1] > pot + re) push(nxt, nxtE, n + 1);
            if (potential[n - nx] > pot + ue) push(nxt, nxtE, n - nx);
            if (potential[n + nx] > pot + de) push(nxt, nxtE, n + nx);
          } else {
            if (potential[n - 1] > pot + le) push(ovr, ovrE, n - 1);
            if (potential[n + 1] > pot + re) push(ovr, ovrE, n + 1);
            if (potential[n - nx] > pot + ue) push(ovr, ovrE, n - nx);
            if (potential[n + nx] > pot + de) push(ovr, ovrE, n + nx);
          }
        }
      }
      curE = nxtE;
      nxtE = 0;
      std::swap(cur, nxt);
      if (curE == 0) {
        threshold += priorityIncrement;
        curE = ovrE;
        ovrE = 0;
        std::swap(cur, ovr);
      }
      if (potential[startCell] < kPotHigh) break;
    }
    cycles_used = cycle;
    return cycle < cycles;
  }
  // The fixed point of DijkstraExpansion::updateCell's rule (dijkstra.cpp:170-229: getCost, then PotentialCalculator or
  // QuadraticCalculator on the cost narrowed to unsigned char) from the same seeds: a FIFO relaxation until no cell changes -
  // no 10 000-entry buffers, no push tests, no early stop.  The checker of the HIP path's tiled wavefront mode
  // (navgpu_global_planner_plan_wavefront); see NavFnOracle::propagateFixedPoint for what "fixed point" means here.
  bool dijkstraFixedPoint(double start_x, double start_y, double end_x, double end_y, bool precise) {
    std::fill(potential.begin(), potential.end(), kPotHigh);
    std::vector<int> fifo;
    std::vector<uint8_t> queued(ns, 0);
    size_t head = 0;
    auto push = [&](int n) {
      if (n >= nx && n < ns - nx && !queued[n] && getCost(n) < p.lethal_cost) {
        queued[n] = 1;
        fifo.push_back(n);
      }
    };
    const int k = (int)start_x + nx * (int)start_y;
    if (precise) {  // dijkstra.cpp:88-103
      double dx = start_x - (int)start_x, dy = start_y - (int)start_y;
      dx = floorf(dx * 100 + 0.5) / 100;
      dy = floorf(dy * 100 + 0.5) / 100;
      potential[k] = p.neutral_cost * 2 * dx * dy;
      potential[k + 1] = p.neutral_cost * 2 * (1 - dx) * dy;
      potential[k + nx] = p.neutral_cost * 2 * dx * (1 - dy);
      potential[k + nx + 1] = p.neutral_cost * 2 * (1 - dx) * (1 - dy);
      for (int c : {k, k + 1, k + nx, k + nx + 1})
        for (int d : {c, c - 1, c + 1, c - nx, c + nx}) push(d);
    } else {
      potential[k] = 0;
      for (int d : {k - 1, k + 1, k - nx, k + nx}) push(d);
    }
    while (head < fifo.size()) {
      const int n = fifo[head++];
      queued[n] = 0;
      const float pot = calculatePotential((uint8_t)getCost(n), n);
      if (pot < potential[n]) {
        potential[n] = pot;
        push(n - 1);
        push(n + 1);
        push(n - nx);
        push(n + nx);
      }
      if (head > (1u << 22)) {
        fifo.erase(fifo.begin(), fifo.begin() + head);
        head = 0;
      }
    }
    cycles_used = 0;
    return potential[(int)end_x + nx * (int)end_y] < kPotHigh;
  }
  // AStarExpansion::calculatePotentials + add (astar.cpp:46-95), std::push_heap / pop_heap with greater1
  struct Index {
    int i;
    float cost;
  };
  struct Greater1 {
    bool operator()(const Index& a, const Index& b) const { return a.cost > b.cost; }
  };
  bool astar(double start_x, double start_y, double end_x, double end_y, int cycles) {
    std::vector<Index> queue;
    const int start_i = (int)start_x + nx * (int)start_y;
    queue.push_back(Index{start_i, 0});
    std::fill(potential.begin(), potential.end(), kPotHigh);
    potential[start_i] = 0;
    const int goal_i = (int)end_x + nx * (int)end_y;
    int cycle = 0;
    auto add = [&](float prev_potential, int next_i, int ex, int ey) {
      if (next_i < 0 || next_i >= ns) return;
      if (potential[next_i] < kPotHigh) return;
      if (costs[next_i] >= p.lethal_cost && !(p.allow_unknown && costs[next_i] == 255)) return;
      potential[next_i] = calculatePotential((uint8_t)(costs[next_i] + p.neutral_cost), next_i, prev_potential);
      const int x = next_i % nx, y = next_i / nx;
      const float distance = abs(ex - x) + abs(ey - y);
      queue.push_back(Index{next_i, potential[next_i] + distance * p.neutral_cost});
      std::push_heap(queue.begin(), queue.end(), Greater1());
    };
    while (queue.size() > 0 && cycle < cycles) {
      const Index top = queue[0];
      std::pop_heap(queue.begin(), queue.end(), Greater1());
      queue.pop_back();
      const int i = top.i;
      if (i == goal_i) {
        cycles_used = cycle;
        return true;
      }
      add(potential[i], i + 1, (int)end_x, (int)end_y);
      add(potential[i], i - 1, (int)end_x, (int)end_y);
      add(potential[i], i + nx, (int)end_x, (int)end_y);
      add(potential[i], i - nx, (int)end_x, (int)end_y);
      cycle++;
    }
    cycles_used = cycle;
    return false;
  }
  // AStarExpansion's RULE relaxed to its fixed point (astar.cpp:46-106): the potential of a cell from its neighbours' with the
  // cost A* passes (costs + neutral_cost narrowed to unsigned char, :95 - no cost_factor), over the cells A* may enter (:90: below
  // lethal_cost, or NO_INFORMATION when unknown is allowed - one cost value more than Dijkstra's getCost admits), from the start
  // cell at 0.  A* itself sets a cell ONCE, when its first neighbour pops, and stops at the goal: its array lies at or above this
  // one.  What a tiled wavefront for use_dijkstra = 0 would compute; tests/test_navfn.py holds it to the path-level contract.
  bool astarFixedPoint(double start_x, double start_y, double end_x, double end_y) {
    std::fill(potential.begin(), potential.end(), kPotHigh);
    std::vector<int> fifo;
    std::vector<uint8_t> queued(ns, 0);
    size_t head = 0;
    auto push = [&](int n) {
      if (n >= nx && n < ns - nx && !queued[n] && !(costs[n] >= p.lethal_cost && !(p.allow_unknown && costs[n] == 255))) {
        queued[n] = 1;
        fifo.push_back(n);
      }
    };
    const int k = (int)start_x + nx * (int)start_y;
    potential[k] = 0;
    for (int d : {k - 1, k + 1, k - nx, k + nx}) push(d);
    while (head < fifo.size()) {
      const int n = fifo[head++];
      queued[n] = 0;
      if (n == k) continue;
      const float pot = calculatePotential((uint8_t)(costs[n] + p.neutral_cost), n);
      if (pot < potential[n]) {
        potential[n] = pot;
        push(n - 1);
        push(n + 1);
        push(n - nx);
        push(n + nx);
      }
      if (head > (1u << 22)) {
        fifo.erase(fifo.begin(), fifo.begin() + head);
        head = 0;
      }
    }
    return potential[(int)end_x + nx * (int)end_y] < kPotHigh;
  }
  // Expander::clearEndpoint (expander.h:76-89)
  void clearEndpoint(int gx, int gy, int s) {
    const int startCell = gx + nx * gy;
    for (int i = -s; i <= s; i++)
      for (int j = -s; j <= s; j++) {
        const int n = startCell + i + nx * j;
        if (potential[n] < kPotHigh) continue;
        const float c = costs[n] + p.neutral_cost;
        potential[n] = calculatePotential((uint8_t)c, n);
      }
  }
  // GradientPath::gradCell (gradient_path.cpp:268-313)
  float gradCell(int n) {
    if (gradx[n] + grady[n] > 0.0) return 1.0;
    if (n < nx || n > nx * ny - nx) return 0.0;
    const float cv = potential[n];
    float dx = 0.0, dy = 0.0;
    if (cv >= kPotHigh) {
      if (potential[n - 1] < kPotHigh)
        dx = -p.lethal_cost;
      else if (potential[n + 1] < kPotHigh)
        dx = p.lethal_cost;
      if (potential[n - nx] < kPotHigh)
        dy = -p.lethal_cost;
      else if (potential[nx + 1] < kPotHigh)  // sic (:287)
        dy = p.lethal_cost;
    } else {
      if (potential[n - 1] < kPotHigh) dx += potential[n - 1] - cv;
      if (potential[n + 1] < kPotHigh) dx += cv - potential[n + 1];
      if (potential[n - nx] < kPotHigh) dy += potential[n - nx] - cv;
      if (potential[n + nx] < kPotHigh) dy += cv - potential[n + nx];
    }
    float norm = hypot(dx, dy);
    if (norm > 0) {
      norm = 1.0 / norm;
      gradx[n] = norm * dx;
      grady[n] = norm * dy;
    }
    return norm;
  }
  // GradientPath::getPath (gradient_path.cpp:68-248)
  bool gradientPath(double start_x, double start_y, double goal_x, double goal_y) {
    path.clear();
    int stc = (int)goal_x + nx * (int)goal_y;
    float dx = goal_x - (int)goal_x, dy = goal_y - (int)goal_y;
    std::fill(gradx.begin(), gradx.end(), 0.f);
    std::fill(grady.begin(), grady.end(), 0.f);
    long c = 0;
    const long lim = (long)ns * 4;
    while (c++ < lim) {
      const double px = stc % nx + dx, py = stc / nx + dy;
      if (fabs(px - start_x) < .5 && fabs(py - start_y) < .5) {
        path.emplace_back((float)start_x, (float)start_y);
        return true;
      }
      if (stc < nx || stc > nx * ny - nx) return false;
      path.emplace_back((float)px, (float)py);
      bool oscillation_detected = false;
      const int npath = (int)path.size();
      if (npath > 2 && path[npath - 1].first == path[npath - 3].first && path[npath - 1].second == path[npath - 3].second) oscillation_detected = true;
      const int stcnx = stc + nx, stcpx = stc - nx;
      if (potential[stc] >= kPotHigh || potential[stc + 1] >= kPotHigh || potential[stc - 1] >= kPotHigh || potential[stcnx] >= kPotHigh ||
          potential[stcnx + 1] >= kPotHigh || potential[stcnx - 1] >= kPotHigh || potential[stcpx] >= kPotHigh ||
          potential[stcpx + 1] >= kPotHigh || potential[stcpx - 1] >= kPotHigh || oscillation_detected) {
        int minc = stc;
        int minp = truncX86(potential[stc]);  // sic: int (gradient_path.cpp:119)
        const int nb[8] = {stcpx - 1, stcpx, stcpx + 1, stc - 1, stc + 1, stcnx - 1, stcnx, stcnx + 1};
        for (int q = 0; q < 8; ++q)
          if (potential[nb[q]] < minp) {
            minp = potential[nb[q]];
            minc = nb[q];
          }
        stc = minc;
        dx = 0;
        dy = 0;
        if (potential[stc] >= kPotHigh) return false;
      } else {
        gradCell(stc);
        gradCell(stc + 1);
        gradCell(stcnx);
        gradCell(stcnx + 1);
        const float x1 = (1.0 - dx) * gradx[stc] + dx * gradx[stc + 1];
        const float x2 = (1.0 - dx) * gradx[stcnx] + dx * gradx[stcnx + 1];
        const float x = (1.0 - dy) * x1 + dy * x2;
        const float y1 = (1.0 - dx) * grady[stc] + dx * grady[stc + 1];
        const float y2 = (1.0 - dx) * grady[stcnx] + dx * grady[stcnx + 1];
        const float y = (1.0 - dy) * y1 + dy * y2;
        if (x == 0.0 && y == 0.0) return false;
        const float ss = 0.5f / hypot(x, y);  // pathStep_ = 0.5 (gradient_path.cpp:47)
        dx += x * ss;
        dy += y * ss;
        if (dx > 1.0) { stc++; dx -= 1.0; }
        if (dx < -1.0) { stc--; dx += 1.0; }
        if (dy > 1.0) { stc += nx; dy -= 1.0; }
        if (dy < -1.0) { stc -= nx; dy += 1.0; }
      }
    }
    return false;
  }
  // GridPath::getPath (grid_path.cpp:44-82)
  bool gridPath(double start_x, double start_y, double end_x, double end_y) {
    path.clear();
    std::pair<float, float> current((float)end_x, (float)end_y);
    const int start_index = (int)start_x + nx * (int)start_y;
    path.push_back(current);
    long c = 0;
    while ((int)current.first + nx * (int)current.second != start_index) {
      float min_val = 1e10;
      int min_x = 0, min_y = 0;
      for (int xd = -1; xd <= 1; xd++)
        for (int yd = -1; yd <= 1; yd++) {
          if (xd == 0 && yd == 0) continue;
          const int x = current.first + xd, y = current.second + yd;
          const int index = x + nx * y;
          if (potential[index] < min_val) {
            min_val = potential[index];
            min_x = x;
            min_y = y;
          }
        }
      if (min_x == 0 && min_y == 0) return false;
      current.first = min_x;
      current.second = min_y;
      path.push_back(current);
      if (c++ > (long)ns * 4) return false;
    }
    return true;
  }
  // The part of GlobalPlanner::makePlan between worldToMap and the plan assembly (planner_core.cpp:250-306): start / goal
  // are map coordinates as makePlan computes them (cell index, or (w - origin) / res - 0.5 without old_navfn_behavior).
  // Returns found_legal && getPath; the path is the traceback's own (goal first), before getPlanFromPotential reverses it.
  bool plan(const uint8_t* cmap, double start_x, double start_y, double goal_x, double goal_y, int goal_x_i, int goal_y_i, bool* found_legal,
            bool fixed_point = false) {
    memcpy(costs.data(), cmap, (size_t)ns);
    if (p.outline_map) outlineMap(254);  // costmap_2d::LETHAL_OBSTACLE
    bool legal;
    if (fixed_point && !p.use_dijkstra)
      legal = astarFixedPoint(start_x, start_y, goal_x, goal_y);
    else if (fixed_point)
      legal = dijkstraFixedPoint(start_x, start_y, goal_x, goal_y, !p.old_navfn_behavior);
    else if (p.use_dijkstra)
      legal = dijkstra(start_x, start_y, goal_x, goal_y, nx * ny * 2, !p.old_navfn_behavior);  // setPreciseStart(true) unless old behaviour (planner_core.cpp:124-127)
    else
      legal = astar(start_x, start_y, goal_x, goal_y, nx * ny * 2);
    if (!p.old_navfn_behavior) clearEndpoint(goal_x_i, goal_y_i, 2);
    if (found_legal) *found_legal = legal;
    if (!legal) {
      path.clear();
      return false;
    }
    return p.use_grid_path ? gridPath(start_x, start_y, goal_x, goal_y) : gradientPath(start_x, start_y, goal_x, goal_y);
  }
};

}  // namespace oracle
