// navgpu_navfn_*: host side of the navfn::NavFn batch (see navfn_kernels.hip and include/navgpu.h).
#include "navgpu_fleet.h"

struct navgpu_navfn {
  NavfnDev nv{};
  uint32_t n = 0;
  int device = 0;
  std::recursive_mutex mu;  // calls on one handle are serialised inside the library (as on a fleet)
  hipStream_t stream = nullptr;
  std::vector<void*> allocs;
  uint8_t* d_cmap = nullptr;   // staging for host cost maps: [n][ns_padded]
  int32_t* d_goal = nullptr;   // [n][2]
  int32_t* d_start = nullptr;  // [n][2]
  navgpu_navfn_result* h_results = nullptr;  // pinned
  double* d_xy = nullptr;      // [n][2][2] start / goal map coordinates (global_planner)
  void* d_heap = nullptr;      // [n][ns_padded] AStarExpansion's queue_, allocated when A* is first asked for
  NavfnWfStatus* h_wf_status = nullptr;  // pinned; the tiled wavefront's per-plan state as the host last read it
  int32_t *d_seed_cells = nullptr, *d_stop = nullptr;  // [n][4] / [n] the tiled wavefront's seeds and stop cells
  float* d_seed_vals = nullptr;                        // [n][4]
  std::vector<uint8_t> final_array;      // [n] which potential array holds a plan's result (1: potalt, wavefront mode only)
  template <class T>
  int alloc(T** p, size_t count) {
    void* q = nullptr;
    const size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    if (hipMalloc(&q, bytes) != hipSuccess) {
      g_last_error = "hipMalloc failed (navfn)";
      return NAVGPU_ERR_HIP;
    }
    hipMemsetAsync(q, 0, bytes, stream);
    allocs.push_back(q);
    *p = static_cast<T*>(q);
    return NAVGPU_OK;
  }
};

namespace {
struct NavfnGuard {  // lock + make the handle's GPU current on the calling thread
  std::lock_guard<std::recursive_mutex> lk;
  explicit NavfnGuard(navgpu_navfn* h) : lk(h->mu) { (void)hipSetDevice(h->device); }
};
}  // namespace

extern "C" {

int navgpu_navfn_create(uint32_t nx, uint32_t ny, uint32_t n_plans, int32_t device, navgpu_navfn** out) {
  if (!out || nx < 3 || ny < 3 || nx > 32768 || ny > 32768 || !n_plans) return NAVGPU_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    g_last_error = "no usable HIP device (navgpu has no CPU fallback)";
    return NAVGPU_ERR_NO_DEVICE;
  }
  HIP_TRY(hipSetDevice(device));
  navgpu_navfn* h = new navgpu_navfn();
  h->n = n_plans;
  h->device = device;
  h->final_array.assign(n_plans, 0);
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    g_last_error = "hipStreamCreate failed";
    return NAVGPU_ERR_HIP;
  }
  NavfnDev& nv = h->nv;
  nv.nx = (int)nx;
  nv.ny = (int)ny;
  nv.ns = (int)(nx * ny);
  nv.ns_padded = (uint32_t)((nv.ns + 63) & ~63);
  nv.path_cap = (uint32_t)std::max(nv.ns / 2, 4 * nv.nx) + 4;  // calcPath(nx * ny / 2) | calcPath(nx * 4)
  int rc = 0;
#define A(ptr, cnt)                                  \
  if (!rc) rc = h->alloc(&(ptr), (size_t)(cnt));
  A(nv.costarr, (size_t)n_plans * nv.ns_padded);
  A(nv.pending, (size_t)n_plans * nv.ns_padded);
  A(nv.potarr, (size_t)n_plans * nv.ns_padded);
  A(nv.gradx, (size_t)n_plans * nv.ns_padded);
  A(nv.grady, (size_t)n_plans * nv.ns_padded);
  A(nv.pb, (size_t)n_plans * 3 * 10000);
  A(nv.path, (size_t)n_plans * 2 * nv.path_cap);
  A(nv.results, n_plans);
  A(h->d_cmap, (size_t)n_plans * nv.ns_padded);
  A(h->d_goal, (size_t)n_plans * 2);
  A(h->d_start, (size_t)n_plans * 2);
  A(h->d_xy, (size_t)n_plans * 4);
#undef A
  if (!rc && hipHostMalloc((void**)&h->h_results, sizeof(navgpu_navfn_result) * n_plans, hipHostMallocDefault) != hipSuccess) rc = NAVGPU_ERR_HIP;
  if (rc) {
    navgpu_navfn_destroy(h);
    return rc;
  }
  HIP_TRY(waitStream(h->stream));
  *out = h;
  return NAVGPU_OK;
}

int navgpu_navfn_destroy(navgpu_navfn* h) {
  if (!h) return NAVGPU_ERR_INVALID;
  (void)hipSetDevice(h->device);
  if (h->stream) waitStream(h->stream);
  for (void* p : h->allocs) hipFree(p);
  if (h->h_results) hipHostFree(h->h_results);
  if (h->h_wf_status) hipHostFree(h->h_wf_status);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
  return NAVGPU_OK;
}

static bool navfnRange(const navgpu_navfn* h, uint32_t first, uint32_t count) { return count > 0 && first < h->n && count <= h->n - first; }

int navgpu_navfn_set_costmap(navgpu_navfn* h, uint32_t first, uint32_t count, const uint8_t* cmap, int32_t shared, int32_t cost_mode,
                             int32_t allow_unknown) {
  if (!h || !cmap || !navfnRange(h, first, count) || cost_mode < 0 || cost_mode > 2) return NAVGPU_ERR_INVALID;
  NavfnGuard guard_(h);
  const NavfnDev& nv = h->nv;
  const uint32_t maps = shared ? 1u : count;
  for (uint32_t k = 0; k < maps; ++k)
    HIP_TRY(hipMemcpyAsync(h->d_cmap + (size_t)k * nv.ns_padded, cmap + (size_t)k * nv.ns, (size_t)nv.ns, hipMemcpyHostToDevice, h->stream));
  launch_navfn_costmap(nv, first, count, h->d_cmap, shared ? 0 : nv.ns_padded, cost_mode, allow_unknown, h->stream);
  HIP_TRY(waitStream(h->stream));  // the caller's buffer and the staging area are free again
  return checkLaunch();
}

int navgpu_navfn_set_costmap_from_fleet(navgpu_navfn* h, uint32_t first, uint32_t count, navgpu_fleet* f, uint32_t fleet_first, int32_t allow_unknown) {
  if (!h || !f || !navfnRange(h, first, count) || !f->rangeOk(fleet_first, count)) return NAVGPU_ERR_INVALID;
  if ((int)f->cm.nx != h->nv.nx || (int)f->cm.ny != h->nv.ny || f->desc.device != h->device) return NAVGPU_ERR_INVALID;
  NavfnGuard guard_(h);
  FleetGuard fleet_guard_(f);
  HIP_TRY(waitStream(f->stream));  // the fleet's last costmap update has landed
  launch_navfn_costmap(h->nv, first, count, f->cm.master + (size_t)fleet_first * f->cm.cells_padded, f->cm.cells_padded, 1, allow_unknown, h->stream);
  HIP_TRY(waitStream(h->stream));
  return checkLaunch();
}

int navgpu_navfn_plan(navgpu_navfn* h, uint32_t first, uint32_t count, const int32_t* goals, const int32_t* starts, int32_t astar, int32_t at_start,
                      navgpu_navfn_result* results) {
  if (!h || !goals || !starts || !navfnRange(h, first, count)) return NAVGPU_ERR_INVALID;
  NavfnGuard guard_(h);
  const NavfnDev& nv = h->nv;
  for (uint32_t k = 0; k < count; ++k) {  // the reference indexes its arrays with these without a check: keep them inside the border
    const int32_t* g = goals + 2 * k;
    const int32_t* s = starts + 2 * k;
    if (g[0] < 1 || g[1] < 1 || g[0] > nv.nx - 2 || g[1] > nv.ny - 2 || s[0] < 0 || s[1] < 0 || s[0] >= nv.nx || s[1] >= nv.ny) {
      g_last_error = "navgpu_navfn_plan: goal / start cell outside the map";
      return NAVGPU_ERR_INVALID;
    }
  }
  HIP_TRY(hipMemcpyAsync(h->d_goal, goals, sizeof(int32_t) * 2 * count, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_start, starts, sizeof(int32_t) * 2 * count, hipMemcpyHostToDevice, h->stream));
  launch_navfn_plan(nv, first, count, h->d_goal, h->d_start, astar ? 1 : 0, at_start ? 1 : 0, h->stream);
  std::fill(h->final_array.begin() + first, h->final_array.begin() + first + count, (uint8_t)0);
  HIP_TRY(hipMemcpyAsync(h->h_results + first, nv.results + first, sizeof(navgpu_navfn_result) * count, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(waitStream(h->stream));
  if (results) memcpy(results, h->h_results + first, sizeof(navgpu_navfn_result) * count);
  return checkLaunch();
}

// The expansion as a tiled wavefront (navfn_kernels.hip: k_navfn_wf_*): rounds are queued a batch at a time, the per-plan
// status (done / which array / rounds) is read between batches; a round launched after its plan has finished leaves at once.
// seeds = count x 4 (cell, value) pairs (cell < 0: unused), stop_cells = count cells whose potential ends the search.
static int runWavefront(navgpu_navfn* h, uint32_t first, uint32_t count, const NavfnWfRule& rule_in, const int32_t* seed_cells, const float* seed_vals,
                        const int32_t* stop_cells, int at_start) {
  NavfnDev& nv = h->nv;
  constexpr int kTile = 32, kMaxRounds = 8192, kBatch = 16;
  NavfnWfRule rule = rule_in;
  rule.max_sweeps = NAVGPU_DEBUG_ENV("NAVGPU_DEBUG_WF_SWEEPS") ? std::max(1, atoi(NAVGPU_DEBUG_ENV("NAVGPU_DEBUG_WF_SWEEPS"))) : 32;  // (a tile the front crosses once settles within 32; measured: 4 ... 160 give the same plan time)
  if (!nv.potalt) {
    nv.wf_tiles_x = (nv.nx + kTile - 1) / kTile;
    nv.wf_tiles_y = (nv.ny + kTile - 1) / kTile;
    nv.wf_max_rounds = kMaxRounds;
    // each buffer is allocated at most once (a call that failed half-way leaves what it got for the next one to complete)
    int rc = 0;
    if (!rc && !nv.wf_act) rc = h->alloc(&nv.wf_act, (size_t)h->n * 2 * nv.wf_tiles_x * nv.wf_tiles_y);
    if (!rc && !nv.wf_nchg) rc = h->alloc(&nv.wf_nchg, (size_t)h->n * kMaxRounds);
    if (!rc && !nv.wf_min) rc = h->alloc(&nv.wf_min, (size_t)h->n * kMaxRounds);
    if (!rc && !nv.wf_status) rc = h->alloc(&nv.wf_status, h->n);
    if (!rc && !h->d_seed_cells) rc = h->alloc(&h->d_seed_cells, (size_t)h->n * 4);
    if (!rc && !h->d_seed_vals) rc = h->alloc(&h->d_seed_vals, (size_t)h->n * 4);
    if (!rc && !h->d_stop) rc = h->alloc(&h->d_stop, h->n);
    if (!rc && !h->h_wf_status && hipHostMalloc((void**)&h->h_wf_status, sizeof(NavfnWfStatus) * h->n, hipHostMallocDefault) != hipSuccess) {
      h->h_wf_status = nullptr;
      rc = NAVGPU_ERR_HIP;
    }
    if (!rc) rc = h->alloc(&nv.potalt, (size_t)h->n * nv.ns_padded);  // last: its presence says the others exist
    if (rc) return rc;
  }
  const size_t tiles = (size_t)nv.wf_tiles_x * nv.wf_tiles_y;
  HIP_TRY(hipMemcpyAsync(h->d_seed_cells, seed_cells, sizeof(int32_t) * 4 * count, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_seed_vals, seed_vals, sizeof(float) * 4 * count, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_stop, stop_cells, sizeof(int32_t) * count, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemsetAsync(nv.wf_act + (size_t)first * 2 * tiles, 0, sizeof(uint32_t) * 2 * tiles * count, h->stream));
  HIP_TRY(hipMemsetAsync(nv.wf_nchg + (size_t)first * kMaxRounds, 0, sizeof(uint32_t) * kMaxRounds * (size_t)count, h->stream));
  HIP_TRY(hipMemsetAsync(nv.wf_min + (size_t)first * kMaxRounds, 0xFF, sizeof(uint32_t) * kMaxRounds * (size_t)count, h->stream));
  HIP_TRY(hipMemsetAsync(nv.wf_status + first, 0, sizeof(NavfnWfStatus) * count, h->stream));
  launch_navfn_wf_init(nv, first, count, rule, h->d_seed_cells, h->d_seed_vals, h->stream);
  bool all_done = false;
  for (int round = 0; round < kMaxRounds && !all_done;) {
    for (int b = 0; b < kBatch && round < kMaxRounds; ++b, ++round) launch_navfn_wf_round(nv, first, count, rule, h->d_stop, at_start ? 1 : 0, round, h->stream);
    HIP_TRY(hipMemcpyAsync(h->h_wf_status + first, nv.wf_status + first, sizeof(NavfnWfStatus) * count, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(waitStream(h->stream));  // (the first one also frees the caller-side staging vectors)
    all_done = true;
    for (uint32_t k = 0; k < count; ++k) all_done = all_done && h->h_wf_status[first + k].done;
  }
  if (!all_done) {  // no plan: navgpu_navfn_path must not hand out the previous call's
    for (uint32_t k = 0; k < count; ++k) {
      h->h_results[first + k].found = 0;
      h->h_results[first + k].path_length = 0;
    }
    g_last_error = "tiled wavefront: not settled within 8192 rounds";
    return NAVGPU_ERR_CAPACITY;
  }
  for (uint32_t k = 0; k < count; ++k) h->final_array[first + k] = (uint8_t)h->h_wf_status[first + k].final_array;
  return NAVGPU_OK;
}

int navgpu_navfn_plan_wavefront(navgpu_navfn* h, uint32_t first, uint32_t count, const int32_t* goals, const int32_t* starts, int32_t at_start,
                                navgpu_navfn_result* results) {
  if (!h || !goals || !starts || !navfnRange(h, first, count)) return NAVGPU_ERR_INVALID;
  NavfnGuard guard_(h);
  NavfnDev& nv = h->nv;
  std::vector<int32_t> seed_cells((size_t)count * 4, -1), stop(count);
  std::vector<float> seed_vals((size_t)count * 4, 0.0f);
  for (uint32_t k = 0; k < count; ++k) {
    const int32_t* g = goals + 2 * k;
    const int32_t* s = starts + 2 * k;
    if (g[0] < 1 || g[1] < 1 || g[0] > nv.nx - 2 || g[1] > nv.ny - 2 || s[0] < 0 || s[1] < 0 || s[0] >= nv.nx || s[1] >= nv.ny) {
      g_last_error = "navgpu_navfn_plan_wavefront: goal / start cell outside the map";
      return NAVGPU_ERR_INVALID;
    }
    seed_cells[4 * k] = g[0] + g[1] * nv.nx;  // initCost(goal, 0)
    stop[k] = s[0] + s[1] * nv.nx;
  }
  NavfnWfRule rule{};
  rule.quadratic = 1;
  rule.outline = 1;
  HIP_TRY(hipMemcpyAsync(h->d_goal, goals, sizeof(int32_t) * 2 * count, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_start, starts, sizeof(int32_t) * 2 * count, hipMemcpyHostToDevice, h->stream));
  int rc = runWavefront(h, first, count, rule, seed_cells.data(), seed_vals.data(), stop.data(), at_start);
  if (rc) return rc;
  launch_navfn_wf_path(nv, first, count, h->d_goal, h->d_start, h->stream);
  HIP_TRY(hipMemcpyAsync(h->h_results + first, nv.results + first, sizeof(navgpu_navfn_result) * count, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(waitStream(h->stream));
  if (results) memcpy(results, h->h_results + first, sizeof(navgpu_navfn_result) * count);
  return checkLaunch();
}

static int gpValidate(const navgpu_navfn* h, uint32_t count, const navgpu_global_planner_params* gp, const double* starts, const double* goals,
                      const int32_t* goal_cells) {
  const NavfnDev& nv = h->nv;
  if (gp->lethal_cost < 2 || gp->lethal_cost > 255 || gp->neutral_cost < 0 || gp->neutral_cost > 255) return NAVGPU_ERR_INVALID;
  for (uint32_t k = 0; k < count; ++k) {  // the reference indexes its arrays with these without a check: keep them inside the outline
    const double sx = starts[2 * k], sy = starts[2 * k + 1], gx = goals[2 * k], gy = goals[2 * k + 1];
    const int32_t gi = goal_cells[2 * k], gj = goal_cells[2 * k + 1];
    if (!(sx >= 2 && sy >= 2 && sx < nv.nx - 3 && sy < nv.ny - 3 && gx >= 1 && gy >= 1 && gx < nv.nx - 1 && gy < nv.ny - 1) || gi < 0 || gj < 0 ||
        gi >= nv.nx || gj >= nv.ny) {
      g_last_error = "navgpu_global_planner_plan: start / goal too close to the map border";
      return NAVGPU_ERR_INVALID;
    }
  }
  return NAVGPU_OK;
}

// GlobalPlanner::makePlan's core with DijkstraExpansion run as the tiled wavefront (use_dijkstra only); clearEndpoint and the
// traceback are the reference-order kernel's own code on one lane
int navgpu_global_planner_plan_wavefront(navgpu_navfn* h, uint32_t first, uint32_t count, const navgpu_global_planner_params* gp, const double* starts,
                                         const double* goals, const int32_t* goal_cells, navgpu_navfn_result* results) {
  if (!h || !gp || !starts || !goals || !goal_cells || !navfnRange(h, first, count)) return NAVGPU_ERR_INVALID;
  NavfnGuard guard_(h);
  if (!gp->use_dijkstra) {
    g_last_error = "navgpu_global_planner_plan_wavefront: AStarExpansion has no fixed-point reading; use navgpu_global_planner_plan";
    return NAVGPU_ERR_INVALID;
  }
  int rc = gpValidate(h, count, gp, starts, goals, goal_cells);
  if (rc) return rc;
  NavfnDev& nv = h->nv;
  std::vector<int32_t> seed_cells((size_t)count * 4, -1), stop(count);
  std::vector<float> seed_vals((size_t)count * 4, 0.0f);
  for (uint32_t q = 0; q < count; ++q) {
    const double start_x = starts[2 * q], start_y = starts[2 * q + 1];
    const int k = (int)start_x + nv.nx * (int)start_y;  // toIndex(double, double)
    if (!gp->old_navfn_behavior) {                      // setPreciseStart(true) (planner_core.cpp:124-127, dijkstra.cpp:88-103)
      double dx = start_x - (int)start_x, dy = start_y - (int)start_y;
      dx = floorf((float)(dx * 100 + 0.5)) / 100;
      dy = floorf((float)(dy * 100 + 0.5)) / 100;
      const int cells[4] = {k, k + 1, k + nv.nx, k + nv.nx + 1};
      const float vals[4] = {(float)(gp->neutral_cost * 2 * dx * dy), (float)(gp->neutral_cost * 2 * (1 - dx) * dy),
                             (float)(gp->neutral_cost * 2 * dx * (1 - dy)), (float)(gp->neutral_cost * 2 * (1 - dx) * (1 - dy))};
      for (int i = 0; i < 4; ++i) {
        seed_cells[4 * q + i] = cells[i];
        seed_vals[4 * q + i] = vals[i];
      }
    } else {
      seed_cells[4 * q] = k;
    }
    stop[q] = (int)goals[2 * q] + nv.nx * (int)goals[2 * q + 1];
  }
  NavfnWfRule rule{};
  rule.global_planner = 1;
  rule.quadratic = gp->use_quadratic ? 1 : 0;
  rule.outline = gp->outline_map ? 1 : 0;
  rule.allow_unknown = gp->allow_unknown ? 1 : 0;
  rule.lethal_cost = gp->lethal_cost;
  rule.neutral_cost = gp->neutral_cost;
  rule.cost_factor = gp->cost_factor;
  HIP_TRY(hipMemcpyAsync(h->d_xy, starts, sizeof(double) * 2 * count, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_xy + (size_t)2 * h->n, goals, sizeof(double) * 2 * count, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_goal, goal_cells, sizeof(int32_t) * 2 * count, hipMemcpyHostToDevice, h->stream));
  rc = runWavefront(h, first, count, rule, seed_cells.data(), seed_vals.data(), stop.data(), 1);
  if (rc) return rc;
  launch_gp_wf_finish(nv, first, count, *gp, h->d_xy, h->d_xy + (size_t)2 * h->n, h->d_goal, h->stream);
  HIP_TRY(hipMemcpyAsync(h->h_results + first, nv.results + first, sizeof(navgpu_navfn_result) * count, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(waitStream(h->stream));
  if (results) memcpy(results, h->h_results + first, sizeof(navgpu_navfn_result) * count);
  return checkLaunch();
}

int navgpu_global_planner_plan(navgpu_navfn* h, uint32_t first, uint32_t count, const navgpu_global_planner_params* gp, const double* starts,
                               const double* goals, const int32_t* goal_cells, navgpu_navfn_result* results) {
  if (!h || !gp || !starts || !goals || !goal_cells || !navfnRange(h, first, count)) return NAVGPU_ERR_INVALID;
  NavfnGuard guard_(h);
  const NavfnDev& nv = h->nv;
  {
    int rc = gpValidate(h, count, gp, starts, goals, goal_cells);
    if (rc) return rc;
  }
  if (!gp->use_dijkstra && !h->d_heap) {
    uint64_t* q = nullptr;  // 8 bytes per entry (int index, float cost); a cell enters the queue at most once
    int rc = h->alloc(&q, (size_t)h->n * nv.ns_padded);
    if (rc) return rc;
    h->d_heap = q;
  }
  HIP_TRY(hipMemcpyAsync(h->d_xy, starts, sizeof(double) * 2 * count, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_xy + (size_t)2 * h->n, goals, sizeof(double) * 2 * count, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_goal, goal_cells, sizeof(int32_t) * 2 * count, hipMemcpyHostToDevice, h->stream));
  launch_gp_plan(nv, first, count, *gp, h->d_xy, h->d_xy + (size_t)2 * h->n, h->d_goal, h->d_heap, h->stream);
  std::fill(h->final_array.begin() + first, h->final_array.begin() + first + count, (uint8_t)0);
  HIP_TRY(hipMemcpyAsync(h->h_results + first, nv.results + first, sizeof(navgpu_navfn_result) * count, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(waitStream(h->stream));
  if (results) memcpy(results, h->h_results + first, sizeof(navgpu_navfn_result) * count);
  return checkLaunch();
}

int navgpu_navfn_path(navgpu_navfn* h, uint32_t plan, float* xy, uint32_t cap) {
  if (!h || plan >= h->n || (!xy && cap)) return NAVGPU_ERR_INVALID;
  NavfnGuard guard_(h);
  const NavfnDev& nv = h->nv;
  const int len = h->h_results[plan].path_length;
  const uint32_t n = std::min<uint32_t>((uint32_t)std::max(len, 0), cap);
  if (n) {
    std::vector<float> px(n), py(n);
    HIP_TRY(hipMemcpyAsync(px.data(), nv.path + (size_t)plan * 2 * nv.path_cap, sizeof(float) * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(py.data(), nv.path + (size_t)plan * 2 * nv.path_cap + nv.path_cap, sizeof(float) * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(waitStream(h->stream));
    for (uint32_t i = 0; i < n; ++i) {
      xy[2 * i] = px[i];
      xy[2 * i + 1] = py[i];
    }
  }
  return len;
}

int navgpu_navfn_potential(navgpu_navfn* h, uint32_t plan, float* potarr) {
  if (!h || plan >= h->n || !potarr) return NAVGPU_ERR_INVALID;
  NavfnGuard guard_(h);
  HIP_TRY(hipMemcpyAsync(potarr, (h->final_array[plan] ? h->nv.potalt : h->nv.potarr) + (size_t)plan * h->nv.ns_padded, sizeof(float) * (size_t)h->nv.ns, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(waitStream(h->stream));
  return NAVGPU_OK;
}

}  // extern "C"
