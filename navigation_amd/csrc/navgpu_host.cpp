// C-ABI host side of libnavgpu.so (include/navgpu.h).  Owns device memory, the fleet's HIP
// stream and the launch sequence; the only arithmetic done here is the per-cycle, per-instance
// scalar bookkeeping the reference also does once per cycle on the host (footprint transform,
// nose goal, inflation cost table) — in fp64 with libm, like the reference.
// There is no CPU fallback: every data-parallel step is a HIP kernel.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "navgpu_device.h"

using namespace navgpu;

namespace {
thread_local std::string g_last_error;

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      g_last_error = std::string(#expr) + ": " + hipGetErrorString(e_);                            \
      return NAVGPU_ERR_HIP;                                                                       \
    }                                                                                              \
  } while (0)

// hipStreamSynchronize also flushes the runtime's batched launches; a hipStreamQuery polling loop does
// not (measured: the queue then only drains when the poll gives up), so no polling here.
static hipError_t waitStream(hipStream_t s) { return hipStreamSynchronize(s); }

struct EventPair {
  int kernel;
  hipEvent_t a, b;
};
}  // namespace

struct navgpu_fleet {
  navgpu_fleet_desc desc{};
  hipStream_t stream = nullptr;
  CostmapDev cm{};
  PlannerDev pl{};
  bool planner_configured = false, inflation_configured = false, planner_staged = false;
  bool shift_pending = false;                   // rolling window: staged origins not yet applied to the grids
  uint32_t shift_first = 0, shift_count = 0;
  std::vector<void*> allocs;
  // host mirrors
  std::vector<double> h_origin;                 // [n][2]
  std::vector<double> h_fp_spec;                // [n][kMaxFootprint][2]
  std::vector<uint32_t> h_fp_n;                 // [n]
  navgpu_inflation_params infl{};
  navgpu_obstacle_params obsp{};
  double fp_radius = 0.0;                       // largest vertex distance over all instances
  // pinned host mirrors of the per-cycle staging arrays (full fleet size): H2D copies are truly
  // asynchronous and no per-call allocation happens on the staging path
  std::vector<void*> pinned;
  ObsCsr* hp_obs = nullptr;
  uint32_t *hp_cnt = nullptr, *hp_used = nullptr, *hp_plan_cnt = nullptr;
  float* hp_pts = nullptr;
  double *hp_fpw = nullptr, *hp_pose = nullptr, *hp_plan = nullptr, *hp_front = nullptr;
  int32_t *hp_shift = nullptr, *hp_align = nullptr;
  navgpu_robot_state* hp_state = nullptr;
  navgpu_plan_result* hp_result = nullptr;
  // DWAPlannerROS mirror (navgpu_local_planner_*): per-instance controller state, host only
  struct LocalPlannerState {
    std::vector<double> plan;      // stored global plan, (x, y, yaw) triples in the plan's frame (prunePlan shrinks it)
    double T[3] = {0, 0, 0};       // planar plan -> global transform
    bool has_T = false, have_plan = false;
    bool xy_tolerance_latch = false, rotating_to_goal = false;  // LatchedStopRotateController members
  };
  std::vector<LocalPlannerState> lp;
  navgpu_local_limits lp_limits{};
  bool lp_configured = false;
  // legacy TrajectoryPlanner (navgpu_tp_*)
  struct TpHost {
    std::vector<double> plan;  // global_plan_ as x, y pairs
    double final_goal_x = 0, final_goal_y = 0;
    bool final_goal_position_valid = false;
    navgpu_tp_state st{};
    std::vector<navgpu_tp_sample> made;  // the generateTrajectory calls of the last cycle, in call order
    int n_points = 0;                    // of the winner
  };
  TpDev tp{};
  bool tp_configured = false;
  std::vector<TpHost> tph;
  std::vector<double> tp_h_samples, tp_h_start;
  std::vector<TpOut> tp_h_out;
  std::vector<uint32_t> tp_h_nsamples, tp_h_within, tp_h_within_count;
  std::vector<int32_t> tp_h_winner;
  // scratch device buffers
  double* d_bounds_tmp = nullptr;               // [n][4]
  int32_t* d_boxes_tmp = nullptr;               // [n][4]
  float* d_explicit = nullptr;                  // [3]
  int8_t* d_occ = nullptr;
  // profiling
  bool profiling = false;
  std::vector<EventPair> events;
  std::vector<EventPair> free_events;
  double prof_ms[NAVGPU_K_COUNT] = {0};
  uint64_t prof_n[NAVGPU_K_COUNT] = {0};

  template <class T>
  int alloc(T** p, size_t count) {
    void* q = nullptr;
    size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) {
      g_last_error = std::string("hipMalloc: ") + hipGetErrorString(e);
      return NAVGPU_ERR_HIP;
    }
    e = hipMemsetAsync(q, 0, bytes, stream);
    if (e != hipSuccess) {
      g_last_error = std::string("hipMemsetAsync: ") + hipGetErrorString(e);
      return NAVGPU_ERR_HIP;
    }
    allocs.push_back(q);
    *p = static_cast<T*>(q);
    return NAVGPU_OK;
  }
  template <class T>
  int allocPinned(T** p, size_t count) {
    void* q = nullptr;
    hipError_t e = hipHostMalloc(&q, std::max<size_t>(count * sizeof(T), 16), hipHostMallocDefault);
    if (e != hipSuccess) {
      g_last_error = std::string("hipHostMalloc: ") + hipGetErrorString(e);
      return NAVGPU_ERR_HIP;
    }
    memset(q, 0, std::max<size_t>(count * sizeof(T), 16));
    pinned.push_back(q);
    *p = static_cast<T*>(q);
    return NAVGPU_OK;
  }
  void release(void* q) {
    if (!q) return;
    auto it = std::find(allocs.begin(), allocs.end(), q);
    if (it != allocs.end()) allocs.erase(it);
    hipFree(q);
  }
  bool rangeOk(uint32_t first, uint32_t count) const { return count > 0 && first < desc.n_instances && count <= desc.n_instances - first; }

  int beginKernel(int k, EventPair* ep) {
    if (!profiling) return NAVGPU_OK;
    if (free_events.empty()) {
      EventPair n{};
      HIP_TRY(hipEventCreate(&n.a));
      HIP_TRY(hipEventCreate(&n.b));
      free_events.push_back(n);
    }
    *ep = free_events.back();
    free_events.pop_back();
    ep->kernel = k;
    HIP_TRY(hipEventRecord(ep->a, stream));
    return NAVGPU_OK;
  }
  int endKernel(EventPair* ep) {
    if (!profiling) return NAVGPU_OK;
    HIP_TRY(hipEventRecord(ep->b, stream));
    events.push_back(*ep);
    if (events.size() > 8192) return foldEvents();
    return NAVGPU_OK;
  }
  int foldEvents() {
    if (events.empty()) return NAVGPU_OK;
    HIP_TRY(waitStream(stream));
    for (auto& e : events) {
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, e.a, e.b));
      prof_ms[e.kernel] += ms;
      prof_n[e.kernel] += 1;
      free_events.push_back(e);
    }
    events.clear();
    return NAVGPU_OK;
  }
};

#define PROFILED(fleet, kid, launch_expr)            \
  do {                                               \
    EventPair ep_{};                                 \
    int rc_ = (fleet)->beginKernel((kid), &ep_);     \
    if (rc_) return rc_;                             \
    launch_expr;                                     \
    rc_ = (fleet)->endKernel(&ep_);                  \
    if (rc_) return rc_;                             \
  } while (0)

static int checkLaunch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_last_error = std::string("kernel launch: ") + hipGetErrorString(e);
    return NAVGPU_ERR_HIP;
  }
  return NAVGPU_OK;
}

extern "C" {

const char* navgpu_version(void) { return "navgpu 0.1 (gfx950)"; }
const char* navgpu_strerror(int status) {
  switch (status) {
    case NAVGPU_OK: return "ok";
    case NAVGPU_ERR_INVALID: return "invalid argument";
    case NAVGPU_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case NAVGPU_ERR_HIP: return "HIP runtime error";
    case NAVGPU_ERR_CAPACITY: return "input exceeds a capacity given at fleet creation";
    case NAVGPU_ERR_STATE: return "call sequence violated";
    default: return "unknown status";
  }
}
const char* navgpu_last_error(void) { return g_last_error.c_str(); }
int navgpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
const char* navgpu_kernel_name(int32_t k) {
  static const char* names[NAVGPU_K_COUNT] = {"k_obstacle", "k_merge", "k_inflate", "k_bfs", "k_score", "k_select"};
  return (k >= 0 && k < NAVGPU_K_COUNT) ? names[k] : "?";
}

int navgpu_fleet_create(const navgpu_fleet_desc* d, navgpu_fleet** out) {
  if (!d || !out || d->n_instances == 0 || d->size_x == 0 || d->size_y == 0 || !(d->resolution > 0)) return NAVGPU_ERR_INVALID;
  if ((uint64_t)((d->size_x + 127) / 128) * ((d->size_y + 15) / 16) > 8192) {  // activity flags of k_bfs_global (kMaxTiles)
    g_last_error = "navgpu_fleet_create: costmap larger than 16.7 M cells";
    return NAVGPU_ERR_CAPACITY;
  }
  if (d->max_footprint > (uint32_t)kMaxFootprint) return NAVGPU_ERR_CAPACITY;
  if ((uint64_t)d->size_x * d->size_y > (1ull << 30) || d->size_x > 65535 || d->size_y > 65535) return NAVGPU_ERR_CAPACITY;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || d->device < 0 || d->device >= ndev) {
    g_last_error = "no HIP device";
    return NAVGPU_ERR_NO_DEVICE;
  }
  HIP_TRY(hipSetDevice(d->device));
  auto* f = new navgpu_fleet();
  f->desc = *d;
  if (f->desc.max_observations == 0) f->desc.max_observations = 1;
  if (f->desc.max_points == 0) f->desc.max_points = 1;
  if (f->desc.max_plan == 0) f->desc.max_plan = 1;
  if (f->desc.max_sim_steps == 0) f->desc.max_sim_steps = 64;
  const uint32_t n = d->n_instances;
  if (hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) != hipSuccess) {
    g_last_error = "hipStreamCreate failed";
    delete f;
    return NAVGPU_ERR_NO_DEVICE;
  }
  CostmapDev& cm = f->cm;
  cm.nx = d->size_x;
  cm.ny = d->size_y;
  cm.cells = d->size_x * d->size_y;
  cm.cells_padded = (cm.cells + 63) & ~63u;
  cm.res = d->resolution;
  cm.layers = d->layers;
  cm.track_unknown = d->track_unknown;
  cm.master_default = d->track_unknown ? kNoInfo : kFree;    // layered_costmap.cpp:53-57
  cm.obstacle_default = d->track_unknown ? kNoInfo : kFree;  // obstacle_layer.cpp:60-64
  cm.obs_enabled = 1;
  cm.footprint_clearing = 1;
  cm.combination_method = 1;
  cm.max_obstacle_height = 2.0;
  cm.z_voxels = 10;
  cm.unknown_threshold = 15;
  cm.mark_threshold = 0;
  cm.origin_z = 0.0;
  cm.z_resolution = 0.2;
  cm.max_obs = f->desc.max_observations;
  cm.max_points = f->desc.max_points;
  int rc = 0;
#define A(ptr, cnt)                        \
  if ((rc = f->alloc(&(ptr), (cnt))) != 0) { \
    navgpu_fleet_destroy(f);               \
    return rc;                             \
  }
  A(cm.origin, (size_t)n * 2);
  A(cm.master, (size_t)n * cm.cells_padded);
  if (d->layers & NAVGPU_LAYER_STATIC) A(cm.stat, (size_t)n * cm.cells_padded);
  if (d->layers & (NAVGPU_LAYER_OBSTACLE | NAVGPU_LAYER_VOXEL)) A(cm.obst, (size_t)n * cm.cells_padded);
  if (d->layers & NAVGPU_LAYER_VOXEL) A(cm.voxel, (size_t)n * cm.cells_padded);
  A(cm.lut, 66 * 66);
  A(cm.lut2, 256);
  A(cm.state, n);
  A(cm.pose, (size_t)n * 3);
  A(cm.fp_world, (size_t)n * kMaxFootprint * 2);
  A(cm.fp_n, n);
  A(cm.obs, (size_t)n * cm.max_obs);
  A(cm.obs_count, n);
  A(cm.points, (size_t)n * cm.max_points * 3);
  A(cm.shift, (size_t)n * 2);
  if (d->rolling_window) {
    if (d->layers & NAVGPU_LAYER_STATIC) {
      g_last_error = "rolling_window with a static layer needs tf (StaticLayer::updateCosts rolling branch) and is not supported";
      navgpu_fleet_destroy(f);
      return NAVGPU_ERR_INVALID;
    }
    A(cm.master_alt, (size_t)n * cm.cells_padded);
    if (cm.obst) A(cm.obst_alt, (size_t)n * cm.cells_padded);
    if (cm.voxel) A(cm.voxel_alt, (size_t)n * cm.cells_padded);
  }
  if (cm.voxel) A(cm.mark_seq, (size_t)n * cm.max_points);
#define AP(ptr, cnt)                              \
  if ((rc = f->allocPinned(&(ptr), (cnt))) != 0) { \
    navgpu_fleet_destroy(f);                       \
    return rc;                                     \
  }
  AP(f->hp_obs, (size_t)n * cm.max_obs);
  AP(f->hp_cnt, n);
  AP(f->hp_used, n);
  AP(f->hp_pts, (size_t)n * cm.max_points * 3);
  AP(f->hp_fpw, (size_t)n * kMaxFootprint * 2);
  AP(f->hp_pose, (size_t)n * 3);
  AP(f->hp_shift, (size_t)n * 2);
  AP(f->hp_state, n);
  AP(f->hp_plan, (size_t)n * f->desc.max_plan * 2);
  AP(f->hp_plan_cnt, n);
  AP(f->hp_front, (size_t)n * 2);
  AP(f->hp_align, n);
  AP(f->hp_result, n);
  f->pl.result = f->hp_result;  // k_select writes results straight into pinned host memory (72 B per robot)
#undef AP
  A(f->d_bounds_tmp, (size_t)n * 4);
  A(f->d_boxes_tmp, (size_t)n * 4);
  A(f->d_explicit, 4);
  PlannerDev& pl = f->pl;
  pl.nx = cm.nx;
  pl.ny = cm.ny;
  pl.cells = cm.cells;
  pl.cells_padded = cm.cells_padded;
  pl.res = cm.res;
  pl.inv_res = 1.0 / cm.res;
  pl.origin = cm.origin;
  pl.master = cm.master;
  pl.max_plan = f->desc.max_plan;
  pl.max_sim_steps = f->desc.max_sim_steps;
  A(pl.state, n);
  A(pl.plan, (size_t)n * pl.max_plan * 2);
  A(pl.plan_count, n);
  A(pl.front_last, (size_t)n * 2);
  A(pl.align_on, n);
  A(pl.fp_spec, (size_t)n * kMaxFootprint * 2);
  A(pl.fp_n, n);
  A(pl.axis_count, (size_t)n * 4);
  A(pl.path, (size_t)n * pl.cells);
  A(pl.goal, (size_t)n * pl.cells);
  A(pl.goal_front, (size_t)n * pl.cells);
  if (bfs_scratch_words(cm.nx, cm.ny)) A(pl.bfs_scratch, (size_t)n * bfs_scratch_words(cm.nx, cm.ny));
  pl.bfs_grids = 3;
  pl.within = nullptr;
  A(pl.counters, (size_t)n * 2);
  A(pl.osc_flags, n);
  A(pl.osc_prev, (size_t)n * 3);
  A(pl.traj, (size_t)n * pl.max_sim_steps * 3);
#undef A
  f->h_origin.assign((size_t)n * 2, 0.0);
  f->h_fp_spec.assign((size_t)n * kMaxFootprint * 2, 0.0);
  f->h_fp_n.assign(n, 0);
  // grids start at their default values (Costmap2D ctor -> resetMaps)
  launch_fill_u8(cm.master, cm.master_default, (size_t)n * cm.cells_padded, f->stream);
  if (cm.obst) launch_fill_u8(cm.obst, cm.obstacle_default, (size_t)n * cm.cells_padded, f->stream);
  if (cm.voxel) launch_fill_u32(cm.voxel, 0x0000FFFFu, (size_t)n * cm.cells_padded, f->stream);  // voxel_grid.cpp:54
  // InflationLayer ctor: last_* = -/+FLT_MAX (inflation_layer.cpp:63-66)
  std::vector<InstCostmapState> st(n);
  for (auto& s : st) {
    memset(&s, 0, sizeof(s));
    s.last_min_x = -FLT_MAX;
    s.last_min_y = -FLT_MAX;
    s.last_max_x = FLT_MAX;
    s.last_max_y = FLT_MAX;
  }
  hipError_t e = hipMemcpyAsync(cm.state, st.data(), sizeof(InstCostmapState) * n, hipMemcpyHostToDevice, f->stream);
  if (e == hipSuccess) e = waitStream(f->stream);
  if (e != hipSuccess || checkLaunch() != NAVGPU_OK) {
    if (e != hipSuccess) g_last_error = std::string("fleet init: ") + hipGetErrorString(e);
    navgpu_fleet_destroy(f);
    return NAVGPU_ERR_NO_DEVICE;  // kernels not loadable on this device (not gfx950) or device lost
  }
  *out = f;
  return NAVGPU_OK;
}

int navgpu_fleet_destroy(navgpu_fleet* f) {
  if (!f) return NAVGPU_ERR_INVALID;
  if (f->stream) waitStream(f->stream);
  for (auto& e : f->events) {
    hipEventDestroy(e.a);
    hipEventDestroy(e.b);
  }
  for (auto& e : f->free_events) {
    hipEventDestroy(e.a);
    hipEventDestroy(e.b);
  }
  for (void* p : f->allocs) hipFree(p);
  for (void* p : f->pinned) hipHostFree(p);
  if (f->stream) hipStreamDestroy(f->stream);
  delete f;
  return NAVGPU_OK;
}
int navgpu_sync(navgpu_fleet* f) {
  if (!f) return NAVGPU_ERR_INVALID;
  HIP_TRY(waitStream(f->stream));
  return NAVGPU_OK;
}
void* navgpu_stream(navgpu_fleet* f) { return f ? (void*)f->stream : nullptr; }

int navgpu_fleet_set_origin(navgpu_fleet* f, uint32_t first, uint32_t count, const double* xy) {
  if (!f || !xy || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  memcpy(&f->h_origin[(size_t)first * 2], xy, sizeof(double) * 2 * count);
  HIP_TRY(hipMemcpyAsync(f->cm.origin + (size_t)first * 2, xy, sizeof(double) * 2 * count, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(waitStream(f->stream));
  return NAVGPU_OK;
}

int navgpu_fleet_get_origin(navgpu_fleet* f, uint32_t first, uint32_t count, double* xy) {
  if (!f || !xy || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  memcpy(xy, &f->h_origin[(size_t)first * 2], sizeof(double) * 2 * count);
  return NAVGPU_OK;
}

static int gridInfo(navgpu_fleet* f, int grid, void** base, size_t* elem, size_t* stride_elems, size_t* used_elems) {
  CostmapDev& cm = f->cm;
  PlannerDev& pl = f->pl;
  switch (grid) {
    case NAVGPU_GRID_MASTER: *base = cm.master; *elem = 1; *stride_elems = cm.cells_padded; break;
    case NAVGPU_GRID_STATIC: *base = cm.stat; *elem = 1; *stride_elems = cm.cells_padded; break;
    case NAVGPU_GRID_OBSTACLE: *base = cm.obst; *elem = 1; *stride_elems = cm.cells_padded; break;
    case NAVGPU_GRID_VOXEL: *base = cm.voxel; *elem = 4; *stride_elems = cm.cells_padded; break;
    case NAVGPU_GRID_PATH: *base = pl.path; *elem = 4; *stride_elems = pl.cells; break;
    case NAVGPU_GRID_GOAL: *base = pl.goal; *elem = 4; *stride_elems = pl.cells; break;
    case NAVGPU_GRID_GOAL_FRONT: *base = pl.goal_front; *elem = 4; *stride_elems = pl.cells; break;
    default: return NAVGPU_ERR_INVALID;
  }
  *used_elems = cm.cells;
  if (!*base) return NAVGPU_ERR_STATE;  // layer not part of this fleet
  return NAVGPU_OK;
}
int navgpu_grid_upload(navgpu_fleet* f, int grid, uint32_t first, uint32_t count, const void* host) {
  if (!f || !host || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  void* base;
  size_t elem, stride, used;
  int rc = gridInfo(f, grid, &base, &elem, &stride, &used);
  if (rc) return rc;
  HIP_TRY(hipMemcpy2DAsync((char*)base + (size_t)first * stride * elem, stride * elem, host, used * elem, used * elem, count,
                           hipMemcpyHostToDevice, f->stream));
  HIP_TRY(waitStream(f->stream));
  return NAVGPU_OK;
}
int navgpu_grid_download(navgpu_fleet* f, int grid, uint32_t first, uint32_t count, void* host) {
  if (!f || !host || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  void* base;
  size_t elem, stride, used;
  int rc = gridInfo(f, grid, &base, &elem, &stride, &used);
  if (rc) return rc;
  HIP_TRY(hipMemcpy2DAsync(host, used * elem, (char*)base + (size_t)first * stride * elem, stride * elem, used * elem, count,
                           hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(waitStream(f->stream));
  return NAVGPU_OK;
}
int navgpu_grid_device(navgpu_fleet* f, int grid, void** ptr, size_t* stride_bytes) {
  if (!f || !ptr) return NAVGPU_ERR_INVALID;
  void* base;
  size_t elem, stride, used;
  int rc = gridInfo(f, grid, &base, &elem, &stride, &used);
  if (rc) return rc;
  *ptr = base;
  if (stride_bytes) *stride_bytes = stride * elem;
  return NAVGPU_OK;
}
int navgpu_grid_reset(navgpu_fleet* f, int grid, uint32_t first, uint32_t count) {
  if (!f || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  CostmapDev& cm = f->cm;
  switch (grid) {
    case NAVGPU_GRID_MASTER: launch_fill_u8(cm.master + (size_t)first * cm.cells_padded, cm.master_default, (size_t)count * cm.cells_padded, f->stream); break;
    case NAVGPU_GRID_OBSTACLE:
      if (!cm.obst) return NAVGPU_ERR_STATE;
      launch_fill_u8(cm.obst + (size_t)first * cm.cells_padded, cm.obstacle_default, (size_t)count * cm.cells_padded, f->stream);
      if (cm.voxel) launch_fill_u32(cm.voxel + (size_t)first * cm.cells_padded, 0x0000FFFFu, (size_t)count * cm.cells_padded, f->stream);  // VoxelLayer::resetMaps
      break;
    case NAVGPU_GRID_VOXEL:
      if (!cm.voxel) return NAVGPU_ERR_STATE;
      launch_fill_u32(cm.voxel + (size_t)first * cm.cells_padded, 0x0000FFFFu, (size_t)count * cm.cells_padded, f->stream);
      break;
    default: return NAVGPU_ERR_INVALID;
  }
  return checkLaunch();
}

// ------------------------------------------------------------------------------------------------ layers
int navgpu_static_set_map(navgpu_fleet* f, uint32_t first, uint32_t count, const int8_t* occ, int32_t track_unknown_space,
                          int32_t use_maximum, int32_t trinary, int32_t lethal_cost_threshold, int32_t unknown_cost_value) {
  if (!f || !occ || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  CostmapDev& cm = f->cm;
  if (!cm.stat) return NAVGPU_ERR_STATE;
  if (!f->d_occ) {
    int rc = f->alloc(&f->d_occ, cm.cells);
    if (rc) return rc;
  }
  HIP_TRY(hipMemcpyAsync(f->d_occ, occ, cm.cells, hipMemcpyHostToDevice, f->stream));
  int lethal = std::max(std::min(lethal_cost_threshold, 100), 0);  // static_layer.cpp:80
  launch_static_interpret(cm.stat + (size_t)first * cm.cells_padded, f->d_occ, cm.cells, cm.cells_padded, count, track_unknown_space,
                          trinary, lethal, unknown_cost_value, f->stream);
  cm.static_use_maximum = use_maximum;
  cm.static_received = 1;
  // has_updated_data_ = true for these instances
  std::vector<InstCostmapState> st(count);
  HIP_TRY(hipMemcpyAsync(st.data(), cm.state + first, sizeof(InstCostmapState) * count, hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(waitStream(f->stream));
  for (auto& s : st) s.static_has_updated_data = 1;
  HIP_TRY(hipMemcpyAsync(cm.state + first, st.data(), sizeof(InstCostmapState) * count, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(waitStream(f->stream));
  return checkLaunch();
}

int navgpu_obstacle_configure(navgpu_fleet* f, const navgpu_obstacle_params* p) {
  if (!f || !p) return NAVGPU_ERR_INVALID;
  if (p->z_voxels < 0 || p->z_voxels > 16) return NAVGPU_ERR_INVALID;
  CostmapDev& cm = f->cm;
  f->obsp = *p;
  cm.obs_enabled = p->enabled;
  cm.footprint_clearing = p->footprint_clearing_enabled;
  cm.combination_method = p->combination_method;
  cm.max_obstacle_height = p->max_obstacle_height;
  if (cm.layers & NAVGPU_LAYER_VOXEL) {
    cm.z_voxels = p->z_voxels;
    cm.origin_z = p->origin_z;
    cm.z_resolution = p->z_resolution;
    cm.unknown_threshold = p->unknown_threshold;
    cm.mark_threshold = p->mark_threshold;
  }
  return NAVGPU_OK;
}

int navgpu_inflation_configure(navgpu_fleet* f, const navgpu_inflation_params* p) {
  if (!f || !p) return NAVGPU_ERR_INVALID;
  CostmapDev& cm = f->cm;
  // cellDistance (costmap_2d.cpp:181-185)
  double cells_dist = std::max(0.0, ceil(p->inflation_radius / cm.res));
  if (cells_dist > 64) return NAVGPU_ERR_CAPACITY;
  const uint32_t R = (uint32_t)cells_dist;
  const uint32_t n = R + 2;
  // computeCaches (inflation_layer.cpp:295-328) + computeCost (inflation_layer.h:114-129), fp64 libm on the host
  std::vector<uint8_t> lut((size_t)n * n);
  for (uint32_t i = 0; i < n; ++i)
    for (uint32_t j = 0; j < n; ++j) {
      double distance = hypot(i, j);
      uint8_t cost = 0;
      if (distance == 0)
        cost = kLethal;
      else if (distance * cm.res <= p->inscribed_radius)
        cost = kInscribed;
      else {
        double euclidean_distance = distance * cm.res;
        double factor = exp(-1.0 * p->cost_scaling_factor * (euclidean_distance - p->inscribed_radius));
        cost = (uint8_t)((kInscribed - 1) * factor);
      }
      // enqueue() drops cells whose cached distance exceeds the cell radius (inflation_layer.cpp:286)
      if (distance > R) cost = 0;
      lut[(size_t)i * n + j] = cost;
    }
  // the exact-EDT kernel takes the max over candidate seeds; that equals "cost of the nearest
  // seed" only if the table is monotone in distance — true for cost_scaling_factor >= 0.
  if (p->cost_scaling_factor < 0) {
    g_last_error = "cost_scaling_factor < 0 is not supported";
    return NAVGPU_ERR_INVALID;
  }
  HIP_TRY(hipMemcpyAsync(cm.lut, lut.data(), lut.size(), hipMemcpyHostToDevice, f->stream));
  // cost by squared distance for the bit-parallel kernel: max over the (i, j) pairs that share d^2,
  // usable only if it is non-increasing in d^2 (then max-over-seeds == cost of the nearest seed)
  std::vector<uint8_t> lut2(256, 0);
  bool lut2_ok = R <= 14;
  if (lut2_ok) {
    for (uint32_t i = 0; i <= R; ++i)
      for (uint32_t j = 0; j <= R; ++j) {
        const uint32_t d2 = i * i + j * j;
        if (d2 < 256) lut2[d2] = std::max(lut2[d2], lut[(size_t)i * n + j]);
      }
    uint8_t prev = 255;
    for (uint32_t d2 = 0; d2 <= R * R; ++d2) {
      bool reachable = false;
      for (uint32_t i = 0; i * i <= d2 && !reachable; ++i) {
        const uint32_t rem = d2 - i * i, j = (uint32_t)llround(sqrt((double)rem));
        reachable = j * j == rem;
      }
      if (!reachable) continue;
      if (lut2[d2] > prev) lut2_ok = false;
      prev = lut2[d2];
    }
  }
  HIP_TRY(hipMemcpyAsync(cm.lut2, lut2.data(), lut2.size(), hipMemcpyHostToDevice, f->stream));
  cm.lut2_ok = lut2_ok ? 1 : 0;
  HIP_TRY(waitStream(f->stream));
  const bool changed = !f->inflation_configured || f->infl.inflation_radius != p->inflation_radius ||
                       f->infl.cost_scaling_factor != p->cost_scaling_factor || f->infl.inscribed_radius != p->inscribed_radius ||
                       f->infl.enabled != p->enabled;
  f->infl = *p;
  f->inflation_configured = true;
  cm.R = R;
  cm.inflation_radius = p->inflation_radius;
  cm.infl_enabled = p->enabled;
  if (changed) {  // need_reinflation_ = true (setInflationParameters / onFootprintChanged / reconfigureCB)
    const uint32_t nI = f->desc.n_instances;
    std::vector<InstCostmapState> st(nI);
    HIP_TRY(hipMemcpyAsync(st.data(), cm.state, sizeof(InstCostmapState) * nI, hipMemcpyDeviceToHost, f->stream));
    HIP_TRY(waitStream(f->stream));
    for (auto& s : st) s.need_reinflation = 1;
    HIP_TRY(hipMemcpyAsync(cm.state, st.data(), sizeof(InstCostmapState) * nI, hipMemcpyHostToDevice, f->stream));
    HIP_TRY(waitStream(f->stream));
  }
  return NAVGPU_OK;
}

static void updateWindow(navgpu_fleet* f) {
  // costmap window staged in LDS by k_score: everything a trajectory's footprint or shifted
  // point can touch.  Results never depend on it (cells outside fall back to global loads).
  const navgpu_dwa_config& c = f->pl.cfg;
  double vmax;
  if (c.max_trans_vel >= 0)
    vmax = c.max_trans_vel + 1e-4;
  else
    vmax = hypot(std::max(fabs(c.min_vel_x), fabs(c.max_vel_x)), std::max(fabs(c.min_vel_y), fabs(c.max_vel_y)));
  double reach = vmax * c.sim_time + std::max(f->fp_radius, fabs(c.forward_point_distance));
  double cells = ceil(reach / f->pl.res) + 2;
  uint32_t win = (uint32_t)std::min(cells * 2 + 1, 240.0);
  f->pl.win = win;
  f->pl.fp_rcells = (uint32_t)ceil(f->fp_radius / f->pl.res) + 1;  // vertex cells lie within this Chebyshev radius of the centre cell
  // shared heading tables (k_score<TABLES>): constant velocity + fixed step count only
  uint32_t max_nfp = 0;
  for (uint32_t v : f->h_fp_n) max_nfp = std::max(max_nfp, v);
  f->pl.use_tables = 0;
  if (c.use_dwa && c.discretize_by_time && max_nfp <= 8) {
    f->pl.tab_steps = (uint32_t)ceil(c.sim_time / c.sim_granularity);
    f->pl.tab_nfp = max_nfp;
    f->pl.tab_nth = (uint32_t)std::max(c.vth_samples, 2) + 1;
    const size_t lds = score_window_bytes(win) + score_table_bytes(f->pl);
    if (f->pl.tab_steps >= 1 && f->pl.tab_steps <= f->pl.max_sim_steps && lds <= 60 * 1024) f->pl.use_tables = 1;
  }
}

// (re)allocate the per-robot LDS images of k_score for the current window / table geometry
static int ensurePrep(navgpu_fleet* f) {
  PlannerDev tmp = f->pl;
  tmp.use_tables = 1;  // upper bound: the image with tables, whatever the launch decides
  const size_t need = (score_prep_bytes(tmp) + 255) & ~(size_t)255;
  if (f->pl.prep && need <= f->pl.prep_stride) return NAVGPU_OK;
  HIP_TRY(waitStream(f->stream));
  f->release(f->pl.prep);
  f->pl.prep = nullptr;
  int rc = f->alloc(&f->pl.prep, (size_t)f->desc.n_instances * need);
  if (rc != NAVGPU_OK) return rc;
  f->pl.prep_stride = (uint32_t)need;
  return NAVGPU_OK;
}

int navgpu_set_footprint(navgpu_fleet* f, uint32_t first, uint32_t count, const double* xy, uint32_t nv) {
  if (!f || !f->rangeOk(first, count) || (nv && !xy)) return NAVGPU_ERR_INVALID;
  if (nv > f->desc.max_footprint || nv > (uint32_t)kMaxFootprint) return NAVGPU_ERR_CAPACITY;
  for (uint32_t i = first; i < first + count; ++i) {
    f->h_fp_n[i] = nv;
    for (uint32_t k = 0; k < nv * 2; ++k) f->h_fp_spec[(size_t)i * kMaxFootprint * 2 + k] = xy[k];
  }
  for (uint32_t k = 0; k < nv; ++k) f->fp_radius = std::max(f->fp_radius, hypot(xy[2 * k], xy[2 * k + 1]));
  if (f->fp_radius / f->cm.res > 500) return NAVGPU_ERR_CAPACITY;  // polygon column span must fit the 1024-column LDS table
  HIP_TRY(hipMemcpyAsync(f->pl.fp_spec + (size_t)first * kMaxFootprint * 2, &f->h_fp_spec[(size_t)first * kMaxFootprint * 2],
                         sizeof(double) * kMaxFootprint * 2 * count, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(f->pl.fp_n + first, &f->h_fp_n[first], sizeof(uint32_t) * count, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(waitStream(f->stream));
  if (f->planner_configured) {
    updateWindow(f);
    int rc = ensurePrep(f);
    if (rc != NAVGPU_OK) return rc;
  }
  return NAVGPU_OK;
}

int navgpu_costmap_stage(navgpu_fleet* f, uint32_t first, uint32_t count, const double* poses, const navgpu_observation* obs,
                         uint32_t n_obs, const float* points, uint32_t n_points_total) {
  if (!f || !poses || !f->rangeOk(first, count) || (n_obs && (!obs || (!points && n_points_total)))) return NAVGPU_ERR_INVALID;
  CostmapDev& cm = f->cm;
  if (f->desc.rolling_window && f->shift_pending) return NAVGPU_ERR_STATE;  // previous stage not consumed by an update yet
  HIP_TRY(waitStream(f->stream));  // the pinned mirrors may still feed an earlier copy
  for (uint32_t li = 0; li < count; ++li) f->hp_cnt[first + li] = f->hp_used[first + li] = 0;
  for (uint32_t k = 0; k < n_obs; ++k) {
    const navgpu_observation& o = obs[k];
    if (o.instance < first || o.instance >= first + count) return NAVGPU_ERR_INVALID;
    const uint32_t i = o.instance;
    if (f->hp_cnt[i] >= cm.max_obs || f->hp_used[i] + o.n_points > cm.max_points) return NAVGPU_ERR_CAPACITY;
    if ((uint64_t)o.first_point + o.n_points > n_points_total) return NAVGPU_ERR_INVALID;
    ObsCsr& d = f->hp_obs[(size_t)i * cm.max_obs + f->hp_cnt[i]++];
    d.first_point = f->hp_used[i];
    d.n_points = o.n_points;
    d.flags = o.flags;
    d.pad = 0;
    d.ox = o.origin_x;
    d.oy = o.origin_y;
    d.oz = o.origin_z;
    d.obstacle_range = o.obstacle_range;
    d.raytrace_range = o.raytrace_range;
    if (o.n_points)
      memcpy(&f->hp_pts[((size_t)i * cm.max_points + f->hp_used[i]) * 3], points + (size_t)o.first_point * 3, sizeof(float) * 3 * o.n_points);
    f->hp_used[i] += o.n_points;
  }
  // rolling window: LayeredCostmap::updateMap :86-91 + Costmap2D::updateOrigin :264-276, evaluated here
  // in fp64 exactly as the reference does; the grids are shifted on the device by navgpu_costmap_update
  if (f->desc.rolling_window) {
    const double size_m_x = (cm.nx - 1 + 0.5) * cm.res, size_m_y = (cm.ny - 1 + 0.5) * cm.res;  // getSizeInMetersX/Y
    for (uint32_t li = 0; li < count; ++li) {
      double& ox = f->h_origin[(size_t)(first + li) * 2];
      double& oy = f->h_origin[(size_t)(first + li) * 2 + 1];
      const double new_origin_x = poses[3 * li] - size_m_x / 2, new_origin_y = poses[3 * li + 1] - size_m_y / 2;
      const int cell_ox = int((new_origin_x - ox) / cm.res), cell_oy = int((new_origin_y - oy) / cm.res);
      ox = ox + cell_ox * cm.res;
      oy = oy + cell_oy * cm.res;
      f->hp_shift[2 * (first + li)] = cell_ox;
      f->hp_shift[2 * (first + li) + 1] = cell_oy;
    }
    HIP_TRY(hipMemcpyAsync(cm.shift + (size_t)first * 2, f->hp_shift + (size_t)first * 2, sizeof(int32_t) * 2 * count, hipMemcpyHostToDevice, f->stream));
    HIP_TRY(hipMemcpyAsync(cm.origin + (size_t)first * 2, &f->h_origin[(size_t)first * 2], sizeof(double) * 2 * count, hipMemcpyHostToDevice, f->stream));
    f->shift_pending = true;
    f->shift_first = first;
    f->shift_count = count;
  }
  // transformFootprint (footprint.cpp:103-118) per instance, fp64 libm
  for (uint32_t li = 0; li < count; ++li) {
    const uint32_t i = first + li;
    const double x = poses[3 * li], y = poses[3 * li + 1], th = poses[3 * li + 2];
    f->hp_pose[3 * i] = x;
    f->hp_pose[3 * i + 1] = y;
    f->hp_pose[3 * i + 2] = th;
    const double cos_th = cos(th), sin_th = sin(th);
    for (uint32_t v = 0; v < f->h_fp_n[i]; ++v) {
      const double sx = f->h_fp_spec[((size_t)i * kMaxFootprint + v) * 2], sy = f->h_fp_spec[((size_t)i * kMaxFootprint + v) * 2 + 1];
      f->hp_fpw[((size_t)i * kMaxFootprint + v) * 2] = x + (sx * cos_th - sy * sin_th);
      f->hp_fpw[((size_t)i * kMaxFootprint + v) * 2 + 1] = y + (sx * sin_th + sy * cos_th);
    }
  }
  HIP_TRY(hipMemcpyAsync(cm.pose + (size_t)first * 3, f->hp_pose + (size_t)first * 3, sizeof(double) * 3 * count, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(cm.obs + (size_t)first * cm.max_obs, f->hp_obs + (size_t)first * cm.max_obs, sizeof(ObsCsr) * (size_t)count * cm.max_obs, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(cm.obs_count + first, f->hp_cnt + first, sizeof(uint32_t) * count, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(cm.points + (size_t)first * cm.max_points * 3, f->hp_pts + (size_t)first * cm.max_points * 3, sizeof(float) * 3 * (size_t)count * cm.max_points, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(cm.fp_world + (size_t)first * kMaxFootprint * 2, f->hp_fpw + (size_t)first * kMaxFootprint * 2, sizeof(double) * 2 * kMaxFootprint * (size_t)count, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(cm.fp_n + first, &f->h_fp_n[first], sizeof(uint32_t) * count, hipMemcpyHostToDevice, f->stream));
  return NAVGPU_OK;  // copies stay in flight on the fleet's stream; the kernels are ordered behind them
}

int navgpu_costmap_update(navgpu_fleet* f, uint32_t first, uint32_t count) {
  if (!f || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  CostmapDev& cm = f->cm;
  if ((cm.layers & NAVGPU_LAYER_INFLATION) && !f->inflation_configured) return NAVGPU_ERR_STATE;
  if (f->desc.rolling_window && f->shift_pending) {
    if (first != f->shift_first || count != f->shift_count) return NAVGPU_ERR_STATE;
    // the ping-pong swap is fleet-wide, so a rolling fleet is staged and updated as a whole
    if (first != 0 || count != f->desc.n_instances) return NAVGPU_ERR_INVALID;
    launch_shift_u8(cm.master, cm.master_alt, cm, first, count, cm.master_default, f->stream);
    std::swap(cm.master, cm.master_alt);
    f->pl.master = cm.master;
    if (cm.obst) {
      launch_shift_u8(cm.obst, cm.obst_alt, cm, first, count, cm.obstacle_default, f->stream);
      std::swap(cm.obst, cm.obst_alt);
    }
    if (cm.voxel) {
      launch_shift_u32(cm.voxel, cm.voxel_alt, cm, first, count, 0x0000FFFFu, f->stream);
      std::swap(cm.voxel, cm.voxel_alt);
    }
    f->shift_pending = false;
  }
  PROFILED(f, NAVGPU_K_OBSTACLE, launch_obstacle(cm, first, count, nullptr, 0, f->stream));
  PROFILED(f, NAVGPU_K_MERGE, launch_merge(cm, first, count, nullptr, f->stream));
  if (cm.layers & NAVGPU_LAYER_INFLATION) PROFILED(f, NAVGPU_K_INFLATE, launch_inflate(cm, first, count, nullptr, f->stream));
  return checkLaunch();
}

int navgpu_costmap_bounds(navgpu_fleet* f, uint32_t first, uint32_t count, int32_t* boxes) {
  if (!f || !boxes || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  std::vector<InstCostmapState> st(count);
  HIP_TRY(hipMemcpyAsync(st.data(), f->cm.state + first, sizeof(InstCostmapState) * count, hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(waitStream(f->stream));
  for (uint32_t i = 0; i < count; ++i)
    for (int k = 0; k < 4; ++k) boxes[4 * i + k] = st[i].box[k];
  return NAVGPU_OK;
}

int navgpu_inflate(navgpu_fleet* f, uint32_t first, uint32_t count, const int32_t* boxes) {
  if (!f || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  if (!f->inflation_configured) return NAVGPU_ERR_STATE;
  const int32_t* d_boxes = nullptr;
  if (boxes) {
    HIP_TRY(hipMemcpyAsync(f->d_boxes_tmp, boxes, sizeof(int32_t) * 4 * count, hipMemcpyHostToDevice, f->stream));
    d_boxes = f->d_boxes_tmp;
  }
  PROFILED(f, NAVGPU_K_INFLATE, launch_inflate(f->cm, first, count, d_boxes, f->stream));
  if (boxes) HIP_TRY(waitStream(f->stream));  // d_boxes_tmp is reused by the next call
  return checkLaunch();
}

int navgpu_obstacle_update_bounds(navgpu_fleet* f, uint32_t first, uint32_t count, double* bounds) {
  if (!f || !bounds || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  if (!f->cm.obst) return NAVGPU_ERR_STATE;
  HIP_TRY(hipMemcpyAsync(f->d_bounds_tmp, bounds, sizeof(double) * 4 * count, hipMemcpyHostToDevice, f->stream));
  PROFILED(f, NAVGPU_K_OBSTACLE, launch_obstacle(f->cm, first, count, f->d_bounds_tmp, 1, f->stream));
  HIP_TRY(hipMemcpyAsync(bounds, f->d_bounds_tmp, sizeof(double) * 4 * count, hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(waitStream(f->stream));
  return checkLaunch();
}

int navgpu_obstacle_update_costs(navgpu_fleet* f, uint32_t first, uint32_t count, const int32_t* boxes) {
  if (!f || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  const int32_t* d_boxes = nullptr;
  if (boxes) {
    HIP_TRY(hipMemcpyAsync(f->d_boxes_tmp, boxes, sizeof(int32_t) * 4 * count, hipMemcpyHostToDevice, f->stream));
    d_boxes = f->d_boxes_tmp;
  }
  PROFILED(f, NAVGPU_K_MERGE, launch_merge(f->cm, first, count, d_boxes, f->stream));
  if (boxes) HIP_TRY(waitStream(f->stream));
  return checkLaunch();
}

// ------------------------------------------------------------------------------------------------ planner
int navgpu_planner_configure(navgpu_fleet* f, const navgpu_dwa_config* c) {
  if (!f || !c) return NAVGPU_ERR_INVALID;
  if (!(c->sim_time > 0) || !(c->sim_granularity > 0) || !(c->angular_sim_granularity > 0)) return NAVGPU_ERR_INVALID;
  PlannerDev& pl = f->pl;
  navgpu_dwa_config cfg = *c;
  if (cfg.vx_samples <= 0) cfg.vx_samples = 1;  // dwa_planner.cpp:89-105
  if (cfg.vy_samples <= 0) cfg.vy_samples = 1;
  if (cfg.vth_samples <= 0) cfg.vth_samples = 1;
  const uint32_t max_axis = (uint32_t)std::max(std::max(cfg.vx_samples, cfg.vy_samples), std::max(cfg.vth_samples, 2)) + 1;
  if (max_axis > 128) return NAVGPU_ERR_CAPACITY;
  // step-count capacity (simple_trajectory_generator.cpp:202-212)
  double steps;
  if (cfg.discretize_by_time)
    steps = ceil(cfg.sim_time / cfg.sim_granularity);
  else {
    double vmax = cfg.max_trans_vel >= 0 ? cfg.max_trans_vel + 1e-4
                                         : hypot(std::max(fabs(cfg.min_vel_x), fabs(cfg.max_vel_x)), std::max(fabs(cfg.min_vel_y), fabs(cfg.max_vel_y)));
    steps = ceil(std::max(vmax * cfg.sim_time / cfg.sim_granularity, fabs(cfg.max_rot_vel) * cfg.sim_time / cfg.angular_sim_granularity)) + 1;
  }
  if (steps > pl.max_sim_steps) {
    g_last_error = "max_sim_steps too small for this sim_time / granularity";
    return NAVGPU_ERR_CAPACITY;
  }
  const uint32_t nI = f->desc.n_instances;
  const uint32_t ax = (uint32_t)(std::max(cfg.vx_samples, 2) + 1), ay = (uint32_t)(std::max(cfg.vy_samples, 2) + 1),
                 at = (uint32_t)(std::max(cfg.vth_samples, 2) + 1);
  const uint32_t max_samples = ax * ay * at;
  const uint32_t score_blocks = (max_samples + kScoreThreads - 1) / kScoreThreads;
  if (max_axis != pl.max_axis || max_samples != pl.max_samples) {
    HIP_TRY(waitStream(f->stream));
    f->release(pl.axis_samples);
    f->release(pl.part_cost);
    f->release(pl.part_index);
    f->release(pl.sample_cost);
    f->release(pl.sample_status);
    pl.axis_samples = nullptr;
    pl.part_cost = nullptr;
    pl.part_index = nullptr;
    pl.sample_cost = nullptr;
    pl.sample_status = nullptr;
    int rc;
    if ((rc = f->alloc(&pl.axis_samples, (size_t)nI * 3 * max_axis))) return rc;
    if ((rc = f->alloc(&pl.part_cost, (size_t)nI * score_blocks))) return rc;
    if ((rc = f->alloc(&pl.part_index, (size_t)nI * score_blocks))) return rc;
    if (f->desc.keep_sample_costs) {
      if ((rc = f->alloc(&pl.sample_cost, (size_t)nI * max_samples))) return rc;
      if ((rc = f->alloc(&pl.sample_status, (size_t)nI * max_samples))) return rc;
    }
    pl.max_axis = max_axis;
    pl.max_samples = max_samples;
    pl.score_blocks = score_blocks;
  }
  pl.cfg = cfg;
  // DWAPlanner::reconfigure scales (dwa_planner.cpp:64-75)
  pl.scale_path = pl.res * cfg.path_distance_bias * 0.5;
  pl.scale_goal = pl.res * cfg.goal_distance_bias * 0.5;
  pl.scale_obstacle = pl.res * cfg.occdist_scale;
  f->planner_configured = true;
  updateWindow(f);
  {
    int rc = ensurePrep(f);
    if (rc != NAVGPU_OK) return rc;
  }
  HIP_TRY(waitStream(f->stream));
  return NAVGPU_OK;
}

int navgpu_planner_set_plan(navgpu_fleet* f, uint32_t first, uint32_t count) {
  if (!f || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  HIP_TRY(hipMemsetAsync(f->pl.osc_flags + first, 0, sizeof(uint32_t) * count, f->stream));  // resetOscillationFlags
  return NAVGPU_OK;
}

int navgpu_planner_stage(navgpu_fleet* f, uint32_t first, uint32_t count, const navgpu_robot_state* states, const double* plan_xy,
                         uint32_t n_plan_total) {
  if (!f || !states || !plan_xy || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  if (!f->planner_configured) return NAVGPU_ERR_STATE;
  PlannerDev& pl = f->pl;
  const navgpu_dwa_config& c = pl.cfg;
  for (uint32_t li = 0; li < count; ++li) {
    const navgpu_robot_state& s = states[li];
    if (s.plan_count == 0) return NAVGPU_ERR_INVALID;  // the ROS wrapper rejects empty plans before this point
    if (s.plan_count > pl.max_plan) return NAVGPU_ERR_CAPACITY;
    if ((uint64_t)s.plan_first + s.plan_count > n_plan_total) return NAVGPU_ERR_INVALID;
  }
  HIP_TRY(waitStream(f->stream));  // the pinned mirrors may still feed an earlier copy
  for (uint32_t li = 0; li < count; ++li) {
    const uint32_t i = first + li;
    const navgpu_robot_state& s = states[li];
    f->hp_state[i] = s;
    memcpy(&f->hp_plan[(size_t)i * pl.max_plan * 2], plan_xy + (size_t)s.plan_first * 2, sizeof(double) * 2 * s.plan_count);
    f->hp_plan_cnt[i] = s.plan_count;
    // DWAPlanner::updatePlanAndLocalCosts (dwa_planner.cpp:254-285); pos is the float-narrowed pose
    const double gx = plan_xy[((size_t)s.plan_first + s.plan_count - 1) * 2], gy = plan_xy[((size_t)s.plan_first + s.plan_count - 1) * 2 + 1];
    const double sq_dist = (s.pos[0] - gx) * (s.pos[0] - gx) + (s.pos[1] - gy) * (s.pos[1] - gy);
    const double angle_to_goal = atan2(gy - s.pos[1], gx - s.pos[0]);
    f->hp_front[2 * i] = gx + c.forward_point_distance * cos(angle_to_goal);
    f->hp_front[2 * i + 1] = gy + c.forward_point_distance * sin(angle_to_goal);
    f->hp_align[i] = sq_dist > c.forward_point_distance * c.forward_point_distance * c.cheat_factor ? 1 : 0;
  }
  HIP_TRY(hipMemcpyAsync(pl.state + first, f->hp_state + first, sizeof(navgpu_robot_state) * count, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(pl.plan + (size_t)first * pl.max_plan * 2, f->hp_plan + (size_t)first * pl.max_plan * 2, sizeof(double) * 2 * (size_t)count * pl.max_plan, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(pl.plan_count + first, f->hp_plan_cnt + first, sizeof(uint32_t) * count, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(pl.front_last + (size_t)first * 2, f->hp_front + (size_t)first * 2, sizeof(double) * 2 * count, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(pl.align_on + first, f->hp_align + first, sizeof(int32_t) * count, hipMemcpyHostToDevice, f->stream));
  f->planner_staged = true;
  return NAVGPU_OK;
}

int navgpu_planner_cycle(navgpu_fleet* f, uint32_t first, uint32_t count) {
  if (!f || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  if (!f->planner_configured || !f->planner_staged) return NAVGPU_ERR_STATE;
  PlannerDev& pl = f->pl;
  launch_samples(pl, first, count, f->stream);
  PROFILED(f, NAVGPU_K_BFS, launch_bfs(pl, first, count, f->stream));
  uint32_t n_blocks = 0;
  PROFILED(f, NAVGPU_K_SCORE, n_blocks = launch_score(pl, first, count, nullptr, f->stream));
  PROFILED(f, NAVGPU_K_SELECT, launch_select(pl, first, count, n_blocks, f->stream));
  return checkLaunch();
}

int navgpu_planner_results(navgpu_fleet* f, uint32_t first, uint32_t count, navgpu_plan_result* results) {
  if (!f || !results || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  // zero-copy: the results already sit in pinned host memory once the stream has drained (an explicit
  // D2H copy was measured at ~4 ms per call when another HIP user, e.g. PyTorch, shares the process)
  static const bool dbg = getenv("NAVGPU_DEBUG_TIMING") != nullptr;
  timespec t0, t1, t2;
  if (dbg) clock_gettime(CLOCK_MONOTONIC, &t0);
  HIP_TRY(waitStream(f->stream));
  if (dbg) clock_gettime(CLOCK_MONOTONIC, &t1);
  memcpy(results, f->hp_result + first, sizeof(navgpu_plan_result) * count);
  if (dbg) {
    clock_gettime(CLOCK_MONOTONIC, &t2);
    fprintf(stderr, "results: wait %.3f ms, memcpy %.3f ms\n", (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6,
            (t2.tv_sec - t1.tv_sec) * 1e3 + (t2.tv_nsec - t1.tv_nsec) * 1e-6);
  }
  return NAVGPU_OK;
}

int navgpu_planner_trajectory(navgpu_fleet* f, uint32_t instance, double* xyth, uint32_t cap) {
  if (!f || !xyth || instance >= f->desc.n_instances) return NAVGPU_ERR_INVALID;
  HIP_TRY(waitStream(f->stream));
  const navgpu_plan_result r = f->hp_result[instance];
  uint32_t n = std::min<uint32_t>(r.n_points > 0 ? r.n_points : 0, cap);
  if (n) {
    HIP_TRY(hipMemcpyAsync(xyth, f->pl.traj + (size_t)instance * f->pl.max_sim_steps * 3, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, f->stream));
    HIP_TRY(waitStream(f->stream));
  }
  return r.n_points;
}

int navgpu_planner_samples(navgpu_fleet* f, uint32_t instance, double* costs, int32_t* status, float* vel, uint32_t cap) {
  if (!f || instance >= f->desc.n_instances) return NAVGPU_ERR_INVALID;
  PlannerDev& pl = f->pl;
  if (!pl.sample_cost) return NAVGPU_ERR_STATE;
  int32_t cnt[4];
  HIP_TRY(hipMemcpyAsync(cnt, pl.axis_count + 4 * instance, sizeof(cnt), hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(waitStream(f->stream));
  const uint32_t n = std::min<uint32_t>(cnt[3], cap);
  if (costs && n) HIP_TRY(hipMemcpyAsync(costs, pl.sample_cost + (size_t)instance * pl.max_samples, sizeof(double) * n, hipMemcpyDeviceToHost, f->stream));
  if (status && n) HIP_TRY(hipMemcpyAsync(status, pl.sample_status + (size_t)instance * pl.max_samples, sizeof(int32_t) * n, hipMemcpyDeviceToHost, f->stream));
  std::vector<float> ax((size_t)3 * pl.max_axis);
  if (vel && n) HIP_TRY(hipMemcpyAsync(ax.data(), pl.axis_samples + (size_t)instance * 3 * pl.max_axis, sizeof(float) * ax.size(), hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(waitStream(f->stream));
  if (vel)
    for (uint32_t s = 0; s < n; ++s) {
      const int ix = s / (cnt[1] * cnt[2]), rem = s - ix * (cnt[1] * cnt[2]), iy = rem / cnt[2], it = rem - iy * cnt[2];
      vel[3 * s] = ax[ix];
      vel[3 * s + 1] = ax[pl.max_axis + iy];
      vel[3 * s + 2] = ax[2 * pl.max_axis + it];
    }
  return cnt[3];
}

int navgpu_planner_check_trajectory(navgpu_fleet* f, uint32_t instance, const float vs[3], int32_t* ok) {
  if (!f || !vs || !ok || instance >= f->desc.n_instances) return NAVGPU_ERR_INVALID;
  if (!f->planner_configured || !f->planner_staged) return NAVGPU_ERR_STATE;
  PlannerDev& pl = f->pl;
  // checkTrajectory resets the oscillation flags first (dwa_planner.cpp:217)
  HIP_TRY(hipMemsetAsync(pl.osc_flags + instance, 0, sizeof(uint32_t), f->stream));
  HIP_TRY(hipMemcpyAsync(f->d_explicit, vs, sizeof(float) * 3, hipMemcpyHostToDevice, f->stream));
  launch_score(pl, instance, 1, f->d_explicit, f->stream);
  double cost;
  int idx;
  HIP_TRY(hipMemcpyAsync(&cost, pl.part_cost + (size_t)instance * pl.score_blocks, sizeof(double), hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(hipMemcpyAsync(&idx, pl.part_index + (size_t)instance * pl.score_blocks, sizeof(int), hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(waitStream(f->stream));
  *ok = (idx != 0x7FFFFFFF) ? 1 : 0;
  return checkLaunch();
}

int navgpu_planner_get_oscillation(navgpu_fleet* f, uint32_t first, uint32_t count, uint32_t* flags, float* prev) {
  if (!f || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  if (flags) HIP_TRY(hipMemcpyAsync(flags, f->pl.osc_flags + first, sizeof(uint32_t) * count, hipMemcpyDeviceToHost, f->stream));
  if (prev) HIP_TRY(hipMemcpyAsync(prev, f->pl.osc_prev + (size_t)first * 3, sizeof(float) * 3 * count, hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(waitStream(f->stream));
  return NAVGPU_OK;
}
int navgpu_planner_set_oscillation(navgpu_fleet* f, uint32_t first, uint32_t count, const uint32_t* flags, const float* prev) {
  if (!f || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  if (flags) HIP_TRY(hipMemcpyAsync(f->pl.osc_flags + first, flags, sizeof(uint32_t) * count, hipMemcpyHostToDevice, f->stream));
  if (prev) HIP_TRY(hipMemcpyAsync(f->pl.osc_prev + (size_t)first * 3, prev, sizeof(float) * 3 * count, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(waitStream(f->stream));
  return NAVGPU_OK;
}

// ------------------------------------------------------------------------------------------------ DWAPlannerROS mirror
// angles::normalize_angle_positive / normalize_angle / shortest_angular_distance (ros/angles, fmod form)
static double normalizeAnglePositive(double a) { return fmod(fmod(a, 2.0 * M_PI) + 2.0 * M_PI, 2.0 * M_PI); }
static double normalizeAngle(double a) {
  double r = normalizeAnglePositive(a);
  if (r > M_PI) r -= 2.0 * M_PI;
  return r;
}
double navgpu_shortest_angular_distance(double from, double to) { return normalizeAngle(to - from); }
static double signOf(double x) { return x < 0.0 ? -1.0 : 1.0; }  // base_local_planner sign()

int navgpu_local_plan_window(const double* plan, uint32_t n, const double pose[3], const double* T, double dist_threshold,
                             int32_t prune, double* out, uint32_t capacity, uint32_t* n_out, uint32_t* n_erased) {
  if (!plan || !pose || !out || !n_out || !n_erased) return NAVGPU_ERR_INVALID;
  *n_out = 0;
  *n_erased = 0;
  if (n == 0) return NAVGPU_ERR_INVALID;  // "Received plan with zero length" (goal_functions.cpp:98-101)
  // the robot in the frame of the plan (tf.transformPose, :113-114)
  double rx = pose[0], ry = pose[1];
  double c = 1.0, sn = 0.0;
  if (T) {
    c = cos(T[2]);
    sn = sin(T[2]);
    const double dx = pose[0] - T[0], dy = pose[1] - T[1];
    rx = c * dx + sn * dy;
    ry = -sn * dx + c * dy;
  }
  const double sq_thr = dist_threshold * dist_threshold;
  uint32_t i = 0;
  double sq_dist = 0;
  while (i < n) {  // :126-134: up to the first pose within reach
    const double xd = rx - plan[3 * i], yd = ry - plan[3 * i + 1];
    sq_dist = xd * xd + yd * yd;
    if (sq_dist <= sq_thr) break;
    ++i;
  }
  uint32_t m = 0;
  while (i < n && sq_dist <= sq_thr) {  // :140-154: the pose that leaves the reach is still taken
    if (m >= capacity) return NAVGPU_ERR_CAPACITY;
    const double px = plan[3 * i], py = plan[3 * i + 1], pth = plan[3 * i + 2];
    if (T) {
      out[3 * m] = c * px - sn * py + T[0];
      out[3 * m + 1] = sn * px + c * py + T[1];
      out[3 * m + 2] = pth + T[2];
    } else {
      out[3 * m] = px;
      out[3 * m + 1] = py;
      out[3 * m + 2] = pth;
    }
    ++m;
    const double xd = rx - px, yd = ry - py;
    sq_dist = xd * xd + yd * yd;
    ++i;
  }
  uint32_t erased = 0;
  if (prune) {  // prunePlan (:69-86): drop leading poses until one is closer than 1 m
    while (erased < m) {
      const double xd = pose[0] - out[3 * erased], yd = pose[1] - out[3 * erased + 1];
      if (xd * xd + yd * yd < 1) break;
      ++erased;
    }
    if (erased) memmove(out, out + 3 * (size_t)erased, sizeof(double) * 3 * (m - erased));
  }
  *n_out = m - erased;
  *n_erased = erased;
  return NAVGPU_OK;
}

int navgpu_local_planner_configure(navgpu_fleet* f, const navgpu_local_limits* lim) {
  if (!f || !lim) return NAVGPU_ERR_INVALID;
  f->lp_limits = *lim;
  f->lp_configured = true;
  if (f->lp.size() != f->desc.n_instances) f->lp.assign(f->desc.n_instances, navgpu_fleet::LocalPlannerState());
  return NAVGPU_OK;
}

int navgpu_local_planner_set_plan(navgpu_fleet* f, uint32_t instance, const double* plan, uint32_t n, const double* T) {
  if (!f || instance >= f->desc.n_instances || (n && !plan)) return NAVGPU_ERR_INVALID;
  if (!f->lp_configured || !f->planner_configured) return NAVGPU_ERR_STATE;
  navgpu_fleet::LocalPlannerState& st = f->lp[instance];
  st.xy_tolerance_latch = false;  // latchedStopRotateController_.resetLatching() (dwa_planner_ros.cpp:136)
  st.plan.assign(plan, plan + 3 * (size_t)n);
  st.have_plan = true;
  st.has_T = T != nullptr;
  if (T) memcpy(st.T, T, sizeof(st.T));
  return navgpu_planner_set_plan(f, instance, 1);  // DWAPlanner::setPlan: resetOscillationFlags
}

int navgpu_local_planner_get_plan(navgpu_fleet* f, uint32_t instance, double* xyyaw, uint32_t capacity) {
  if (!f || instance >= f->desc.n_instances || !f->lp_configured) return NAVGPU_ERR_INVALID;
  const std::vector<double>& pl = f->lp[instance].plan;
  const uint32_t n = (uint32_t)(pl.size() / 3);
  if (xyyaw) {
    if (capacity < n) return NAVGPU_ERR_CAPACITY;
    memcpy(xyyaw, pl.data(), sizeof(double) * pl.size());
  }
  return (int)n;
}

// getGoalPose (goal_functions.cpp:175-214): the last pose of the stored plan in the global frame
static bool goalPose(const navgpu_fleet::LocalPlannerState& st, double goal[3]) {
  if (st.plan.empty()) return false;
  const double* g = &st.plan[st.plan.size() - 3];
  if (st.has_T) {
    const double c = cos(st.T[2]), sn = sin(st.T[2]);
    goal[0] = c * g[0] - sn * g[1] + st.T[0];
    goal[1] = sn * g[0] + c * g[1] + st.T[1];
    goal[2] = g[2] + st.T[2];
  } else {
    goal[0] = g[0];
    goal[1] = g[1];
    goal[2] = g[2];
  }
  return true;
}
static bool stoppedOdom(const double v[3], double rot_stopped, double trans_stopped) {  // goal_functions.cpp:248-253
  return fabs(v[2]) <= rot_stopped && fabs(v[0]) <= trans_stopped && fabs(v[1]) <= trans_stopped;
}

int navgpu_local_planner_is_goal_reached(navgpu_fleet* f, uint32_t first, uint32_t count, const navgpu_robot_input* in, int32_t* reached) {
  if (!f || !in || !reached || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  if (!f->lp_configured) return NAVGPU_ERR_STATE;
  const navgpu_local_limits& lim = f->lp_limits;
  for (uint32_t k = 0; k < count; ++k) {
    navgpu_fleet::LocalPlannerState& st = f->lp[first + k];
    reached[k] = 0;
    double goal[3];
    if (!in[k].have_pose || !goalPose(st, goal)) continue;
    // LatchedStopRotateController::isGoalReached (:66-109)
    const double dist = hypot(goal[0] - in[k].pose[0], goal[1] - in[k].pose[1]);
    if ((lim.latch_xy_goal_tolerance && st.xy_tolerance_latch) || dist <= lim.xy_goal_tolerance) {
      if (lim.latch_xy_goal_tolerance && !st.xy_tolerance_latch) st.xy_tolerance_latch = true;
      const double angle = navgpu_shortest_angular_distance(in[k].pose[2], goal[2]);
      if (fabs(angle) <= lim.yaw_goal_tolerance && stoppedOdom(in[k].odom_vel, lim.rot_stopped_vel, lim.trans_stopped_vel)) reached[k] = 1;
    }
  }
  return NAVGPU_OK;
}

int navgpu_local_planner_compute_velocity_commands(navgpu_fleet* f, uint32_t first, uint32_t count, const navgpu_robot_input* in,
                                                   navgpu_cmd_result* out) {
  if (!f || !in || !out || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  if (!f->lp_configured || !f->planner_configured) return NAVGPU_ERR_STATE;
  const navgpu_local_limits& lim = f->lp_limits;
  const uint32_t max_plan = f->pl.max_plan;
  const double dist_threshold = std::max(f->cm.nx * f->cm.res / 2.0, f->cm.ny * f->cm.res / 2.0);
  std::vector<double> local((size_t)count * max_plan * 3), packed;
  std::vector<navgpu_robot_state> states(count);
  std::vector<uint8_t> valid(count, 0), dwa(count, 0);
  packed.reserve((size_t)count * max_plan * 2);
  // --- getLocalPlan per robot (computeVelocityCommands :254-271)
  for (uint32_t k = 0; k < count; ++k) {
    navgpu_fleet::LocalPlannerState& st = f->lp[first + k];
    navgpu_cmd_result& o = out[k];
    o = navgpu_cmd_result();
    if (!in[k].have_pose) continue;                    // "Could not get robot pose"
    if (!st.have_plan || st.plan.empty()) continue;    // transformGlobalPlan: "Received plan with zero length"
    uint32_t n_loc = 0, n_er = 0;
    int rc = navgpu_local_plan_window(st.plan.data(), (uint32_t)(st.plan.size() / 3), in[k].pose, st.has_T ? st.T : nullptr,
                                      dist_threshold, lim.prune_plan, &local[(size_t)k * max_plan * 3], max_plan, &n_loc, &n_er);
    if (rc == NAVGPU_ERR_CAPACITY) return rc;
    if (rc != NAVGPU_OK) continue;
    if (n_er) st.plan.erase(st.plan.begin(), st.plan.begin() + 3 * (size_t)std::min<size_t>(n_er, st.plan.size() / 3));
    o.local_plan_points = (int32_t)n_loc;
    if (n_loc == 0) continue;                          // "Received an empty transformed plan."
    valid[k] = 1;
    navgpu_robot_state& rs = states[k];
    for (int a = 0; a < 3; ++a) {
      rs.pos[a] = (float)in[k].pose[a];                // Eigen::Vector3f pos / vel (dwa_planner.cpp:303-304)
      rs.vel[a] = (float)in[k].odom_vel[a];
    }
    rs.plan_first = (uint32_t)(packed.size() / 2);
    rs.plan_count = n_loc;
    for (uint32_t q = 0; q < n_loc; ++q) {
      packed.push_back(local[((size_t)k * max_plan + q) * 3]);
      packed.push_back(local[((size_t)k * max_plan + q) * 3 + 1]);
    }
  }
  // --- updatePlanAndLocalCosts for every robot that has a local plan (:274), in contiguous runs
  for (uint32_t k = 0; k < count;) {
    if (!valid[k]) {
      ++k;
      continue;
    }
    uint32_t e = k;
    while (e < count && valid[e]) ++e;
    int rc = navgpu_planner_stage(f, first + k, e - k, &states[k], packed.data(), (uint32_t)(packed.size() / 2));
    if (rc != NAVGPU_OK) return rc;
    k = e;
  }
  // --- dispatch: isPositionReached (latched_stop_rotate_controller.cpp:37-58)
  for (uint32_t k = 0; k < count; ++k) {
    if (!valid[k]) continue;
    navgpu_fleet::LocalPlannerState& st = f->lp[first + k];
    double goal[3];
    bool reached = false;
    if (goalPose(st, goal)) {
      const double dist = hypot(goal[0] - in[k].pose[0], goal[1] - in[k].pose[1]);
      if ((lim.latch_xy_goal_tolerance && st.xy_tolerance_latch) || dist <= lim.xy_goal_tolerance) {
        st.xy_tolerance_latch = true;
        reached = true;
      }
    }
    if (!reached) {
      dwa[k] = 1;
      out[k].branch = NAVGPU_BRANCH_DWA;
      continue;
    }
    // computeVelocityCommandsStopRotate (:211-273)
    navgpu_cmd_result& o = out[k];
    if (!goalPose(st, goal)) continue;  // "Could not get goal pose"
    if (lim.latch_xy_goal_tolerance && !st.xy_tolerance_latch) st.xy_tolerance_latch = true;
    const double yaw = in[k].pose[2], vel_yaw = in[k].odom_vel[2];
    const double angle = navgpu_shortest_angular_distance(yaw, goal[2]);
    if (fabs(angle) <= lim.yaw_goal_tolerance) {
      o.cmd_vel[0] = o.cmd_vel[1] = o.cmd_vel[2] = 0.0;
      st.rotating_to_goal = false;
      o.ok = 1;
      o.branch = NAVGPU_BRANCH_AT_GOAL;
      continue;
    }
    const double acc[3] = {lim.acc_lim_x, lim.acc_lim_y, lim.acc_lim_theta};
    float vs[3];
    int32_t okc = 0;
    if (!st.rotating_to_goal && !stoppedOdom(in[k].odom_vel, lim.rot_stopped_vel, lim.trans_stopped_vel)) {
      // stopWithAccLimits (:111-146); Eigen::Vector3f narrows the samples to float
      const double vx = signOf(in[k].odom_vel[0]) * std::max(0.0, fabs(in[k].odom_vel[0]) - acc[0] * lim.sim_period);
      const double vy = signOf(in[k].odom_vel[1]) * std::max(0.0, fabs(in[k].odom_vel[1]) - acc[1] * lim.sim_period);
      const double vth = signOf(vel_yaw) * std::max(0.0, fabs(vel_yaw) - acc[2] * lim.sim_period);
      vs[0] = (float)vx;
      vs[1] = (float)vy;
      vs[2] = (float)vth;
      int rc = navgpu_planner_check_trajectory(f, first + k, vs, &okc);
      if (rc != NAVGPU_OK) return rc;
      o.branch = NAVGPU_BRANCH_STOP;
      if (okc) {
        o.cmd_vel[0] = vx;
        o.cmd_vel[1] = vy;
        o.cmd_vel[2] = vth;
        o.ok = 1;
      }  // else: zeros, "Error when stopping." -> false
    } else {
      // rotateToGoal (:148-209)
      st.rotating_to_goal = true;
      const double ang_diff = angle;
      double v = std::min(lim.max_rot_vel, std::max(lim.min_rot_vel, fabs(ang_diff)));
      const double max_acc_vel = fabs(vel_yaw) + acc[2] * lim.sim_period;
      const double min_acc_vel = fabs(vel_yaw) - acc[2] * lim.sim_period;
      v = std::min(std::max(fabs(v), min_acc_vel), max_acc_vel);
      const double max_speed_to_stop = sqrt(2 * acc[2] * fabs(ang_diff));
      v = std::min(max_speed_to_stop, fabs(v));
      v = std::min(lim.max_rot_vel, std::max(lim.min_rot_vel, v));
      if (ang_diff < 0) v = -v;
      vs[0] = 0.f;
      vs[1] = 0.f;
      vs[2] = (float)v;
      int rc = navgpu_planner_check_trajectory(f, first + k, vs, &okc);
      if (rc != NAVGPU_OK) return rc;
      o.branch = NAVGPU_BRANCH_ROTATE;
      if (okc) {
        o.cmd_vel[2] = v;
        o.ok = 1;
      }  // else: "Rotation cmd in collision" -> zeros, false
    }
  }
  // --- dwaComputeVelocityCommands (:176-247) for the others, in contiguous runs
  std::vector<navgpu_plan_result> res(count);
  for (uint32_t k = 0; k < count;) {
    if (!dwa[k]) {
      ++k;
      continue;
    }
    uint32_t e = k;
    while (e < count && dwa[e]) ++e;
    int rc = navgpu_planner_cycle(f, first + k, e - k);
    if (rc != NAVGPU_OK) return rc;
    rc = navgpu_planner_results(f, first + k, e - k, &res[k]);
    if (rc != NAVGPU_OK) return rc;
    for (uint32_t q = k; q < e; ++q) {
      navgpu_cmd_result& o = out[q];
      o.cmd_vel[0] = res[q].drive[0];
      o.cmd_vel[1] = res[q].drive[1];
      o.cmd_vel[2] = res[q].drive[2];
      o.ok = res[q].cost >= 0 ? 1 : 0;  // path.cost_ < 0: "failed to find a valid plan"
      o.trajectory_points = o.ok ? res[q].n_points : 0;
    }
    k = e;
  }
  return NAVGPU_OK;
}

// ------------------------------------------------------------------------------------------------ legacy TrajectoryPlanner
namespace {
struct TpCell {
  int x, y;
};
// Costmap2D::worldToMap on the host (costmap_2d.cpp:208-220)
bool hostWorldToMap(double ox, double oy, double res, uint32_t nx, uint32_t ny, double wx, double wy, uint32_t& mx, uint32_t& my) {
  if (wx < ox || wy < oy) return false;
  const double fx = (wx - ox) / res, fy = (wy - oy) / res;
  if (!(fx < 2147483648.0) || !(fy < 2147483648.0)) return false;
  mx = (uint32_t)(int)fx;
  my = (uint32_t)(int)fy;
  return mx < nx && my < ny;
}
// FootprintHelper::getLineCells (footprint_helper.cpp:51-124)
void tpLineCells(int x0, int x1, int y0, int y1, std::vector<TpCell>& pts) {
  int deltax = abs(x1 - x0), deltay = abs(y1 - y0);
  int x = x0, y = y0;
  int xinc1, xinc2, yinc1, yinc2, den, num, numadd, numpixels;
  xinc1 = xinc2 = (x1 >= x0) ? 1 : -1;
  yinc1 = yinc2 = (y1 >= y0) ? 1 : -1;
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
  for (int curpixel = 0; curpixel <= numpixels; curpixel++) {
    pts.push_back(TpCell{x, y});
    num += numadd;
    if (num >= den) {
      num -= den;
      x += xinc1;
      y += yinc1;
    }
    x += xinc2;
    y += yinc2;
  }
}
// FootprintHelper::getFillCells (:127-181)
void tpFillCells(std::vector<TpCell>& fp) {
  unsigned int i = 0;
  while (i < fp.size() - 1) {
    if (fp[i].x > fp[i + 1].x) {
      std::swap(fp[i], fp[i + 1]);
      if (i > 0) --i;
    } else {
      ++i;
    }
  }
  i = 0;
  TpCell min_pt, max_pt;
  const unsigned int min_x = fp[0].x, max_x = fp[fp.size() - 1].x;
  for (unsigned int x = min_x; x <= max_x; ++x) {
    if (i >= fp.size() - 1) break;
    if (fp[i].y < fp[i + 1].y) {
      min_pt = fp[i];
      max_pt = fp[i + 1];
    } else {
      min_pt = fp[i + 1];
      max_pt = fp[i];
    }
    i += 2;
    while (i < fp.size() && (unsigned int)fp[i].x == x) {
      if (fp[i].y < min_pt.y)
        min_pt = fp[i];
      else if (fp[i].y > max_pt.y)
        max_pt = fp[i];
      ++i;
    }
    for (unsigned int y = min_pt.y; y < (unsigned int)max_pt.y; ++y) fp.push_back(TpCell{(int)x, (int)y});
  }
}
// FootprintHelper::getFootprintCells(pos, spec, costmap, fill = true) (:186-258)
void tpFootprintCells(const float pos[3], const double* spec, uint32_t nfp, double ox, double oy, double res, uint32_t nx, uint32_t ny,
                      std::vector<TpCell>& cells) {
  const double x_i = pos[0], y_i = pos[1], theta_i = pos[2];
  cells.clear();
  if (nfp <= 1) {
    uint32_t mx, my;
    if (hostWorldToMap(ox, oy, res, nx, ny, x_i, y_i, mx, my)) cells.push_back(TpCell{(int)mx, (int)my});
    return;
  }
  const double cos_th = cos(theta_i), sin_th = sin(theta_i);
  uint32_t x0, y0, x1, y1;
  const uint32_t last = nfp - 1;
  auto vertex = [&](uint32_t i, uint32_t& mx, uint32_t& my) {
    const double wx = x_i + (spec[2 * i] * cos_th - spec[2 * i + 1] * sin_th);
    const double wy = y_i + (spec[2 * i] * sin_th + spec[2 * i + 1] * cos_th);
    return hostWorldToMap(ox, oy, res, nx, ny, wx, wy, mx, my);
  };
  for (uint32_t i = 0; i < last; ++i) {
    if (!vertex(i, x0, y0)) return;
    if (!vertex(i + 1, x1, y1)) return;
    tpLineCells((int)x0, (int)x1, (int)y0, (int)y1, cells);
  }
  if (!vertex(last, x0, y0)) return;
  if (!vertex(0, x1, y1)) return;
  tpLineCells((int)x0, (int)x1, (int)y0, (int)y1, cells);
  tpFillCells(cells);
}
// createTrajectories' sample enumeration (:537-665,777-780,871-874): every generateTrajectory call the
// reference could make this cycle, with the stage it belongs to
struct TpPlanned {
  double vx, vy, vth;
  double vth_unlimited;  // stage C: the loop variable before the min_in_place clamp
  int stage;             // 0 forward grid, 1 holonomic pair, 2 in-place rotation, 3 y velocities, 4 backing up
};
}  // namespace

static uint32_t tpMaxSamples(const navgpu_tp_config& c) {
  return (uint32_t)(c.vx_samples * c.vtheta_samples + 2 + c.vtheta_samples + c.n_y_vels + 1);
}

int navgpu_tp_configure(navgpu_fleet* f, const navgpu_tp_config* cfg_in) {
  if (!f || !cfg_in) return NAVGPU_ERR_INVALID;
  navgpu_tp_config c = *cfg_in;
  if (c.heading_scoring || c.simple_attractor) {
    g_last_error = "navgpu_tp_configure: heading_scoring / simple_attractor are not supported";
    return NAVGPU_ERR_INVALID;
  }
  if (c.n_y_vels < 0 || c.n_y_vels > 8 || !(c.sim_time > 0) || !(c.sim_granularity > 0) || !(c.angular_sim_granularity > 0))
    return NAVGPU_ERR_INVALID;
  if (c.vx_samples <= 0) c.vx_samples = 1;          // trajectory_planner.cpp:98-107
  if (c.vtheta_samples <= 0) c.vtheta_samples = 1;
  // step capacity: num_steps = int(max(vmag * sim_time / sim_granularity, |vtheta| / angular_sim_granularity) + 0.5)
  double ymax = 0.1;
  for (int i = 0; i < c.n_y_vels; ++i) ymax = std::max(ymax, fabs(c.y_vels[i]));
  const double vxmax = std::max(std::max(fabs(c.max_vel_x), fabs(c.min_vel_x)), std::max(fabs(c.backup_vel), 0.1));
  const double wmax = std::max(std::max(fabs(c.max_vel_th), fabs(c.min_vel_th)), fabs(c.min_in_place_vel_th));
  const double steps = std::max(hypot(vxmax, ymax) * c.sim_time / c.sim_granularity, wmax / c.angular_sim_granularity) + 1.5;
  if (steps > (double)f->pl.max_sim_steps) {
    g_last_error = "navgpu_tp_configure: trajectories need more points than max_sim_steps";
    return NAVGPU_ERR_CAPACITY;
  }
  HIP_TRY(waitStream(f->stream));
  TpDev& tp = f->tp;
  const uint32_t n = f->desc.n_instances;
  const uint32_t ms = tpMaxSamples(c);
  if (!f->tp_configured || ms > tp.max_samples) {
    f->release(tp.samples);
    f->release(tp.out);
    tp.samples = nullptr;
    tp.out = nullptr;
    int rc = f->alloc(&tp.samples, (size_t)n * ms * 3);
    if (rc) return rc;
    rc = f->alloc(&tp.out, (size_t)n * ms);
    if (rc) return rc;
    tp.max_samples = ms;
  }
  if (!f->tp_configured) {
    const uint32_t W = (f->cm.nx + 31) / 32;
    int rc = f->alloc(&tp.n_samples, n);
    if (!rc) rc = f->alloc(&tp.start, (size_t)n * 6);
    if (!rc) rc = f->alloc(&tp.winner, n);
    if (!rc) rc = f->alloc(&tp.points, (size_t)n * f->pl.max_sim_steps * 3);
    if (!rc) rc = f->alloc(&tp.within_count, n);
    if (!rc) rc = f->alloc(&tp.within_bits, (size_t)n * f->cm.ny * W);
    if (rc) return rc;
    f->tph.assign(n, navgpu_fleet::TpHost());
  }
  tp.cfg = c;
  f->tp_h_samples.assign((size_t)n * tp.max_samples * 3, 0.0);
  f->tp_h_out.assign((size_t)n * tp.max_samples, TpOut());
  f->tp_h_start.assign((size_t)n * 6, 0.0);
  f->tp_h_nsamples.assign(n, 0);
  f->tp_h_within_count.assign(n, 0);
  f->tp_h_winner.assign(n, -1);
  f->tp_configured = true;
  HIP_TRY(waitStream(f->stream));
  return NAVGPU_OK;
}

// the wavefront launch of the legacy planner: two grids, path_map_ optionally with within_robot bits
static int tpLaunchGrids(navgpu_fleet* f, uint32_t first, uint32_t count, bool with_within) {
  PlannerDev pl = f->pl;
  pl.bfs_grids = 2;
  pl.within = with_within ? f->tp.within_bits : nullptr;
  pl.cfg.allow_unknown = f->tp.cfg.allow_unknown;
  PROFILED(f, NAVGPU_K_BFS, launch_bfs(pl, first, count, f->stream));
  return NAVGPU_OK;
}
static int tpUploadPlans(navgpu_fleet* f, uint32_t first, uint32_t count) {
  PlannerDev& pl = f->pl;
  for (uint32_t i = first; i < first + count; ++i) {
    const std::vector<double>& p = f->tph[i].plan;
    const uint32_t np = (uint32_t)(p.size() / 2);
    if (np > pl.max_plan) return NAVGPU_ERR_CAPACITY;
    if (np) memcpy(&f->hp_plan[(size_t)i * pl.max_plan * 2], p.data(), sizeof(double) * p.size());
    f->hp_plan_cnt[i] = np;
  }
  HIP_TRY(hipMemcpyAsync(pl.plan + (size_t)first * pl.max_plan * 2, f->hp_plan + (size_t)first * pl.max_plan * 2,
                         sizeof(double) * 2 * (size_t)count * pl.max_plan, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(pl.plan_count + first, f->hp_plan_cnt + first, sizeof(uint32_t) * count, hipMemcpyHostToDevice, f->stream));
  return NAVGPU_OK;
}

int navgpu_tp_update_plan(navgpu_fleet* f, uint32_t instance, const double* plan_xy, uint32_t n, int32_t compute_dists) {
  if (!f || instance >= f->desc.n_instances || (n && !plan_xy)) return NAVGPU_ERR_INVALID;
  if (!f->tp_configured) return NAVGPU_ERR_STATE;
  if (n > f->pl.max_plan) return NAVGPU_ERR_CAPACITY;
  navgpu_fleet::TpHost& h = f->tph[instance];
  h.plan.assign(plan_xy, plan_xy + 2 * (size_t)n);
  if (n) {  // :480-487
    h.final_goal_x = plan_xy[2 * (size_t)(n - 1)];
    h.final_goal_y = plan_xy[2 * (size_t)(n - 1) + 1];
    h.final_goal_position_valid = true;
  } else {
    h.final_goal_position_valid = false;
  }
  if (compute_dists) {  // :489-499 (resetPathDist clears within_robot)
    int rc = tpUploadPlans(f, instance, 1);
    if (rc) return rc;
    rc = tpLaunchGrids(f, instance, 1, false);
    if (rc) return rc;
    HIP_TRY(waitStream(f->stream));
    return checkLaunch();
  }
  return NAVGPU_OK;
}

static inline bool tpFlag(const navgpu_tp_state& s, uint32_t bit) { return (s.flags & bit) != 0; }
static inline void tpSet(navgpu_tp_state& s, uint32_t bit, bool v) { s.flags = v ? (s.flags | bit) : (s.flags & ~bit); }

int navgpu_tp_find_best_path(navgpu_fleet* f, uint32_t first, uint32_t count, const navgpu_robot_state* states, navgpu_tp_result* results) {
  if (!f || !states || !results || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  if (!f->tp_configured) return NAVGPU_ERR_STATE;
  TpDev& tp = f->tp;
  const navgpu_tp_config& c = tp.cfg;
  const CostmapDev& cm = f->cm;
  const uint32_t ms = tp.max_samples;
  std::vector<std::vector<TpPlanned>> planned(count);
  std::vector<double> dvth(count, 0.0);
  std::vector<std::vector<TpCell>> cells(count);
  size_t max_cells = 0;
  // ---- host: footprint cells under the robot, velocity window, sample enumeration
  for (uint32_t k = 0; k < count; ++k) {
    const uint32_t inst = first + k;
    navgpu_fleet::TpHost& h = f->tph[inst];
    const float* pos = states[k].pos;
    const float* vel = states[k].vel;
    tpFootprintCells(pos, &f->h_fp_spec[(size_t)inst * kMaxFootprint * 2], f->h_fp_n[inst], f->h_origin[2 * inst],
                     f->h_origin[2 * inst + 1], cm.res, cm.nx, cm.ny, cells[k]);
    max_cells = std::max(max_cells, cells[k].size());
    const double x = pos[0], y = pos[1], vx = vel[0], vtheta = vel[2];
    const double acc_x = c.acc_lim_x, acc_theta = c.acc_lim_theta;
    double* st = &f->tp_h_start[(size_t)inst * 6];
    st[0] = x;
    st[1] = y;
    st[2] = pos[2];
    st[3] = vx;
    st[4] = vel[1];
    st[5] = vtheta;
    // :540-566
    double max_vel_x = c.max_vel_x, max_vel_theta, min_vel_x, min_vel_theta;
    if (h.final_goal_position_valid) {
      const double final_goal_dist = hypot(h.final_goal_x - x, h.final_goal_y - y);
      max_vel_x = std::min(max_vel_x, final_goal_dist / c.sim_time);
    }
    const double horizon = c.dwa ? c.sim_period : c.sim_time;
    max_vel_x = std::max(std::min(max_vel_x, vx + acc_x * horizon), c.min_vel_x);
    min_vel_x = std::max(c.min_vel_x, vx - acc_x * horizon);
    max_vel_theta = std::min(c.max_vel_th, vtheta + acc_theta * horizon);
    min_vel_theta = std::max(c.min_vel_th, vtheta - acc_theta * horizon);
    const double dvx = (max_vel_x - min_vel_x) / (c.vx_samples - 1);
    const double dvtheta = (max_vel_theta - min_vel_theta) / (c.vtheta_samples - 1);
    dvth[k] = dvtheta;
    std::vector<TpPlanned>& P = planned[k];
    double vx_samp = min_vel_x, vtheta_samp = min_vel_theta, vy_samp = 0.0;
    if (!tpFlag(h.st, NAVGPU_TP_ESCAPING)) {
      for (int i = 0; i < c.vx_samples; ++i) {  // :584-611
        vtheta_samp = 0;
        P.push_back(TpPlanned{vx_samp, vy_samp, vtheta_samp, 0, 0});
        vtheta_samp = min_vel_theta;
        for (int j = 0; j < c.vtheta_samples - 1; ++j) {
          P.push_back(TpPlanned{vx_samp, vy_samp, vtheta_samp, 0, 0});
          vtheta_samp += dvtheta;
        }
        vx_samp += dvx;
      }
      if (c.holonomic_robot) {  // :614-644
        P.push_back(TpPlanned{0.1, 0.1, 0.0, 0, 1});
        P.push_back(TpPlanned{0.1, -0.1, 0.0, 0, 1});
      }
    }
    vtheta_samp = min_vel_theta;  // :648-720
    for (int i = 0; i < c.vtheta_samples; ++i) {
      const double lim = vtheta_samp > 0 ? std::max(vtheta_samp, c.min_in_place_vel_th) : std::min(vtheta_samp, -1.0 * c.min_in_place_vel_th);
      P.push_back(TpPlanned{0.0, 0.0, lim, vtheta_samp, 2});
      vtheta_samp += dvtheta;
    }
    if (c.holonomic_robot)  // :771-817
      for (int i = 0; i < c.n_y_vels; ++i) P.push_back(TpPlanned{0.0, c.y_vels[i], 0.0, 0, 3});
    P.push_back(TpPlanned{c.backup_vel, 0.0, 0.0, 0, 4});  // :871-876
    if (P.size() > ms) return NAVGPU_ERR_CAPACITY;
    f->tp_h_nsamples[inst] = (uint32_t)P.size();
    for (size_t q = 0; q < P.size(); ++q) {
      double* d = &f->tp_h_samples[((size_t)inst * ms + q) * 3];
      d[0] = P[q].vx;
      d[1] = P[q].vy;
      d[2] = P[q].vth;
    }
  }
  // ---- within_robot cells (capacity grows with the footprint)
  if (max_cells > tp.max_within || !tp.within_cells) {
    HIP_TRY(waitStream(f->stream));
    f->release(tp.within_cells);
    tp.within_cells = nullptr;
    tp.max_within = (uint32_t)std::max<size_t>(max_cells * 2, 256);
    int rc = f->alloc(&tp.within_cells, (size_t)f->desc.n_instances * tp.max_within);
    if (rc) return rc;
    f->tp_h_within.assign((size_t)f->desc.n_instances * tp.max_within, 0);
  }
  for (uint32_t k = 0; k < count; ++k) {
    const uint32_t inst = first + k;
    f->tp_h_within_count[inst] = (uint32_t)cells[k].size();
    for (size_t q = 0; q < cells[k].size(); ++q)
      f->tp_h_within[(size_t)inst * tp.max_within + q] = (uint32_t)cells[k][q].y * cm.nx + (uint32_t)cells[k][q].x;
  }
  // ---- H2D + wavefronts + rollout of every planned sample
  int rc = tpUploadPlans(f, first, count);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(tp.within_cells + (size_t)first * tp.max_within, &f->tp_h_within[(size_t)first * tp.max_within],
                         sizeof(uint32_t) * (size_t)count * tp.max_within, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(tp.within_count + first, &f->tp_h_within_count[first], sizeof(uint32_t) * count, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(tp.samples + (size_t)first * ms * 3, &f->tp_h_samples[(size_t)first * ms * 3], sizeof(double) * 3 * (size_t)count * ms,
                         hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(tp.n_samples + first, &f->tp_h_nsamples[first], sizeof(uint32_t) * count, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(tp.start + (size_t)first * 6, &f->tp_h_start[(size_t)first * 6], sizeof(double) * 6 * count, hipMemcpyHostToDevice, f->stream));
  launch_tp_within(f->pl, tp, first, count, f->stream);
  rc = tpLaunchGrids(f, first, count, true);
  if (rc) return rc;
  PROFILED(f, NAVGPU_K_SCORE, launch_tp_rollout(f->pl, tp, first, count, 0, f->stream));
  HIP_TRY(hipMemcpyAsync(&f->tp_h_out[(size_t)first * ms], tp.out + (size_t)first * ms, sizeof(TpOut) * (size_t)count * ms, hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(waitStream(f->stream));
  rc = checkLaunch();
  if (rc) return rc;
  // ---- host: the reference's sequential selection (createTrajectories :568-905) over the per-sample results
  for (uint32_t k = 0; k < count; ++k) {
    const uint32_t inst = first + k;
    navgpu_fleet::TpHost& h = f->tph[inst];
    navgpu_tp_state& S = h.st;
    const std::vector<TpPlanned>& P = planned[k];
    const TpOut* O = &f->tp_h_out[(size_t)inst * ms];
    const double x = states[k].pos[0], y = states[k].pos[1], theta = states[k].pos[2];
    const double dvtheta = dvth[k];
    h.made.clear();
    int best = -1;        // index into P of best_traj (-1: the initial best_traj with cost -1)
    int best_made = -1;   // its position among the calls actually made
    double best_cost = -1.0, best_xv = 0, best_yv = 0, best_thv = 0;
    auto made = [&](size_t q) {
      h.made.push_back(navgpu_tp_sample{P[q].vx, P[q].vy, P[q].vth, O[q].cost, O[q].n_points, 0});
    };
    auto take = [&](size_t q) {  // always called right after made(q)
      best = (int)q;
      best_made = (int)h.made.size() - 1;
      best_cost = O[q].cost;
      best_xv = P[q].vx;
      best_yv = P[q].vy;
      best_thv = P[q].vth;
    };
    size_t q = 0;
    for (; q < P.size() && P[q].stage <= 1; ++q) {  // forward grid and the two holonomic samples: strict improvement
      made(q);
      if (O[q].cost >= 0 && (O[q].cost < best_cost || best_cost < 0)) take(q);
    }
    double heading_dist = DBL_MAX;
    for (; q < P.size() && P[q].stage == 2; ++q) {  // in-place rotations :654-719
      made(q);
      const double vtheta_samp = P[q].vth_unlimited;
      if (O[q].cost >= 0 && (O[q].cost <= best_cost || best_cost < 0 || best_yv != 0.0) &&
          (vtheta_samp > dvtheta || vtheta_samp < -1 * dvtheta)) {
        if (O[q].ahead_ok) {
          const double ahead_gdist = O[q].ahead;
          if (ahead_gdist < heading_dist) {
            if (vtheta_samp < 0 && !tpFlag(S, NAVGPU_TP_STUCK_LEFT)) {
              take(q);
              heading_dist = ahead_gdist;
            } else if (vtheta_samp > 0 && !tpFlag(S, NAVGPU_TP_STUCK_RIGHT)) {
              take(q);
              heading_dist = ahead_gdist;
            }
          }
        }
      }
    }
    auto resetOscillationIfMoved = [&]() {
      const double dist = hypot(x - S.prev_x, y - S.prev_y);
      if (dist > c.oscillation_reset_dist)
        S.flags &= ~(NAVGPU_TP_ROTATING_LEFT | NAVGPU_TP_ROTATING_RIGHT | NAVGPU_TP_STRAFE_LEFT | NAVGPU_TP_STRAFE_RIGHT |
                     NAVGPU_TP_STUCK_LEFT | NAVGPU_TP_STUCK_RIGHT | NAVGPU_TP_STUCK_LEFT_STRAFE | NAVGPU_TP_STUCK_RIGHT_STRAFE);
    };
    auto resetEscapeIfMoved = [&]() {
      const double dist = hypot(x - S.escape_x, y - S.escape_y);
      if (dist > c.escape_reset_dist || fabs(navgpu_shortest_angular_distance(S.escape_theta, theta)) > c.escape_reset_theta)
        tpSet(S, NAVGPU_TP_ESCAPING, false);
    };
    bool finished = false;
    if (best_cost >= 0) {  // :722-768
      if (!(best_xv > 0)) {
        if (best_thv < 0) {
          if (tpFlag(S, NAVGPU_TP_ROTATING_RIGHT)) tpSet(S, NAVGPU_TP_STUCK_RIGHT, true);
          tpSet(S, NAVGPU_TP_ROTATING_RIGHT, true);
        } else if (best_thv > 0) {
          if (tpFlag(S, NAVGPU_TP_ROTATING_LEFT)) tpSet(S, NAVGPU_TP_STUCK_LEFT, true);
          tpSet(S, NAVGPU_TP_ROTATING_LEFT, true);
        } else if (best_yv > 0) {
          if (tpFlag(S, NAVGPU_TP_STRAFE_RIGHT)) tpSet(S, NAVGPU_TP_STUCK_RIGHT_STRAFE, true);
          tpSet(S, NAVGPU_TP_STRAFE_RIGHT, true);
        } else if (best_yv < 0) {
          if (tpFlag(S, NAVGPU_TP_STRAFE_LEFT)) tpSet(S, NAVGPU_TP_STUCK_LEFT_STRAFE, true);
          tpSet(S, NAVGPU_TP_STRAFE_LEFT, true);
        }
        S.prev_x = x;
        S.prev_y = y;
      }
      resetOscillationIfMoved();
      resetEscapeIfMoved();
      finished = true;
    }
    if (!finished) {
      for (; q < P.size() && P[q].stage == 3; ++q) {  // sideways :771-817
        made(q);
        const double vy_samp = P[q].vy;
        if (O[q].cost >= 0 && (O[q].cost <= best_cost || best_cost < 0)) {
          if (O[q].ahead_ok) {
            const double ahead_gdist = O[q].ahead;
            if (ahead_gdist < heading_dist) {
              if (vy_samp > 0 && !tpFlag(S, NAVGPU_TP_STUCK_LEFT_STRAFE)) {
                take(q);
                heading_dist = ahead_gdist;
              } else if (vy_samp < 0 && !tpFlag(S, NAVGPU_TP_STUCK_RIGHT_STRAFE)) {
                take(q);
                heading_dist = ahead_gdist;
              }
            }
          }
        }
      }
      if (best_cost >= 0) {  // :820-868 — the flags set here are not those of the block above
        if (!(best_xv > 0)) {
          if (best_thv < 0) {
            if (tpFlag(S, NAVGPU_TP_ROTATING_RIGHT)) tpSet(S, NAVGPU_TP_STUCK_RIGHT, true);
            tpSet(S, NAVGPU_TP_ROTATING_LEFT, true);
          } else if (best_thv > 0) {
            if (tpFlag(S, NAVGPU_TP_ROTATING_LEFT)) tpSet(S, NAVGPU_TP_STUCK_LEFT, true);
            tpSet(S, NAVGPU_TP_ROTATING_RIGHT, true);
          } else if (best_yv > 0) {
            if (tpFlag(S, NAVGPU_TP_STRAFE_RIGHT)) tpSet(S, NAVGPU_TP_STUCK_RIGHT_STRAFE, true);
            tpSet(S, NAVGPU_TP_STRAFE_LEFT, true);
          } else if (best_yv < 0) {
            if (tpFlag(S, NAVGPU_TP_STRAFE_LEFT)) tpSet(S, NAVGPU_TP_STUCK_LEFT_STRAFE, true);
            tpSet(S, NAVGPU_TP_STRAFE_RIGHT, true);
          }
          S.prev_x = x;
          S.prev_y = y;
        }
        resetOscillationIfMoved();
        resetEscapeIfMoved();
        finished = true;
      }
    }
    if (!finished) {  // :871-905 back up slowly, whatever the footprint check says
      while (q < P.size() && P[q].stage != 4) ++q;
      made(q);
      take(q);
      resetOscillationIfMoved();
      if (!tpFlag(S, NAVGPU_TP_ESCAPING) && best_cost > -2.0) {
        S.escape_x = x;
        S.escape_y = y;
        S.escape_theta = theta;
        tpSet(S, NAVGPU_TP_ESCAPING, true);
      }
      resetEscapeIfMoved();
      if (best_cost == -1.0) best_cost = 1.0;
    }
    navgpu_tp_result& r = results[k];
    r = navgpu_tp_result();
    r.n_samples = (int32_t)h.made.size();
    r.cost = best_cost;
    if (best >= 0) {
      r.xv = best_xv;
      r.yv = best_yv;
      r.thetav = best_thv;
      r.n_points = O[best].n_points;
      r.best_sample = best_made;
    } else {  // the initial best_traj: Trajectory() with cost -1 and no points (cannot happen: backing up always takes)
      r.best_sample = -1;
    }
    h.n_points = r.n_points;
    if (r.cost >= 0) {  // findBestPath :969-978
      r.drive[0] = r.xv;
      r.drive[1] = r.yv;
      r.drive[2] = r.thetav;
    }
    f->tp_h_winner[inst] = best;
  }
  // ---- second pass: the winner's points (published as the local plan)
  HIP_TRY(hipMemcpyAsync(tp.winner + first, &f->tp_h_winner[first], sizeof(int32_t) * count, hipMemcpyHostToDevice, f->stream));
  launch_tp_rollout(f->pl, tp, first, count, 1, f->stream);
  HIP_TRY(waitStream(f->stream));
  return checkLaunch();
}

int navgpu_tp_trajectory(navgpu_fleet* f, uint32_t instance, double* xyth, uint32_t cap) {
  if (!f || !xyth || instance >= f->desc.n_instances) return NAVGPU_ERR_INVALID;
  if (!f->tp_configured) return NAVGPU_ERR_STATE;
  const int n = f->tph[instance].n_points;
  if (n > (int)cap) return NAVGPU_ERR_CAPACITY;
  if (n > 0) {
    HIP_TRY(hipMemcpyAsync(xyth, f->tp.points + (size_t)instance * f->pl.max_sim_steps * 3, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, f->stream));
    HIP_TRY(waitStream(f->stream));
  }
  return n;
}

int navgpu_tp_samples(navgpu_fleet* f, uint32_t instance, navgpu_tp_sample* samples, uint32_t cap) {
  if (!f || instance >= f->desc.n_instances) return NAVGPU_ERR_INVALID;
  if (!f->tp_configured) return NAVGPU_ERR_STATE;
  const std::vector<navgpu_tp_sample>& m = f->tph[instance].made;
  if (samples) {
    if (m.size() > cap) return NAVGPU_ERR_CAPACITY;
    if (!m.empty()) memcpy(samples, m.data(), sizeof(navgpu_tp_sample) * m.size());
  }
  return (int)m.size();
}

int navgpu_tp_score_trajectory(navgpu_fleet* f, uint32_t instance, const double pose[3], const double vel[3], const double vs[3], double* cost) {
  if (!f || !pose || !vel || !vs || !cost || instance >= f->desc.n_instances) return NAVGPU_ERR_INVALID;
  if (!f->tp_configured) return NAVGPU_ERR_STATE;
  TpDev& tp = f->tp;
  const uint32_t ms = tp.max_samples;
  const double start[6] = {pose[0], pose[1], pose[2], vel[0], vel[1], vel[2]};
  const uint32_t one = 1;
  HIP_TRY(hipMemcpyAsync(tp.start + (size_t)instance * 6, start, sizeof(start), hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(tp.samples + (size_t)instance * ms * 3, vs, sizeof(double) * 3, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(tp.n_samples + instance, &one, sizeof(one), hipMemcpyHostToDevice, f->stream));
  HIP_TRY(waitStream(f->stream));  // the sources above live on this stack frame
  launch_tp_rollout(f->pl, tp, instance, 1, 0, f->stream);
  TpOut o;
  HIP_TRY(hipMemcpyAsync(&o, tp.out + (size_t)instance * ms, sizeof(o), hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(waitStream(f->stream));
  *cost = o.cost;
  return checkLaunch();
}

int navgpu_tp_get_state(navgpu_fleet* f, uint32_t first, uint32_t count, navgpu_tp_state* states) {
  if (!f || !states || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  if (!f->tp_configured) return NAVGPU_ERR_STATE;
  for (uint32_t k = 0; k < count; ++k) states[k] = f->tph[first + k].st;
  return NAVGPU_OK;
}
int navgpu_tp_set_state(navgpu_fleet* f, uint32_t first, uint32_t count, const navgpu_tp_state* states) {
  if (!f || !states || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  if (!f->tp_configured) return NAVGPU_ERR_STATE;
  for (uint32_t k = 0; k < count; ++k) f->tph[first + k].st = states[k];
  return NAVGPU_OK;
}

// ------------------------------------------------------------------------------------------------ measurement
int navgpu_profile_enable(navgpu_fleet* f, int32_t enable) {
  if (!f) return NAVGPU_ERR_INVALID;
  if (!enable && f->profiling) {
    int rc = f->foldEvents();
    if (rc) return rc;
  }
  f->profiling = enable != 0;
  return NAVGPU_OK;
}
int navgpu_profile_reset(navgpu_fleet* f) {
  if (!f) return NAVGPU_ERR_INVALID;
  int rc = f->foldEvents();
  if (rc) return rc;
  for (int k = 0; k < NAVGPU_K_COUNT; ++k) {
    f->prof_ms[k] = 0;
    f->prof_n[k] = 0;
  }
  return NAVGPU_OK;
}
int navgpu_profile_read(navgpu_fleet* f, int32_t k, double* total_ms, uint64_t* launches) {
  if (!f || k < 0 || k >= NAVGPU_K_COUNT) return NAVGPU_ERR_INVALID;
  int rc = f->foldEvents();
  if (rc) return rc;
  if (total_ms) *total_ms = f->prof_ms[k];
  if (launches) *launches = f->prof_n[k];
  return NAVGPU_OK;
}

}  // extern "C"
