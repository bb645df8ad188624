// C-ABI host side of libnavgpu.so (include/navgpu.h).  Owns device memory, the fleet's HIP
// stream and the launch sequence; the only arithmetic done here is the per-cycle, per-instance
// scalar bookkeeping the reference also does once per cycle on the host (footprint transform,
// nose goal, inflation cost table) — in fp64 with libm, like the reference.
// There is no CPU fallback: every data-parallel step is a HIP kernel.
#include "navgpu_fleet.h"

namespace navgpu {
thread_local std::string g_last_error;
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
  if ((d->layers & NAVGPU_LAYER_STATIC) && !d->rolling_window) A(cm.stat, (size_t)n * cm.cells_padded);  // (rolling: navgpu_static_set_rolling_map)
  if (d->layers & (NAVGPU_LAYER_OBSTACLE | NAVGPU_LAYER_VOXEL)) A(cm.obst, (size_t)n * cm.cells_padded);
  if (d->layers & NAVGPU_LAYER_VOXEL) A(cm.voxel, (size_t)n * cm.cells_padded);
  A(cm.lut, 66 * 66);
  A(cm.lut2, 256);
  A(cm.state, n);
  A(cm.shift, (size_t)n * 2);
  // What a cycle hands over lives in two blocks, laid out alike in device memory and in a pinned host mirror: a stage call
  // that covers the whole fleet is ONE host-to-device copy per block (a dozen small copies cost the stream ~40 us and the
  // host as many runtime calls); partial ranges copy array by array.
  {
    size_t off = 0;
    auto take = [&](size_t bytes) {
      const size_t at = off;
      off = (off + bytes + 255) & ~(size_t)255;
      return at;
    };
    const size_t o_pose = take(sizeof(double) * 3 * n), o_fpw = take(sizeof(double) * 2 * kMaxFootprint * n), o_fpn = take(sizeof(uint32_t) * n),
                 o_obs = take(sizeof(ObsCsr) * (size_t)n * cm.max_obs), o_cnt = take(sizeof(uint32_t) * n),
                 o_pts = take(sizeof(float) * 3 * (size_t)n * cm.max_points);
    f->cm_stage_bytes = off;
    uint8_t *d = nullptr, *h = nullptr;
    A(d, off);
    if ((rc = f->allocPinned(&h, off)) != 0) {
      navgpu_fleet_destroy(f);
      return rc;
    }
    f->cm_stage_dev = d;
    f->cm_stage_host = h;
    cm.pose = reinterpret_cast<double*>(d + o_pose);
    cm.fp_world = reinterpret_cast<double*>(d + o_fpw);
    cm.fp_n = reinterpret_cast<uint32_t*>(d + o_fpn);
    cm.obs = reinterpret_cast<ObsCsr*>(d + o_obs);
    cm.obs_count = reinterpret_cast<uint32_t*>(d + o_cnt);
    cm.points = reinterpret_cast<float*>(d + o_pts);
    f->hp_pose = reinterpret_cast<double*>(h + o_pose);
    f->hp_fpw = reinterpret_cast<double*>(h + o_fpw);
    f->hp_fpn = reinterpret_cast<uint32_t*>(h + o_fpn);
    f->hp_obs = reinterpret_cast<ObsCsr*>(h + o_obs);
    f->hp_cnt = reinterpret_cast<uint32_t*>(h + o_cnt);
    f->hp_pts = reinterpret_cast<float*>(h + o_pts);
  }
  if (d->rolling_window) {
    if (d->layers & NAVGPU_LAYER_STATIC) {  // StaticLayer::updateCosts' rolling branch: a transform per robot, identity until set
      A(cm.stat_tf, (size_t)n * 12);
      std::vector<double> ident((size_t)n * 12, 0.0);
      for (uint32_t i = 0; i < n; ++i) ident[(size_t)i * 12 + 0] = ident[(size_t)i * 12 + 4] = ident[(size_t)i * 12 + 8] = 1.0;
      if (hipMemcpy(cm.stat_tf, ident.data(), sizeof(double) * ident.size(), hipMemcpyHostToDevice) != hipSuccess) {
        navgpu_fleet_destroy(f);
        return NAVGPU_ERR_HIP;
      }
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
  AP(f->hp_used, n);
  AP(f->hp_shift, (size_t)n * 2);
  AP(f->hp_result, (size_t)n * 2);  // two slots: navgpu_planner_set_cycles_in_flight
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
  {  // the planner's staged inputs: second block (see above)
    size_t off = 0;
    auto take = [&](size_t bytes) {
      const size_t at = off;
      off = (off + bytes + 255) & ~(size_t)255;
      return at;
    };
    const size_t o_state = take(sizeof(navgpu_robot_state) * n), o_front = take(sizeof(double) * 2 * n), o_align = take(sizeof(int32_t) * n),
                 o_reach = take(sizeof(uint32_t) * n), o_pcnt = take(sizeof(uint32_t) * n), o_plan = take(sizeof(double) * 2 * (size_t)n * pl.max_plan);
    f->pl_stage_bytes = off;
    uint8_t *d = nullptr, *h = nullptr;
    A(d, off);
    if ((rc = f->allocPinned(&h, off)) != 0) {
      navgpu_fleet_destroy(f);
      return rc;
    }
    f->pl_stage_dev = d;
    f->pl_stage_host = h;
    pl.state = reinterpret_cast<navgpu_robot_state*>(d + o_state);
    pl.front_last = reinterpret_cast<double*>(d + o_front);
    pl.align_on = reinterpret_cast<int32_t*>(d + o_align);
    pl.bfs_reach = reinterpret_cast<uint32_t*>(d + o_reach);
    pl.plan_count = reinterpret_cast<uint32_t*>(d + o_pcnt);
    pl.plan = reinterpret_cast<double*>(d + o_plan);
    f->hp_state = reinterpret_cast<navgpu_robot_state*>(h + o_state);
    f->hp_front = reinterpret_cast<double*>(h + o_front);
    f->hp_align = reinterpret_cast<int32_t*>(h + o_align);
    f->hp_reach = reinterpret_cast<uint32_t*>(h + o_reach);
    f->hp_plan_cnt = reinterpret_cast<uint32_t*>(h + o_pcnt);
    f->hp_plan = reinterpret_cast<double*>(h + o_plan);
  }
  A(pl.fp_spec, (size_t)n * kMaxFootprint * 2);
  A(pl.fp_n, n);
  A(pl.axis_count, (size_t)n * 4);
  A(pl.path, (size_t)n * pl.cells);
  A(pl.goal, (size_t)n * pl.cells);
  A(pl.goal_front, (size_t)n * pl.cells);
  if (bfs_scratch_words(cm.nx, cm.ny)) A(pl.bfs_scratch, (size_t)n * bfs_scratch_words(cm.nx, cm.ny));
  pl.bfs_grids = 3;
  pl.within = nullptr;
  A(pl.bfs_box, (size_t)n * 8);
  A(pl.bfs_care, (size_t)n * kCareRows * kCareWords);
  A(pl.bfs_next_item, 4);
  A(pl.bfs_free, (size_t)n * cm.ny * ((cm.nx + 31) / 32));
  A(pl.bfs_levels, (size_t)n * 3);
  A(pl.bfs_order, (size_t)n * 3);
  pl.bfs_bounded = 0;
  A(pl.counters, (size_t)n * 2);
  A(pl.osc_flags, n);
  A(pl.osc_prev, (size_t)n * 3);
  A(pl.traj, (size_t)n * pl.max_sim_steps * 3);
#undef A
  f->h_origin.assign((size_t)n * 2, 0.0);
  f->h_fp_spec.assign((size_t)n * kMaxFootprint * 2, 0.0);
  f->h_fp_n.assign(n, 0);
  f->grid_partial.assign(n, 0);
  f->bounded_grids = true;  // navgpu_planner_set_bounded_map_grids
  f->h_box.assign((size_t)n * 4, 0);
  f->inputs_gen.assign(n, 0);
  f->cycle_gen.assign(n, 0);
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
  if (e == hipSuccess) e = hipMemsetAsync(pl.bfs_reach, 0, sizeof(uint32_t) * n, f->stream);
  if (e == hipSuccess) e = hipMemsetAsync(pl.bfs_levels, 0, sizeof(uint32_t) * 3 * n, f->stream);
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
  for (hipEvent_t e : {f->ev_cycle[0], f->ev_cycle[1], f->ev_cm_h2d, f->ev_pl_h2d})
    if (e) hipEventDestroy(e);
  for (void* p : f->allocs) hipFree(p);
  for (void* p : f->pinned) hipHostFree(p);
  if (f->stream) hipStreamDestroy(f->stream);
  delete f;
  return NAVGPU_OK;
}
int navgpu_sync(navgpu_fleet* f) {
  if (!f) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  HIP_TRY(waitStream(f->stream));
  return NAVGPU_OK;
}
void* navgpu_stream(navgpu_fleet* f) { return f ? (void*)f->stream : nullptr; }
int navgpu_fleet_set_alloc_limit(navgpu_fleet* f, uint64_t max_bytes) {
  if (!f) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  f->alloc_limit = (size_t)max_bytes;
  return NAVGPU_OK;
}

int navgpu_fleet_set_origin(navgpu_fleet* f, uint32_t first, uint32_t count, const double* xy) {
  if (!f || !xy || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  f->touchInputs(first, count);
  memcpy(&f->h_origin[(size_t)first * 2], xy, sizeof(double) * 2 * count);
  HIP_TRY(hipMemcpyAsync(f->cm.origin + (size_t)first * 2, xy, sizeof(double) * 2 * count, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(waitStream(f->stream));
  return NAVGPU_OK;
}

int navgpu_fleet_get_origin(navgpu_fleet* f, uint32_t first, uint32_t count, double* xy) {
  if (!f || !xy || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
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
  FleetGuard guard_(f);
  f->touchInputs(first, count);
  void* base;
  size_t elem, stride, used;
  int rc = gridInfo(f, grid, &base, &elem, &stride, &used);
  if (rc) return rc;
  if (grid >= NAVGPU_GRID_PATH)  // caller-supplied MapGrids are whatever the caller says they are
    for (uint32_t i = first; i < first + count; ++i) f->grid_partial[i] = 0;
  HIP_TRY(hipMemcpy2DAsync((char*)base + (size_t)first * stride * elem, stride * elem, host, used * elem, used * elem, count,
                           hipMemcpyHostToDevice, f->stream));
  HIP_TRY(waitStream(f->stream));
  return NAVGPU_OK;
}
int navgpu_grid_download(navgpu_fleet* f, int grid, uint32_t first, uint32_t count, void* host) {
  if (!f || !host || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  void* base;
  size_t elem, stride, used;
  int rc = gridInfo(f, grid, &base, &elem, &stride, &used);
  if (rc) return rc;
  if (grid >= NAVGPU_GRID_PATH && (rc = ensureCompleteGrids(f, first, count))) return rc;
  HIP_TRY(hipMemcpy2DAsync(host, used * elem, (char*)base + (size_t)first * stride * elem, stride * elem, used * elem, count,
                           hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(waitStream(f->stream));
  return NAVGPU_OK;
}
int navgpu_grid_device(navgpu_fleet* f, int grid, void** ptr, size_t* stride_bytes) {
  if (!f || !ptr) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  void* base;
  size_t elem, stride, used;
  int rc = gridInfo(f, grid, &base, &elem, &stride, &used);
  if (rc) return rc;
  if (grid >= NAVGPU_GRID_PATH && (rc = ensureCompleteGrids(f, 0, f->desc.n_instances))) return rc;
  *ptr = base;
  if (stride_bytes) *stride_bytes = stride * elem;
  return NAVGPU_OK;
}
int navgpu_grid_reset(navgpu_fleet* f, int grid, uint32_t first, uint32_t count) {
  if (!f || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  f->touchInputs(first, count);
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

static int applyPendingShift(navgpu_fleet* f, uint32_t first, uint32_t count);
int navgpu_grid_reset_window(navgpu_fleet* f, int grid, uint32_t first, uint32_t count, uint32_t x0, uint32_t y0, uint32_t xn, uint32_t yn) {
  if (!f || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  CostmapDev& cm = f->cm;
  if (xn > cm.nx || yn > cm.ny || x0 > xn || y0 > yn) return NAVGPU_ERR_INVALID;  // (the reference's memset length underflows for xn < x0)
  uint8_t* base = nullptr;
  uint8_t value = 0;
  switch (grid) {
    case NAVGPU_GRID_MASTER: base = cm.master; value = cm.master_default; break;
    case NAVGPU_GRID_OBSTACLE: base = cm.obst; value = cm.obstacle_default; break;  // (a VoxelLayer's columns are not touched: it overrides resetMaps only)
    default: return NAVGPU_ERR_INVALID;
  }
  if (!base) return NAVGPU_ERR_STATE;
  f->touchInputs(first, count);
  if (xn > x0 && yn > y0) launch_reset_window(base + (size_t)first * cm.cells_padded, cm.cells_padded, count, cm.nx, x0, y0, xn, yn, value, f->stream);
  return checkLaunch();
}
int navgpu_layer_reset_bounding_box(navgpu_fleet* f, uint32_t first, uint32_t count, const double* boxes) {
  if (!f || !boxes || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  CostmapDev& cm = f->cm;
  if (!cm.obst) return NAVGPU_ERR_STATE;
  for (uint32_t i = 0; i < count; ++i)
    if (!(boxes[4 * i] <= boxes[4 * i + 2]) || !(boxes[4 * i + 1] <= boxes[4 * i + 3])) return NAVGPU_ERR_INVALID;
  {
    int rc = applyPendingShift(f, first, count);  // (a staged rolling-window origin is applied first: the box is in world coordinates)
    if (rc) return rc;
  }
  f->touchInputs(first, count);
  HIP_TRY(hipMemcpyAsync(f->d_bounds_tmp, boxes, sizeof(double) * 4 * count, hipMemcpyHostToDevice, f->stream));
  launch_reset_bounding_box(cm, first, count, f->d_bounds_tmp, f->stream);
  HIP_TRY(waitStream(f->stream));  // d_bounds_tmp and the caller's buffer are free again
  return checkLaunch();
}

// ---- footprint helpers (costmap_2d/src/footprint.cpp, costmap_math.{h,cpp}); pure host functions
static double fpDistance(double x0, double y0, double x1, double y1) { return hypot(x1 - x0, y1 - y0); }  // costmap_math.h:58-61
static double fpDistanceToLine(double pX, double pY, double x0, double y0, double x1, double y1) {  // costmap_math.cpp:32-67
  const double A = pX - x0, B = pY - y0, C = x1 - x0, D = y1 - y0;
  const double dot = A * C + B * D, len_sq = C * C + D * D;
  const double param = dot / len_sq;
  double xx, yy;
  if (param < 0) {
    xx = x0;
    yy = y0;
  } else if (param > 1) {
    xx = x1;
    yy = y1;
  } else {
    xx = x0 + param * C;
    yy = y0 + param * D;
  }
  return fpDistance(pX, pY, xx, yy);
}
int navgpu_footprint_radii(const double* xy, uint32_t n, double* inscribed, double* circumscribed) {
  if ((n && !xy) || !inscribed || !circumscribed) return NAVGPU_ERR_INVALID;
  double min_dist = DBL_MAX, max_dist = 0.0;  // footprint.cpp:41-67
  if (n > 2) {
    for (uint32_t i = 0; i < n; ++i) {
      const uint32_t j = (i + 1 == n) ? 0 : i + 1;  // the last edge closes the polygon (:60-65)
      const double vertex_dist = fpDistance(0.0, 0.0, xy[2 * i], xy[2 * i + 1]);
      const double edge_dist = fpDistanceToLine(0.0, 0.0, xy[2 * i], xy[2 * i + 1], xy[2 * j], xy[2 * j + 1]);
      min_dist = std::min(min_dist, std::min(vertex_dist, edge_dist));
      max_dist = std::max(max_dist, std::max(vertex_dist, edge_dist));
    }
  }
  *inscribed = min_dist;
  *circumscribed = max_dist;
  return NAVGPU_OK;
}
int navgpu_footprint_pad(double* xy, uint32_t n, double padding) {  // footprint.cpp:138-147
  if (n && !xy) return NAVGPU_ERR_INVALID;
  auto sign0 = [](double x) { return x < 0.0 ? -1.0 : (x > 0.0 ? 1.0 : 0.0); };
  for (uint32_t i = 0; i < 2 * n; ++i) xy[i] += sign0(xy[i]) * padding;
  return NAVGPU_OK;
}
int navgpu_footprint_from_radius(double radius, double* xy16) {  // footprint.cpp:150-167
  if (!xy16) return NAVGPU_ERR_INVALID;
  const int N = 16;
  for (int i = 0; i < N; ++i) {
    const double angle = i * 2 * M_PI / N;
    xy16[2 * i] = cos(angle) * radius;
    xy16[2 * i + 1] = sin(angle) * radius;
  }
  return NAVGPU_OK;
}

/* the device's sincos over host arrays (the floating-point contract, checkable against the host's libm) */
int navgpu_device_sincos(int32_t device, const double* theta, uint32_t n, double* sin_out, double* cos_out) {
  if (!theta || !sin_out || !cos_out) return NAVGPU_ERR_INVALID;
  if (n == 0) return NAVGPU_OK;
  HIP_TRY(hipSetDevice(device));
  double* d = nullptr;
  HIP_TRY(hipMalloc(&d, (size_t)3 * n * sizeof(double)));
  int rc = NAVGPU_OK;
  if (hipMemcpy(d, theta, (size_t)n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = NAVGPU_ERR_HIP;
  if (rc == NAVGPU_OK) {
    launch_sincos(d, n, d + n, d + 2 * (size_t)n, nullptr);
    if (hipMemcpy(sin_out, d + n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(cos_out, d + 2 * (size_t)n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
      rc = NAVGPU_ERR_HIP;
  }
  hipFree(d);
  return rc;
}

/* Costmap2DPublisher view of a window of the master grid */
int navgpu_costmap_export(navgpu_fleet* f, uint32_t instance, uint32_t x0, uint32_t y0, uint32_t xn, uint32_t yn, int8_t* out) {
  if (!f || !out || instance >= f->desc.n_instances) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  const CostmapDev& cm = f->cm;
  if (x0 >= xn || y0 >= yn || xn > cm.nx || yn > cm.ny) return NAVGPU_ERR_INVALID;
  if (!f->d_occ) {
    int rc = f->alloc(&f->d_occ, cm.cells);
    if (rc) return rc;
  }
  const uint32_t w = xn - x0, h = yn - y0;
  launch_export_window(cm.master + (size_t)instance * cm.cells_padded, cm.nx, x0, y0, w, h, f->d_occ, f->stream);
  HIP_TRY(hipMemcpyAsync(out, f->d_occ, (size_t)w * h, hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(waitStream(f->stream));
  return checkLaunch();
}

// ------------------------------------------------------------------------------------------------ layers
int navgpu_static_set_map(navgpu_fleet* f, uint32_t first, uint32_t count, const int8_t* occ, int32_t track_unknown_space,
                          int32_t use_maximum, int32_t trinary, int32_t lethal_cost_threshold, int32_t unknown_cost_value) {
  if (!f || !occ || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
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

int navgpu_static_set_rolling_map(navgpu_fleet* f, const int8_t* occ, uint32_t size_x, uint32_t size_y, double resolution, double origin_x,
                                  double origin_y, int32_t track_unknown_space, int32_t use_maximum, int32_t trinary, int32_t lethal_cost_threshold,
                                  int32_t unknown_cost_value) {
  if (!f || !occ || size_x == 0 || size_y == 0 || !(resolution > 0)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  CostmapDev& cm = f->cm;
  if (!f->desc.rolling_window || !(cm.layers & NAVGPU_LAYER_STATIC) || !cm.stat_tf) {
    g_last_error = "navgpu_static_set_rolling_map needs a rolling_window fleet with NAVGPU_LAYER_STATIC (else: navgpu_static_set_map)";
    return NAVGPU_ERR_STATE;
  }
  const size_t cells = (size_t)size_x * size_y;
  if (cells > 0xFFFFFFFFull) return NAVGPU_ERR_CAPACITY;
  HIP_TRY(waitStream(f->stream));
  int8_t* d_in = nullptr;
  uint8_t* d_map = nullptr;
  int rc = f->alloc(&d_in, cells);
  if (rc) return rc;
  if ((rc = f->alloc(&d_map, cells))) {
    f->release(d_in);
    return rc;
  }
  const int lethal = std::max(std::min(lethal_cost_threshold, 100), 0);  // static_layer.cpp:80
  hipError_t e = hipMemcpyAsync(d_in, occ, cells, hipMemcpyHostToDevice, f->stream);
  if (e == hipSuccess) {
    launch_static_interpret(d_map, d_in, (uint32_t)cells, (uint32_t)cells, 1, track_unknown_space, trinary, lethal, unknown_cost_value, f->stream);
    e = waitStream(f->stream);
  }
  f->release(d_in);
  if (e != hipSuccess) {  // nothing of the previous map has been touched yet
    f->release(d_map);
    g_last_error = std::string("navgpu_static_set_rolling_map: ") + hipGetErrorString(e);
    return NAVGPU_ERR_HIP;
  }
  if (cm.stat_roll) f->release(cm.stat_roll);
  cm.stat_roll = d_map;
  cm.stat_nx = size_x;
  cm.stat_ny = size_y;
  cm.stat_res = resolution;
  cm.stat_ox = origin_x;
  cm.stat_oy = origin_y;
  cm.static_use_maximum = use_maximum;
  cm.static_received = 1;
  f->touchInputs(0, f->desc.n_instances);
  return checkLaunch();
}

int navgpu_static_set_transform(navgpu_fleet* f, uint32_t first, uint32_t count, const double* m) {
  if (!f || !m || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  if (!f->cm.stat_tf) return NAVGPU_ERR_STATE;
  HIP_TRY(hipMemcpyAsync(f->cm.stat_tf + (size_t)first * 12, m, sizeof(double) * 12 * count, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(waitStream(f->stream));  // the caller's buffer is free on return
  f->touchInputs(first, count);
  return NAVGPU_OK;
}

int navgpu_obstacle_configure(navgpu_fleet* f, const navgpu_obstacle_params* p) {
  if (!f || !p) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
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
  FleetGuard guard_(f);
  CostmapDev& cm = f->cm;
  // the exact-EDT kernel takes the max over candidate seeds; that equals "cost of the nearest
  // seed" only if the table is monotone in distance — true for cost_scaling_factor >= 0.
  if (p->cost_scaling_factor < 0) {
    g_last_error = "cost_scaling_factor < 0 is not supported";
    return NAVGPU_ERR_INVALID;
  }
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
  // reference-order mode: cached_distances_ (hypot by this host's libm, as the reference builds them) and the
  // priority queue's storage - every cell is pushed at most once by each of its four neighbours, plus once as a seed.
  // Everything that can fail (limits, allocations) comes BEFORE the first write to the configuration in use: a failed
  // reconfigure leaves the previous one intact.
  std::vector<uint16_t> dist_lut((size_t)n * n);
  const int want_pq = p->priority_queue_order ? 1 : 0;
  if (want_pq) {
    // the heap only ever COMPARES cached_distances_ (inflation_layer.h:82-85) and tests them against the cell radius
    // (inflation_layer.cpp:286): their ranks among the distinct values carry exactly that, in two bytes
    std::vector<double> d((size_t)n * n);
    for (uint32_t i = 0; i < n; ++i)
      for (uint32_t j = 0; j < n; ++j) d[(size_t)i * n + j] = hypot(i, j);
    std::vector<double> uniq(d);
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    for (size_t k = 0; k < d.size(); ++k)
      dist_lut[k] = d[k] > (double)R ? (uint16_t)0xFFFFu : (uint16_t)(std::lower_bound(uniq.begin(), uniq.end(), d[k]) - uniq.begin());
    if (cm.nx > 65535 || cm.ny > 65535) return NAVGPU_ERR_CAPACITY;
    int rc;
    if (!cm.dist_lut && (rc = f->alloc(&cm.dist_lut, (size_t)66 * 66))) return rc;
    if (!cm.pq_seen && (rc = f->alloc(&cm.pq_seen, (size_t)f->desc.n_instances * cm.cells_padded))) return rc;
    if (!cm.pq_heap) {
      const uint64_t cap = (uint64_t)5 * cm.cells;
      if ((rc = f->alloc(&cm.pq_heap, (size_t)f->desc.n_instances * cap))) return rc;
      cm.pq_cap = cap;
    }
  }
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
  // the tables are read by kernels already queued on the stream: upload behind them, commit the scalars only once all
  // three uploads have been accepted (the kernels of later calls take cm by value)
  HIP_TRY(waitStream(f->stream));
  HIP_TRY(hipMemcpyAsync(cm.lut, lut.data(), lut.size(), hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(cm.lut2, lut2.data(), lut2.size(), hipMemcpyHostToDevice, f->stream));
  if (want_pq) HIP_TRY(hipMemcpyAsync(cm.dist_lut, dist_lut.data(), sizeof(uint16_t) * dist_lut.size(), hipMemcpyHostToDevice, f->stream));
  HIP_TRY(waitStream(f->stream));
  cm.infl_pq = want_pq;
  cm.lut2_ok = lut2_ok ? 1 : 0;
  const bool changed = !f->inflation_configured || f->infl.inflation_radius != p->inflation_radius ||
                       f->infl.cost_scaling_factor != p->cost_scaling_factor || f->infl.inscribed_radius != p->inscribed_radius ||
                       f->infl.enabled != p->enabled || f->infl.priority_queue_order != p->priority_queue_order;
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

// The k_score image geometry (window, footprint chunk, heading tables) of a planner configuration `pl` with the given footprints,
// worked out IN `pl` - a copy the caller commits only once everything the new configuration needs has been allocated.
static void planWindow(const navgpu_fleet* f, PlannerDev& pl, const std::vector<uint32_t>& fp_n, const std::vector<double>& fp_spec, double fp_radius) {
  // costmap window staged in LDS by k_score: everything a trajectory's footprint or shifted
  // point can touch.  Results never depend on it (cells outside fall back to global loads).
  const navgpu_dwa_config& c = pl.cfg;
  double vmax;
  if (c.max_trans_vel >= 0)
    vmax = c.max_trans_vel + 1e-4;
  else
    vmax = hypot(std::max(fabs(c.min_vel_x), fabs(c.max_vel_x)), std::max(fabs(c.min_vel_y), fabs(c.max_vel_y)));
  double reach = vmax * c.sim_time + std::max(fp_radius, fabs(c.forward_point_distance));
  double cells = ceil(reach / pl.res) + 2;
  uint32_t win = (uint32_t)std::min(cells * 2 + 1, 240.0);
  pl.win = win;
  pl.fp_rcells = (uint32_t)ceil(fp_radius / pl.res) + 1;  // vertex cells lie within this Chebyshev radius of the centre cell
  {
    // ... and every OUTLINE cell q within a disc around the centre cell c.  In cells: a vertex cell is floor(P + r_i) with |r_i| <= r =
    // fp_radius / res, c = floor(P), so a_i - c lies within (-1, 1)^2 of r_i; LineIterator's cells keep within 1/2 cell (minor axis) of
    // the segment between their end cells (line_iterator.h:110-127: offset floor((den / 2 + k numadd) / den) against k numadd / den);
    // hence |q - c| < r + |(1, 1.5)| = r + 1.803.  The screens' structuring element is that disc clipped to the Chebyshev square.
    const double rho = fp_radius / pl.res + 1.803 + 0.01;
    for (uint32_t d = 0; d < 32; ++d) {
      pl.fp_halfw[d] = 0xFF;
      if (d <= pl.fp_rcells && (double)d <= rho) pl.fp_halfw[d] = (uint8_t)std::min<double>(pl.fp_rcells, floor(sqrt(rho * rho - (double)d * d)));
    }
  }
  {  // longest footprint edge in cells (both end cells included) over all instances
    double max_edge = 0.0;
    for (uint32_t i = 0; i < f->desc.n_instances; ++i) {
      const uint32_t nv = fp_n[i];
      const double* q = &fp_spec[(size_t)i * kMaxFootprint * 2];
      for (uint32_t a = 0; a < nv; ++a) {
        const uint32_t b = (a + 1) % nv;
        max_edge = std::max(max_edge, std::max(fabs(q[2 * a] - q[2 * b]), fabs(q[2 * a + 1] - q[2 * b + 1])));
        max_edge = std::max(max_edge, hypot(q[2 * a] - q[2 * b], q[2 * a + 1] - q[2 * b + 1]));
      }
    }
    pl.fp_chunk = (uint32_t)ceil(max_edge / pl.res) + 1;
  }
  // shared heading tables (k_score<TABLES>): constant velocity + fixed step count only
  uint32_t max_nfp = 0;
  for (uint32_t v : fp_n) max_nfp = std::max(max_nfp, v);
  pl.use_tables = 0;
  if (c.use_dwa && c.discretize_by_time && max_nfp <= 8) {
    pl.tab_steps = (uint32_t)ceil(c.sim_time / c.sim_granularity);
    pl.tab_dt = c.sim_time / (int)pl.tab_steps;
    pl.tab_nfp = max_nfp;
    pl.tab_nth = (uint32_t)std::max(c.vth_samples, 2) + 1;
    pl.tab_rows = pl.tab_steps >= 1 && pl.tab_steps <= pl.max_sim_steps ? score_table_rows(pl, win) : 0;
    if (pl.tab_rows >= 1) pl.use_tables = 1;
  }
}

// A buffer for the per-robot LDS images of k_score that fits geometry `pl`: *fresh = a new allocation (the caller swaps it in and
// releases the old one once nothing can fail any more), or nullptr when the current buffer is large enough.
static int allocPrep(navgpu_fleet* f, const PlannerDev& pl, uint8_t** fresh, uint32_t* stride) {
  PlannerDev tmp = pl;
  tmp.use_tables = 1;  // upper bound: the image with tables, whatever the launch decides
  const size_t need = (score_prep_slot_bytes(tmp) + 255) & ~(size_t)255;
  *fresh = nullptr;
  *stride = f->pl.prep_stride;
  if (f->pl.prep && need <= f->pl.prep_stride) return NAVGPU_OK;
  int rc = f->alloc(fresh, (size_t)f->desc.n_instances * need);
  if (rc != NAVGPU_OK) return rc;
  *stride = (uint32_t)need;
  return NAVGPU_OK;
}

// All-or-nothing (the threading contract of navgpu.h): the new footprint, the window it implies and the image buffer that window
// needs are prepared on the side; the fleet changes only when nothing can fail any more.
int navgpu_set_footprint(navgpu_fleet* f, uint32_t first, uint32_t count, const double* xy, uint32_t nv) {
  if (!f || !f->rangeOk(first, count) || (nv && !xy)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  if (nv > f->desc.max_footprint || nv > (uint32_t)kMaxFootprint) return NAVGPU_ERR_CAPACITY;
  std::vector<uint32_t> fp_n = f->h_fp_n;
  std::vector<double> fp_spec = f->h_fp_spec;
  double fp_radius = f->fp_radius;
  for (uint32_t i = first; i < first + count; ++i) {
    fp_n[i] = nv;
    for (uint32_t k = 0; k < nv * 2; ++k) fp_spec[(size_t)i * kMaxFootprint * 2 + k] = xy[k];
  }
  for (uint32_t k = 0; k < nv; ++k) fp_radius = std::max(fp_radius, hypot(xy[2 * k], xy[2 * k + 1]));
  if (fp_radius / f->cm.res > 500) return NAVGPU_ERR_CAPACITY;  // polygon column span must fit the 1024-column LDS table
  PlannerDev np = f->pl;
  uint8_t* fresh_prep = nullptr;
  uint32_t stride = f->pl.prep_stride;
  if (f->planner_configured) {
    planWindow(f, np, fp_n, fp_spec, fp_radius);
    int rc = allocPrep(f, np, &fresh_prep, &stride);
    if (rc != NAVGPU_OK) return rc;
  }
  // ---- commit
  hipError_t e = waitStream(f->stream);  // queued kernels took their PlannerDev by value: drain before buffers they use go
  if (e == hipSuccess)
    e = hipMemcpyAsync(f->pl.fp_spec + (size_t)first * kMaxFootprint * 2, &fp_spec[(size_t)first * kMaxFootprint * 2], sizeof(double) * kMaxFootprint * 2 * count,
                       hipMemcpyHostToDevice, f->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(f->pl.fp_n + first, &fp_n[first], sizeof(uint32_t) * count, hipMemcpyHostToDevice, f->stream);
  if (e == hipSuccess) e = waitStream(f->stream);  // (the copies read the local vectors)
  if (e != hipSuccess) {
    if (fresh_prep) f->release(fresh_prep);
    g_last_error = std::string("navgpu_set_footprint: ") + hipGetErrorString(e);
    return NAVGPU_ERR_HIP;
  }
  f->h_fp_n.swap(fp_n);
  f->h_fp_spec.swap(fp_spec);
  f->fp_radius = fp_radius;
  if (f->planner_configured) {
    uint8_t* keep = fresh_prep ? fresh_prep : f->pl.prep;
    if (fresh_prep) f->release(f->pl.prep);
    f->pl = np;
    f->pl.prep = keep;
    f->pl.prep_stride = stride;
  }
  return NAVGPU_OK;
}

int navgpu_costmap_stage(navgpu_fleet* f, uint32_t first, uint32_t count, const double* poses, const navgpu_observation* obs,
                         uint32_t n_obs, const float* points, uint32_t n_points_total) {
  if (!f || !poses || !f->rangeOk(first, count) || (n_obs && (!obs || (!points && n_points_total)))) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  CostmapDev& cm = f->cm;
  if (f->desc.rolling_window && f->shift_pending) return NAVGPU_ERR_STATE;  // previous stage not consumed by an update yet
  if (f->desc.rolling_window) f->touchInputs(first, count);  // the origins move now
  HIP_TRY(f->waitMirrors(f->ev_cm_h2d, f->ev_cm_set));  // the pinned mirrors may still feed an earlier copy
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
  memcpy(f->hp_fpn + first, &f->h_fp_n[first], sizeof(uint32_t) * count);
  if (first == 0 && count == f->desc.n_instances) {  // the whole fleet: one copy of the block
    HIP_TRY(hipMemcpyAsync(f->cm_stage_dev, f->cm_stage_host, f->cm_stage_bytes, hipMemcpyHostToDevice, f->stream));
  } else {
    HIP_TRY(hipMemcpyAsync(cm.pose + (size_t)first * 3, f->hp_pose + (size_t)first * 3, sizeof(double) * 3 * count, hipMemcpyHostToDevice, f->stream));
    HIP_TRY(hipMemcpyAsync(cm.obs + (size_t)first * cm.max_obs, f->hp_obs + (size_t)first * cm.max_obs, sizeof(ObsCsr) * (size_t)count * cm.max_obs, hipMemcpyHostToDevice, f->stream));
    HIP_TRY(hipMemcpyAsync(cm.obs_count + first, f->hp_cnt + first, sizeof(uint32_t) * count, hipMemcpyHostToDevice, f->stream));
    HIP_TRY(hipMemcpyAsync(cm.points + (size_t)first * cm.max_points * 3, f->hp_pts + (size_t)first * cm.max_points * 3, sizeof(float) * 3 * (size_t)count * cm.max_points, hipMemcpyHostToDevice, f->stream));
    HIP_TRY(hipMemcpyAsync(cm.fp_world + (size_t)first * kMaxFootprint * 2, f->hp_fpw + (size_t)first * kMaxFootprint * 2, sizeof(double) * 2 * kMaxFootprint * (size_t)count, hipMemcpyHostToDevice, f->stream));
    HIP_TRY(hipMemcpyAsync(cm.fp_n + first, f->hp_fpn + first, sizeof(uint32_t) * count, hipMemcpyHostToDevice, f->stream));
  }
  if (f->cycles_in_flight > 1) {
    HIP_TRY(hipEventRecord(f->ev_cm_h2d, f->stream));
    f->ev_cm_set = true;
  }
  return NAVGPU_OK;  // copies stay in flight on the fleet's stream; the kernels are ordered behind them
}

// rolling window: the origins staged by navgpu_costmap_stage are applied to the resident grids (Costmap2D::updateOrigin,
// costmap_2d.cpp:264-313; VoxelLayer::updateOrigin, voxel_layer.cpp:385-440) before anything reads or writes them
static int applyPendingShift(navgpu_fleet* f, uint32_t first, uint32_t count) {
  CostmapDev& cm = f->cm;
  if (!(f->desc.rolling_window && f->shift_pending)) return NAVGPU_OK;
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
  return NAVGPU_OK;
}

int navgpu_costmap_update(navgpu_fleet* f, uint32_t first, uint32_t count) {
  if (!f || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  f->touchInputs(first, count);
  CostmapDev& cm = f->cm;
  if ((cm.layers & NAVGPU_LAYER_INFLATION) && !f->inflation_configured) return NAVGPU_ERR_STATE;
  {
    int rc = applyPendingShift(f, first, count);
    if (rc) return rc;
  }
  PROFILED(f, NAVGPU_K_OBSTACLE, launch_obstacle(cm, first, count, nullptr, 0, f->stream));
  PROFILED(f, NAVGPU_K_MERGE, launch_merge(cm, first, count, nullptr, f->stream));
  if (cm.layers & NAVGPU_LAYER_INFLATION) PROFILED(f, NAVGPU_K_INFLATE, launch_inflate(cm, first, count, nullptr, f->stream));
  return checkLaunch();
}

int navgpu_costmap_bounds(navgpu_fleet* f, uint32_t first, uint32_t count, int32_t* boxes) {
  if (!f || !boxes || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  std::vector<InstCostmapState> st(count);
  HIP_TRY(hipMemcpyAsync(st.data(), f->cm.state + first, sizeof(InstCostmapState) * count, hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(waitStream(f->stream));
  for (uint32_t i = 0; i < count; ++i)
    for (int k = 0; k < 4; ++k) boxes[4 * i + k] = st[i].box[k];
  return NAVGPU_OK;
}

int navgpu_inflate(navgpu_fleet* f, uint32_t first, uint32_t count, const int32_t* boxes) {
  if (!f || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  f->touchInputs(first, count);
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
  FleetGuard guard_(f);
  if (!f->cm.obst) return NAVGPU_ERR_STATE;
  {
    int rc = applyPendingShift(f, first, count);  // ObstacleLayer::updateBounds :344-345: updateOrigin first
    if (rc) return rc;
  }
  HIP_TRY(hipMemcpyAsync(f->d_bounds_tmp, bounds, sizeof(double) * 4 * count, hipMemcpyHostToDevice, f->stream));
  PROFILED(f, NAVGPU_K_OBSTACLE, launch_obstacle(f->cm, first, count, f->d_bounds_tmp, 1, f->stream));
  HIP_TRY(hipMemcpyAsync(bounds, f->d_bounds_tmp, sizeof(double) * 4 * count, hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(waitStream(f->stream));
  return checkLaunch();
}

int navgpu_obstacle_update_costs(navgpu_fleet* f, uint32_t first, uint32_t count, const int32_t* boxes) {
  if (!f || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  f->touchInputs(first, count);
  const int32_t* d_boxes = nullptr;
  if (boxes) {
    HIP_TRY(hipMemcpyAsync(f->d_boxes_tmp, boxes, sizeof(int32_t) * 4 * count, hipMemcpyHostToDevice, f->stream));
    d_boxes = f->d_boxes_tmp;
  }
  PROFILED(f, NAVGPU_K_MERGE, launch_merge(f->cm, first, count, d_boxes, f->stream, true));
  if (boxes) HIP_TRY(waitStream(f->stream));
  return checkLaunch();
}

// ------------------------------------------------------------------------------------------------ planner

// ---- bounded MapGrid wavefronts ------------------------------------------------------------------------------------
// The critics read a MapGrid at trajectory points and at the forward point only (map_grid_cost_function.cpp:75-129),
// all within `reach` of the robot: speed bound x sim_time + forward_point_distance.  Every axis velocity a sample can
// take lies between the robot's velocity and the limits (simple_trajectory_generator.cpp:77-110, velocity_iterator.h;
// computeNewVelocities moves from the one towards the other), so |v_axis| <= max(|min|, |max|, |v|).  k_samples turns
// the reach into the robot's box and a wavefront stops once that box is settled (k_bfs_wave); what it leaves open is
// completed by ensureCompleteGrids before anybody else reads the grid.
static const int kBoxMarginCells = 4;  // cell rounding of pose and points, the touched-obstacle ring, slack
static double reachMetres(const navgpu_dwa_config& c, const float vel[3], const float* sample) {
  auto axis = [&](double lo, double hi, int a) {
    double b = std::max(std::max(fabs(lo), fabs(hi)), fabs((double)vel[a]));
    if (sample) b = std::max(fabs((double)sample[a]), fabs((double)vel[a]));
    return b;
  };
  const double bx = axis(c.min_vel_x, c.max_vel_x, 0), by = axis(c.min_vel_y, c.max_vel_y, 1);
  return hypot(bx, by) * c.sim_time * 1.001 + fabs(c.forward_point_distance);
}
static uint32_t bfsReachCells(const navgpu_fleet* f, const navgpu_robot_state& s, double goal_x, double goal_y) {
  if (!f->bounded_grids) return 0;  // (every wavefront kernel can stop at the robot's box)
  if (f->pl.mg_generic) return 0;  // (a sideways-shifted look-up leaves the box the reach is derived for: whole grids)
  const double reach = reachMetres(f->pl.cfg, s.vel, nullptr);
  const double cells = ceil(reach / f->pl.res) + kBoxMarginCells;
  if (!(cells < 32768.0)) return 0;
  // close to the goal the stop-and-rotate controller may take over and keep checking trajectories against these
  // grids for many cycles (latched_stop_rotate_controller.cpp:188-269): give it complete ones
  if (!(hypot(goal_x - s.pos[0], goal_y - s.pos[1]) > 2.0 * reach)) return 0;
  return (uint32_t)cells;
}
// the cell box k_samples derives from a reach (same arithmetic, fp64); false = robot not on the map = whole grid
static bool robotBox(const navgpu_fleet* f, uint32_t i, const float pos[3], uint32_t reach, int32_t box[4]) {
  const double ox = f->h_origin[2 * i], oy = f->h_origin[2 * i + 1], res = f->pl.res;
  const double wx = pos[0], wy = pos[1];
  if (!reach || !(wx >= ox) || !(wy >= oy)) return false;
  const double fx = (wx - ox) / res, fy = (wy - oy) / res;
  if (!(fx < 2147483648.0) || !(fy < 2147483648.0)) return false;
  const int64_t mx = (int)fx, my = (int)fy;
  if (mx >= (int64_t)f->pl.nx || my >= (int64_t)f->pl.ny) return false;
  box[0] = (int32_t)std::max<int64_t>(mx - reach, 0);
  box[1] = (int32_t)std::min<int64_t>(mx + reach, f->pl.nx - 1);
  box[2] = (int32_t)std::max<int64_t>(my - reach, 0);
  box[3] = (int32_t)std::min<int64_t>(my + reach, f->pl.ny - 1);
  return true;
}
// finish the grids of robots whose last wavefronts stopped early; fails when their inputs have changed since
extern "C++" int navgpu::ensureCompleteGrids(navgpu_fleet* f, uint32_t first, uint32_t count) {
  if (NAVGPU_DEBUG_ENV("NAVGPU_DEBUG_RAW_GRIDS")) return NAVGPU_OK;  // tool builds only (tools/probe_levels.py): look at what a bounded search left
  for (uint32_t i = first; i < first + count; ++i)
    if (f->grid_partial[i] && f->cycle_gen[i] != f->inputs_gen[i]) {
      g_last_error = "MapGrids of instance " + std::to_string(i) + " were searched inside the robot's box only and the costmap / plan they came from has changed; "
                     "read them before the next update or call navgpu_planner_set_bounded_map_grids(f, 0)";
      return NAVGPU_ERR_STATE;
    }
  PlannerDev pl = f->pl;
  pl.bfs_bounded = 0;
  for (uint32_t i = first; i < first + count;) {
    if (!f->grid_partial[i]) {
      ++i;
      continue;
    }
    uint32_t j = i;
    while (j < first + count && f->grid_partial[j]) f->grid_partial[j++] = 0;
    launch_bfs(pl, i, j - i, f->stream);
    i = j;
  }
  return checkLaunch();
}

// the staged reach depends on the configuration: a reconfigure (or the switch below) between stage and cycle re-derives it
static int restageReach(navgpu_fleet* f) {
  if (!f->planner_staged) return NAVGPU_OK;
  PlannerDev& pl = f->pl;
  HIP_TRY(waitStream(f->stream));
  const uint32_t n = f->desc.n_instances;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t np = f->hp_plan_cnt[i];
    if (!np) {  // never staged
      f->hp_reach[i] = 0;
      continue;
    }
    const double* last = &f->hp_plan[((size_t)i * pl.max_plan + np - 1) * 2];
    f->hp_reach[i] = bfsReachCells(f, f->hp_state[i], last[0], last[1]);
  }
  HIP_TRY(hipMemcpyAsync(pl.bfs_reach, f->hp_reach, sizeof(uint32_t) * n, hipMemcpyHostToDevice, f->stream));
  f->hp_dma_pending = true;
  if (f->cycles_in_flight > 1) {  // the next stage waits for THIS copy out of hp_reach, not for a marker recorded before it
    HIP_TRY(hipEventRecord(f->ev_pl_h2d, f->stream));
    f->ev_pl_set = true;
  }
  return NAVGPU_OK;
}

int navgpu_planner_set_bounded_map_grids(navgpu_fleet* f, int32_t enable) {
  if (!f) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  f->bounded_grids = enable != 0;
  return restageReach(f);
}
int navgpu_planner_set_map_grid_options(navgpu_fleet* f, int32_t critic, int32_t aggregation, double yshift) {
  if (!f || critic < 0 || critic > 3 || aggregation < 0 || aggregation > 2 || !std::isfinite(yshift)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  PlannerDev& pl = f->pl;
  pl.mg_agg[critic] = aggregation;
  pl.mg_yshift[critic] = yshift;
  pl.mg_generic = 0;
  for (int k = 0; k < 4; ++k)
    if (pl.mg_agg[k] != 0 || pl.mg_yshift[k] != 0.0) pl.mg_generic = 1;
  return restageReach(f);  // (grids a bounded cycle left behind: navgpu_planner_check_trajectory completes them first)
}
int navgpu_planner_wavefront_levels(navgpu_fleet* f, uint32_t first, uint32_t count, uint32_t* levels) {
  if (!f || !levels || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  HIP_TRY(hipMemcpyAsync(levels, f->pl.bfs_levels + (size_t)first * 3, sizeof(uint32_t) * 3 * count, hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(waitStream(f->stream));
  return NAVGPU_OK;
}

int navgpu_planner_configure(navgpu_fleet* f, const navgpu_dwa_config* c) {
  if (!f || !c) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  if (!(c->sim_time > 0) || !(c->sim_granularity > 0) || !(c->angular_sim_granularity > 0)) return NAVGPU_ERR_INVALID;
  PlannerDev& pl = f->pl;
  navgpu_dwa_config cfg = *c;
  if (cfg.vx_samples <= 0) cfg.vx_samples = 1;  // dwa_planner.cpp:89-105
  if (cfg.vy_samples <= 0) cfg.vy_samples = 1;
  if (cfg.vth_samples <= 0) cfg.vth_samples = 1;
  if (cfg.rollout_trig != 0 && cfg.rollout_trig != 1) return NAVGPU_ERR_INVALID;
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
  // capacity of the per-workgroup partial results: the 256-thread launch, and the table launch's row groups (each
  // rounds its share up to whole workgroups: at most one more per v_theta value, planner_score*.hip)
  const uint32_t score_blocks = 2 * ((max_samples + std::min(kScoreThreads, NAVGPU_SCORE_TAB_THREADS) - 1) / std::min(kScoreThreads, NAVGPU_SCORE_TAB_THREADS)) + at + 2;
  // All-or-nothing (the threading contract of navgpu.h: a reconfigure that fails leaves the previous configuration in force, and
  // the control thread may run the next cycle on it): the new configuration is put together in a COPY of the planner state, its
  // tables and image buffer are allocated on the side, and only when nothing can fail any more is the stream drained, the copy
  // committed and the old buffers released.
  PlannerDev np = pl;
  np.cfg = cfg;
  np.scale_path = pl.res * cfg.path_distance_bias * 0.5;  // DWAPlanner::reconfigure scales (dwa_planner.cpp:64-75)
  np.scale_goal = pl.res * cfg.goal_distance_bias * 0.5;
  np.scale_obstacle = pl.res * cfg.occdist_scale;
  const bool resize = max_axis != pl.max_axis || max_samples != pl.max_samples;
  float* n_axis = nullptr;
  double* n_part_cost = nullptr;
  int32_t* n_part_index = nullptr;
  double* n_sample_cost = nullptr;
  int32_t* n_sample_status = nullptr;
  uint8_t* n_prep = nullptr;
  uint32_t n_stride = pl.prep_stride;
  auto dropNew = [&] {
    f->release(n_axis);
    f->release(n_part_cost);
    f->release(n_part_index);
    f->release(n_sample_cost);
    f->release(n_sample_status);
    f->release(n_prep);
  };
  int rc = NAVGPU_OK;
  if (resize) {
    np.max_axis = max_axis;
    np.max_samples = max_samples;
    np.score_blocks = score_blocks;
    if (!rc) rc = f->alloc(&n_axis, (size_t)nI * 3 * max_axis);
    if (!rc) rc = f->alloc(&n_part_cost, (size_t)nI * score_blocks);
    if (!rc) rc = f->alloc(&n_part_index, (size_t)nI * score_blocks);
    if (!rc && f->desc.keep_sample_costs) rc = f->alloc(&n_sample_cost, (size_t)nI * max_samples);
    if (!rc && f->desc.keep_sample_costs) rc = f->alloc(&n_sample_status, (size_t)nI * max_samples);
  }
  if (!rc) {
    planWindow(f, np, f->h_fp_n, f->h_fp_spec, f->fp_radius);
    rc = allocPrep(f, np, &n_prep, &n_stride);
  }
  if (!rc && waitStream(f->stream) != hipSuccess) rc = NAVGPU_ERR_HIP;  // (also the allocations' memsets; kernels queued before took their PlannerDev by value)
  if (rc) {
    dropNew();
    return rc;
  }
  // ---- commit: nothing below can fail before the new state is complete
  if (resize) {
    f->release(pl.axis_samples);
    f->release(pl.part_cost);
    f->release(pl.part_index);
    f->release(pl.sample_cost);
    f->release(pl.sample_status);
    np.axis_samples = n_axis;
    np.part_cost = n_part_cost;
    np.part_index = n_part_index;
    np.sample_cost = n_sample_cost;
    np.sample_status = n_sample_status;
  }
  if (n_prep) {
    f->release(pl.prep);
    np.prep = n_prep;
    np.prep_stride = n_stride;
  }
  pl = np;
  f->planner_configured = true;
  f->touchInputs(0, nI);  // allow_unknown decides which cells a wavefront may enter
  // new limits, new boxes: the staged reach follows from the configuration (a failure here leaves a complete, consistent new
  // configuration whose boxes are the old ones: searches that are larger or smaller than they need be, never wrong)
  if ((rc = restageReach(f)) != NAVGPU_OK) return rc;
  HIP_TRY(waitStream(f->stream));
  return NAVGPU_OK;
}

int navgpu_planner_set_plan(navgpu_fleet* f, uint32_t first, uint32_t count) {
  if (!f || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  HIP_TRY(hipMemsetAsync(f->pl.osc_flags + first, 0, sizeof(uint32_t) * count, f->stream));  // resetOscillationFlags
  return NAVGPU_OK;
}

int navgpu_planner_stage(navgpu_fleet* f, uint32_t first, uint32_t count, const navgpu_robot_state* states, const double* plan_xy,
                         uint32_t n_plan_total) {
  if (!f || !states || !plan_xy || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  if (!f->planner_configured) return NAVGPU_ERR_STATE;
  PlannerDev& pl = f->pl;
  const navgpu_dwa_config& c = pl.cfg;
  for (uint32_t li = 0; li < count; ++li) {
    const navgpu_robot_state& s = states[li];
    if (s.plan_count == 0) return NAVGPU_ERR_INVALID;  // the ROS wrapper rejects empty plans before this point
    if (s.plan_count > pl.max_plan) return NAVGPU_ERR_CAPACITY;
    if ((uint64_t)s.plan_first + s.plan_count > n_plan_total) return NAVGPU_ERR_INVALID;
  }
  HIP_TRY(f->waitMirrors(f->ev_pl_h2d, f->ev_pl_set));  // the pinned mirrors may still feed an earlier copy
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
    f->hp_reach[i] = bfsReachCells(f, s, gx, gy);
  }
  f->touchInputs(first, count);
  if (first == 0 && count == f->desc.n_instances) {  // the whole fleet: one copy of the block
    HIP_TRY(hipMemcpyAsync(f->pl_stage_dev, f->pl_stage_host, f->pl_stage_bytes, hipMemcpyHostToDevice, f->stream));
  } else {
    HIP_TRY(hipMemcpyAsync(pl.state + first, f->hp_state + first, sizeof(navgpu_robot_state) * count, hipMemcpyHostToDevice, f->stream));
    HIP_TRY(hipMemcpyAsync(pl.plan + (size_t)first * pl.max_plan * 2, f->hp_plan + (size_t)first * pl.max_plan * 2, sizeof(double) * 2 * (size_t)count * pl.max_plan, hipMemcpyHostToDevice, f->stream));
    HIP_TRY(hipMemcpyAsync(pl.plan_count + first, f->hp_plan_cnt + first, sizeof(uint32_t) * count, hipMemcpyHostToDevice, f->stream));
    HIP_TRY(hipMemcpyAsync(pl.front_last + (size_t)first * 2, f->hp_front + (size_t)first * 2, sizeof(double) * 2 * count, hipMemcpyHostToDevice, f->stream));
    HIP_TRY(hipMemcpyAsync(pl.align_on + first, f->hp_align + first, sizeof(int32_t) * count, hipMemcpyHostToDevice, f->stream));
    HIP_TRY(hipMemcpyAsync(pl.bfs_reach + first, f->hp_reach + first, sizeof(uint32_t) * count, hipMemcpyHostToDevice, f->stream));
  }
  if (f->cycles_in_flight > 1) {
    HIP_TRY(hipEventRecord(f->ev_pl_h2d, f->stream));
    f->ev_pl_set = true;
  }
  f->planner_staged = true;
  f->hp_dma_pending = true;
  return NAVGPU_OK;
}

// A control cycle whose plan has not changed (the plan arrives at ~1 Hz, the pose at controller_frequency): only pose and
// velocity are staged, 24 B per robot; the nose goal, the alignment switch and the wavefront box are re-derived from the
// resident plan exactly as navgpu_planner_stage derives them.  The values travel as kernel arguments (k_stage_poses): the call
// never waits for the stream.
int navgpu_planner_stage_poses(navgpu_fleet* f, uint32_t first, uint32_t count, const float* pos_xyth, const float* vel_xyth) {
  if (!f || !pos_xyth || !vel_xyth || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  if (!f->planner_configured || !f->planner_staged) return NAVGPU_ERR_STATE;
  PlannerDev& pl = f->pl;
  const navgpu_dwa_config& c = pl.cfg;
  for (uint32_t i = first; i < first + count; ++i)
    if (!f->hp_plan_cnt[i]) {
      g_last_error = "navgpu_planner_stage_poses: instance " + std::to_string(i) + " has no staged plan";
      return NAVGPU_ERR_STATE;
    }
  if (f->hp_dma_pending) {  // the host mirrors written below may still feed a full stage's copies
    HIP_TRY(f->waitMirrors(f->ev_pl_h2d, f->ev_pl_set));
    f->hp_dma_pending = false;
  }
  for (uint32_t li = 0; li < count; ++li) {
    const uint32_t i = first + li;
    navgpu_robot_state& s = f->hp_state[i];
    for (int k = 0; k < 3; ++k) {
      s.pos[k] = pos_xyth[3 * li + k];
      s.vel[k] = vel_xyth[3 * li + k];
    }
    const double* last = &f->hp_plan[((size_t)i * pl.max_plan + f->hp_plan_cnt[i] - 1) * 2];
    const double gx = last[0], gy = last[1];
    // DWAPlanner::updatePlanAndLocalCosts (dwa_planner.cpp:254-285), as in navgpu_planner_stage
    const double sq_dist = (s.pos[0] - gx) * (s.pos[0] - gx) + (s.pos[1] - gy) * (s.pos[1] - gy);
    const double angle_to_goal = atan2(gy - s.pos[1], gx - s.pos[0]);
    f->hp_front[2 * i] = gx + c.forward_point_distance * cos(angle_to_goal);
    f->hp_front[2 * i + 1] = gy + c.forward_point_distance * sin(angle_to_goal);
    f->hp_align[i] = sq_dist > c.forward_point_distance * c.forward_point_distance * c.cheat_factor ? 1 : 0;
    f->hp_reach[i] = bfsReachCells(f, s, gx, gy);
  }
  f->touchInputs(first, count);
  PoseChunk ch;
  ch.state = pl.state;
  ch.front_last = pl.front_last;
  ch.align_on = pl.align_on;
  ch.bfs_reach = pl.bfs_reach;
  for (uint32_t at = 0; at < count; at += kPoseChunk) {
    ch.first = first + at;
    ch.count = std::min(kPoseChunk, count - at);
    memcpy(ch.st, f->hp_state + ch.first, sizeof(navgpu_robot_state) * ch.count);
    memcpy(ch.front, f->hp_front + (size_t)ch.first * 2, sizeof(double) * 2 * ch.count);
    memcpy(ch.align, f->hp_align + ch.first, sizeof(int32_t) * ch.count);
    memcpy(ch.reach, f->hp_reach + ch.first, sizeof(uint32_t) * ch.count);
    launch_stage_poses(ch, f->stream);
  }
  return checkLaunch();
}

// the cell box (x0, x1, y0, y1, inclusive) the last cycle's bounded wavefronts had to settle for each instance;
// {0, nx-1, 0, ny-1} for a robot whose grids were searched whole
int navgpu_planner_wavefront_boxes(navgpu_fleet* f, uint32_t first, uint32_t count, int32_t* boxes) {
  if (!f || !boxes || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  for (uint32_t li = 0; li < count; ++li) {
    const uint32_t i = first + li;
    if (f->grid_partial[i])
      memcpy(boxes + 4 * li, &f->h_box[(size_t)4 * i], sizeof(int32_t) * 4);
    else {
      boxes[4 * li] = 0;
      boxes[4 * li + 1] = (int32_t)f->pl.nx - 1;
      boxes[4 * li + 2] = 0;
      boxes[4 * li + 3] = (int32_t)f->pl.ny - 1;
    }
  }
  return NAVGPU_OK;
}

#ifdef NAVGPU_SCORE_STATS  // experiment builds only: the LDS image k_score_prep* stored for one robot
extern "C" int navgpu_debug_prep_image(navgpu_fleet* f, uint32_t inst, uint8_t* out, uint32_t cap, uint32_t* win, uint32_t* bytes) {
  waitStream(f->stream);
  const uint32_t n = std::min<uint32_t>(cap, f->pl.prep_stride);
  hipMemcpy(out, f->pl.prep + (size_t)inst * f->pl.prep_stride, n, hipMemcpyDeviceToHost);
  *win = f->pl.win;
  *bytes = n;
  return 0;
}
#endif
int navgpu_planner_cycle(navgpu_fleet* f, uint32_t first, uint32_t count) {
  if (!f || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  if (!f->planner_configured || !f->planner_staged) return NAVGPU_ERR_STATE;
  PlannerDev& pl = f->pl;
  const int prev_slot = f->res_slot;
  if (f->cycles_in_flight > 1) {  // the cycle before this one may still run: its results stay where they are
    if (first != 0 || count != f->desc.n_instances) return NAVGPU_ERR_STATE;
    pl.result = f->hp_result + (size_t)(prev_slot ^ 1) * f->desc.n_instances;  // (the slot flips once the launches are out)
  }
  pl.bfs_bounded = 1;  // per robot: bfs_reach (0 = whole grid)
  for (uint32_t i = first; i < first + count; ++i) {
    f->grid_partial[i] = robotBox(f, i, f->hp_state[i].pos, f->hp_reach[i], &f->h_box[(size_t)4 * i]) ? 1 : 0;
    f->cycle_gen[i] = f->inputs_gen[i];
  }
  launch_samples(pl, first, count, f->stream);
#ifdef NAVGPU_DEBUG_SWITCHES  // tool builds only (tools/trace_bfs.py): per-item phase stamps of the wavefront launch, written to the named file
  if (getenv("NAVGPU_DEBUG_BFS_TRACE") && !pl.bfs_trace) f->alloc(&pl.bfs_trace, (size_t)f->desc.n_instances * 3 * 8);
#endif
  PROFILED(f, NAVGPU_K_BFS, launch_bfs(pl, first, count, f->stream, pl.bfs_order + (size_t)first * 3, true));
#ifdef NAVGPU_DEBUG_SWITCHES
  if (pl.bfs_trace) {
    std::vector<unsigned long long> h((size_t)count * 24);
    hipMemcpyAsync(h.data(), pl.bfs_trace, h.size() * 8, hipMemcpyDeviceToHost, f->stream);
    waitStream(f->stream);
    if (FILE* fp = fopen(getenv("NAVGPU_DEBUG_BFS_TRACE"), "w")) {
      for (size_t i = 0; i < (size_t)count * 3; ++i)
        fprintf(fp, "%zu %llu %llu %llu %llu %llu %llu %llu %llu\n", i, h[8 * i], h[8 * i + 1] & 0xFFFFFFFFFFFFull, h[8 * i + 1] >> 48, h[8 * i + 2], h[8 * i + 3], h[8 * i + 4], h[8 * i + 5], h[8 * i + 6]);
      fclose(fp);
    }
  }
#endif
  uint32_t n_blocks = 0;
  PROFILED(f, NAVGPU_K_SCORE, n_blocks = launch_score(pl, first, count, nullptr, f->stream));
  PROFILED(f, NAVGPU_K_SELECT, launch_select(pl, first, count, n_blocks, f->stream));
  int rc = checkLaunch();
  if (f->cycles_in_flight > 1) {
    const int slot = prev_slot ^ 1;
    if (rc == NAVGPU_OK && hipEventRecord(f->ev_cycle[slot], f->stream) != hipSuccess) rc = NAVGPU_ERR_HIP;
    if (rc == NAVGPU_OK) {  // only a cycle whose launches and marker are out becomes "the latest"
      f->res_slot = slot;
      f->ev_cycle_set[slot] = true;
    } else {
      pl.result = f->hp_result + (size_t)prev_slot * f->desc.n_instances;
    }
  }
  return rc;
}

int navgpu_planner_set_cycles_in_flight(navgpu_fleet* f, int32_t cycles) {
  if (!f || cycles < 1 || cycles > 2) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  HIP_TRY(waitStream(f->stream));  // nothing in flight across the switch
  if (cycles > 1)
    for (hipEvent_t* e : {&f->ev_cycle[0], &f->ev_cycle[1], &f->ev_cm_h2d, &f->ev_pl_h2d})
      if (!*e) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
  if (cycles == 1 && f->res_slot) {  // back to slot 0, the latest results with it
    memcpy(f->hp_result, f->hp_result + f->desc.n_instances, sizeof(navgpu_plan_result) * f->desc.n_instances);
    f->res_slot = 0;
    f->pl.result = f->hp_result;
  }
  f->ev_cycle_set[0] = f->ev_cycle_set[1] = f->ev_cm_set = f->ev_pl_set = false;
  f->cycles_in_flight = cycles;
  return NAVGPU_OK;
}

int navgpu_planner_results_previous(navgpu_fleet* f, uint32_t first, uint32_t count, navgpu_plan_result* results) {
  if (!f || !results || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  const int slot = f->res_slot ^ 1;
  if (f->cycles_in_flight < 2 || !f->ev_cycle_set[slot]) return NAVGPU_ERR_STATE;  // no cycle before the latest one
  HIP_TRY(hipEventSynchronize(f->ev_cycle[slot]));
  memcpy(results, f->hp_result + (size_t)slot * f->desc.n_instances + first, sizeof(navgpu_plan_result) * count);
  return NAVGPU_OK;
}

int navgpu_planner_results(navgpu_fleet* f, uint32_t first, uint32_t count, navgpu_plan_result* results) {
  if (!f || !results || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  // zero-copy: the results already sit in pinned host memory once the stream has drained (an explicit
  // D2H copy was measured at ~4 ms per call when another HIP user, e.g. PyTorch, shares the process)
  HIP_TRY(waitStream(f->stream));
  memcpy(results, f->hp_result + (size_t)f->res_slot * f->desc.n_instances + first, sizeof(navgpu_plan_result) * count);
  return NAVGPU_OK;
}

int navgpu_planner_trajectory(navgpu_fleet* f, uint32_t instance, double* xyth, uint32_t cap) {
  if (!f || !xyth || instance >= f->desc.n_instances) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  HIP_TRY(waitStream(f->stream));
  const navgpu_plan_result r = f->hp_result[(size_t)f->res_slot * f->desc.n_instances + instance];
  uint32_t n = std::min<uint32_t>(r.n_points > 0 ? r.n_points : 0, cap);
  if (n) {
    HIP_TRY(hipMemcpyAsync(xyth, f->pl.traj + (size_t)instance * f->pl.max_sim_steps * 3, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, f->stream));
    HIP_TRY(waitStream(f->stream));
  }
  return r.n_points;
}

int navgpu_planner_samples(navgpu_fleet* f, uint32_t instance, double* costs, int32_t* status, float* vel, uint32_t cap) {
  if (!f || instance >= f->desc.n_instances) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
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
  FleetGuard guard_(f);
  if (!f->planner_configured || !f->planner_staged) return NAVGPU_ERR_STATE;
  PlannerDev& pl = f->pl;
  if (f->grid_partial[instance]) {  // the sample must stay inside the box the last wavefronts settled
    const navgpu_robot_state& st = f->hp_state[instance];
    const uint32_t need = (uint32_t)std::min(ceil(reachMetres(pl.cfg, st.vel, vs) / pl.res) + 3.0, 32768.0);
    int32_t nb[4];
    const int32_t* hb = &f->h_box[(size_t)4 * instance];
    // (a sideways-shifted look-up - navgpu_planner_set_map_grid_options after the cycle - leaves the box reachMetres
    // describes: those always get complete grids)
    const bool inside = !pl.mg_generic && robotBox(f, instance, st.pos, need, nb) && nb[0] >= hb[0] && nb[1] <= hb[1] && nb[2] >= hb[2] && nb[3] <= hb[3];
    if (!inside) {
      int rc = ensureCompleteGrids(f, instance, 1);
      if (rc) return rc;
    }
  }
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

int navgpu_planner_cost_cloud(navgpu_fleet* f, uint32_t instance, float* points, uint32_t capacity) {
  if (!f || instance >= f->desc.n_instances || (capacity && !points)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  if (!f->planner_configured) return NAVGPU_ERR_STATE;
  {
    int rc = ensureCompleteGrids(f, instance, 1);
    if (rc) return rc;
  }
  const PlannerDev& pl = f->pl;
  if (!f->d_cell_costs) {
    int rc = f->alloc(&f->d_cell_costs, (size_t)pl.cells);
    if (rc) return rc;
  }
  launch_cell_costs(pl, instance, f->d_cell_costs, f->stream);
  std::vector<float4> h(pl.cells);
  HIP_TRY(hipMemcpyAsync(h.data(), f->d_cell_costs, sizeof(float4) * pl.cells, hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(waitStream(f->stream));
  int rc = checkLaunch();
  if (rc) return rc;
  // MapGridVisualizer::publishCostCloud (map_grid_visualizer.cpp:55-83): cx outer, cy inner, mapToWorld per cell
  const double ox = f->h_origin[2 * instance], oy = f->h_origin[2 * instance + 1], res = pl.res;
  uint32_t n = 0;
  for (uint32_t cx = 0; cx < pl.nx; ++cx)
    for (uint32_t cy = 0; cy < pl.ny; ++cy) {
      const float4& c = h[(size_t)cy * pl.nx + cx];
      if (c.w != c.w) continue;  // NaN: getCellCosts returned false
      if (n < capacity) {
        float* p = points + (size_t)7 * n;
        p[0] = (float)(ox + (cx + 0.5) * res);
        p[1] = (float)(oy + (cy + 0.5) * res);
        p[2] = 0.0f;
        p[3] = c.x;
        p[4] = c.y;
        p[5] = c.z;
        p[6] = c.w;
      }
      ++n;
    }
  return (int)n;  // points of the cloud (may exceed capacity: call again with a larger buffer)
}

int navgpu_planner_get_oscillation(navgpu_fleet* f, uint32_t first, uint32_t count, uint32_t* flags, float* prev) {
  if (!f || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  if (flags) HIP_TRY(hipMemcpyAsync(flags, f->pl.osc_flags + first, sizeof(uint32_t) * count, hipMemcpyDeviceToHost, f->stream));
  if (prev) HIP_TRY(hipMemcpyAsync(prev, f->pl.osc_prev + (size_t)first * 3, sizeof(float) * 3 * count, hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(waitStream(f->stream));
  return NAVGPU_OK;
}
int navgpu_planner_set_oscillation(navgpu_fleet* f, uint32_t first, uint32_t count, const uint32_t* flags, const float* prev) {
  if (!f || !f->rangeOk(first, count)) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  if (flags) HIP_TRY(hipMemcpyAsync(f->pl.osc_flags + first, flags, sizeof(uint32_t) * count, hipMemcpyHostToDevice, f->stream));
  if (prev) HIP_TRY(hipMemcpyAsync(f->pl.osc_prev + (size_t)first * 3, prev, sizeof(float) * 3 * count, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(waitStream(f->stream));
  return NAVGPU_OK;
}

// ------------------------------------------------------------------------------------------------ measurement
int navgpu_profile_enable(navgpu_fleet* f, int32_t enable) {
  if (!f) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  if (!enable && f->profiling) {
    int rc = f->foldEvents();
    if (rc) return rc;
  }
  f->profiling = enable != 0;
  return NAVGPU_OK;
}
int navgpu_profile_select(navgpu_fleet* f, uint32_t kernel_mask) {
  if (!f) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
  f->prof_mask = kernel_mask;
  return NAVGPU_OK;
}
int navgpu_profile_reset(navgpu_fleet* f) {
  if (!f) return NAVGPU_ERR_INVALID;
  FleetGuard guard_(f);
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
  FleetGuard guard_(f);
  int rc = f->foldEvents();
  if (rc) return rc;
  if (total_ms) *total_ms = f->prof_ms[k];
  if (launches) *launches = f->prof_n[k];
  return NAVGPU_OK;
}

}  // extern "C"
