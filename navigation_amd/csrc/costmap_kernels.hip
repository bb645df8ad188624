// HIP kernels (gfx950) for the costmap half of the hot path:
//   k_obstacle : ObstacleLayer/VoxelLayer::updateBounds  (raytrace clearing, marking, bounds) +
//                InflationLayer/StaticLayer::updateBounds + LayeredCostmap box + footprint clearing
//   k_merge    : Costmap2D::resetMap window + StaticLayer/ObstacleLayer::updateCosts merges
//   k_inflate  : InflationLayer::updateCosts as an order-independent windowed exact EDT
// Reference citations are on each kernel.  Compiled with -ffp-contract=off: every fp64 expression
// must round exactly like the reference's x86-64 build.
#include "navgpu_device.h"

namespace navgpu {

// ------------------------------------------------------------------------------------------------
// small fills / static map interpretation
// ------------------------------------------------------------------------------------------------
__global__ void k_fill_u8(uint8_t* dst, uint8_t v, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  uint32_t w = v * 0x01010101u;
  uint4 q = make_uint4(w, w, w, w);
  size_t n16 = n / 16;
  for (size_t k = i; k < n16; k += stride) reinterpret_cast<uint4*>(dst)[k] = q;
  for (size_t k = n16 * 16 + i; k < n; k += stride) dst[k] = v;
}
__global__ void k_fill_u32(uint32_t* dst, uint32_t v, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = i; k < n; k += stride) dst[k] = v;
}
void launch_fill_u8(uint8_t* dst, uint8_t v, size_t n, hipStream_t s) {
  if (!n) return;
  size_t blocks = (n / 16 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_fill_u8, dim3((unsigned)blocks), dim3(256), 0, s, dst, v, n);
}
void launch_fill_u32(uint32_t* dst, uint32_t v, size_t n, hipStream_t s) {
  if (!n) return;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_fill_u32, dim3((unsigned)blocks), dim3(256), 0, s, dst, v, n);
}

// StaticLayer::interpretValue (plugins/static_layer.cpp:149-163) applied to an OccupancyGrid and
// broadcast to `count` instances (incomingMap :210-220).
__global__ void k_static_interpret(uint8_t* dst, const int8_t* occ, uint32_t cells, uint32_t cells_padded, uint32_t count,
                                   int track_unknown_space, int trinary, int lethal_threshold, int unknown_cost_value) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cells) return;
  uint8_t value = (uint8_t)occ[i];
  uint8_t lethal_thr = (uint8_t)lethal_threshold;
  uint8_t unknown = (uint8_t)unknown_cost_value;
  uint8_t out;
  if (track_unknown_space && value == unknown)
    out = kNoInfo;
  else if (!track_unknown_space && value == unknown)
    out = kFree;
  else if (value >= lethal_thr)
    out = kLethal;
  else if (trinary)
    out = kFree;
  else {
    double scale = (double)value / lethal_thr;
    out = (uint8_t)(scale * kLethal);
  }
  for (uint32_t k = 0; k < count; ++k) dst[(size_t)k * cells_padded + i] = out;
}
void launch_static_interpret(uint8_t* dst, const int8_t* occ, uint32_t cells, uint32_t cells_padded, uint32_t count,
                             int track_unknown_space, int trinary, int lethal_threshold, int unknown_cost_value, hipStream_t s) {
  hipLaunchKernelGGL(k_static_interpret, dim3((cells + 255) / 256), dim3(256), 0, s, dst, occ, cells, cells_padded, count,
                     track_unknown_space, trinary, lethal_threshold, unknown_cost_value);
}

// ------------------------------------------------------------------------------------------------
// k_export_window: Costmap2DPublisher's occupancy view of a window of the master grid
// (costmap_2d_publisher.cpp:57-74 table, :103-115 full grid, :146-156 update window): 0 -> 0, 253 -> 99,
// 254 -> 100, 255 -> -1, 1..252 -> 1 + 97 (v - 1) / 251.
// ------------------------------------------------------------------------------------------------
__global__ void k_export_window(const uint8_t* master, uint32_t nx, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, int8_t* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w * h) return;
  const uint32_t y = i / w, x = i - y * w;
  const uint32_t v = master[(size_t)(y0 + y) * nx + x0 + x];
  int8_t o;
  if (v == 0) o = 0;
  else if (v == 253) o = 99;
  else if (v == 254) o = 100;
  else if (v == 255) o = -1;
  else o = (int8_t)(1 + (97 * ((int)v - 1)) / 251);
  out[i] = o;
}
void launch_export_window(const uint8_t* master, uint32_t nx, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, int8_t* out, hipStream_t s) {
  hipLaunchKernelGGL(k_export_window, dim3((w * h + 255) / 256), dim3(256), 0, s, master, nx, x0, y0, w, h, out);
}

// ------------------------------------------------------------------------------------------------
// Costmap2D::updateOrigin (costmap_2d.cpp:264-313) / VoxelLayer::updateOrigin (voxel_layer.cpp:385-440):
// the overlap of the old and the new window keeps its contents, everything else becomes the default
// value.  dst(x, y) = src(x + cell_ox, y + cell_oy) when that lies in the grid.  Ping-pong buffers,
// the host swaps the pointers afterwards.
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_shift(const T* src, T* dst, CostmapDev cm, uint32_t first, T fill) {
  const uint32_t inst = first + blockIdx.y;
  const int ox = cm.shift[2 * inst], oy = cm.shift[2 * inst + 1];
  const T* s = src + (size_t)inst * cm.cells_padded;
  T* d = dst + (size_t)inst * cm.cells_padded;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < cm.cells; i += gridDim.x * blockDim.x) {
    const int y = i / cm.nx, x = i - y * cm.nx;
    const int sx = x + ox, sy = y + oy;
    d[i] = (sx >= 0 && sy >= 0 && sx < (int)cm.nx && sy < (int)cm.ny) ? s[sy * cm.nx + sx] : fill;
  }
}
void launch_shift_u8(const uint8_t* src, uint8_t* dst, const CostmapDev& cm, uint32_t first, uint32_t count, uint8_t fill, hipStream_t s) {
  dim3 grid(min((cm.cells + 255) / 256, 64u), count);
  hipLaunchKernelGGL(k_shift<uint8_t>, grid, dim3(256), 0, s, src, dst, cm, first, fill);
}
void launch_shift_u32(const uint32_t* src, uint32_t* dst, const CostmapDev& cm, uint32_t first, uint32_t count, uint32_t fill, hipStream_t s) {
  dim3 grid(min((cm.cells + 255) / 256, 64u), count);
  hipLaunchKernelGGL(k_shift<uint32_t>, grid, dim3(256), 0, s, src, dst, cm, first, fill);
}

// ------------------------------------------------------------------------------------------------
// Bresenham walkers
// ------------------------------------------------------------------------------------------------
// Costmap2D::raytraceLine + bresenham2D (costmap_2d.h:359-417): `at(offset)` on every visited cell
template <class F>
__device__ __forceinline__ void raytrace2d(uint32_t nx, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t max_length,
                                           F&& at) {
  int dx = (int)x1 - (int)x0, dy = (int)y1 - (int)y0;
  uint32_t abs_dx = dx < 0 ? -dx : dx, abs_dy = dy < 0 ? -dy : dy;
  int offset_dx = dx > 0 ? 1 : -1;  // sign(0) == -1 (costmap_2d.h:414-417)
  int offset_dy = (dy > 0 ? 1 : -1) * (int)nx;
  uint32_t offset = y0 * nx + x0;
  double dist = sqrt((double)dx * (double)dx + (double)dy * (double)dy);  // exact integers: == hypot(dx, dy)
  double scale = (dist == 0.0) ? 1.0 : fmin(1.0, (double)max_length / dist);
  uint32_t abs_da, abs_db, lim;
  int offset_a, offset_b;
  if (abs_dx >= abs_dy) {
    abs_da = abs_dx;
    abs_db = abs_dy;
    offset_a = offset_dx;
    offset_b = offset_dy;
  } else {
    abs_da = abs_dy;
    abs_db = abs_dx;
    offset_a = offset_dy;
    offset_b = offset_dx;
  }
  lim = (uint32_t)(scale * abs_da);
  int error_b = abs_da / 2;
  uint32_t end = lim < abs_da ? lim : abs_da;
  for (uint32_t i = 0; i < end; ++i) {
    at(offset);
    offset += offset_a;
    error_b += abs_db;
    if ((uint32_t)error_b >= abs_da) {
      offset += offset_b;
      error_b -= abs_da;
    }
  }
  at(offset);
}

// VoxelGrid::raytraceLine + bresenham3D (voxel_grid.h:226-308): `at(offset, z_mask)`
template <class F>
__device__ __forceinline__ void raytrace3d(uint32_t nx, double x0, double y0, double z0, double x1, double y1, double z1,
                                           uint32_t max_length, F&& at) {
  int dx = int(x1) - int(x0), dy = int(y1) - int(y0), dz = int(z1) - int(z0);
  uint32_t abs_dx = dx < 0 ? -dx : dx, abs_dy = dy < 0 ? -dy : dy, abs_dz = dz < 0 ? -dz : dz;
  int offset_dx = dx > 0 ? 1 : -1;
  int offset_dy = (dy > 0 ? 1 : -1) * (int)nx;
  int offset_dz = dz > 0 ? 1 : -1;
  uint32_t z_mask = ((1u << 16) | 1u) << (uint32_t)z0;
  uint32_t offset = (uint32_t)y0 * nx + (uint32_t)x0;
  double dist = sqrt((x0 - x1) * (x0 - x1) + (y0 - y1) * (y0 - y1) + (z0 - z1) * (z0 - z1));
  double scale = fmin(1.0, (double)max_length / dist);
  // axis roles: 0 = grid offset, 1 = z mask
  uint32_t abs_da, abs_db, abs_dc;
  int off_a, off_b, off_c, role_a, role_b, role_c;
  uint32_t mmax = abs_dy > abs_dz ? abs_dy : abs_dz;
  if (abs_dx >= mmax) {
    abs_da = abs_dx; abs_db = abs_dy; abs_dc = abs_dz;
    off_a = offset_dx; off_b = offset_dy; off_c = offset_dz;
    role_a = 0; role_b = 0; role_c = 1;
  } else if (abs_dy >= abs_dz) {
    abs_da = abs_dy; abs_db = abs_dx; abs_dc = abs_dz;
    off_a = offset_dy; off_b = offset_dx; off_c = offset_dz;
    role_a = 0; role_b = 0; role_c = 1;
  } else {
    abs_da = abs_dz; abs_db = abs_dx; abs_dc = abs_dy;
    off_a = offset_dz; off_b = offset_dx; off_c = offset_dy;
    role_a = 1; role_b = 0; role_c = 0;
  }
  auto step = [&](int role, int v) {
    if (role == 0)
      offset += v;
    else {
      if (v > 0)
        z_mask <<= 1;
      else
        z_mask >>= 1;
    }
  };
  uint32_t lim = (uint32_t)(scale * abs_da);
  uint32_t end = lim < abs_da ? lim : abs_da;
  int error_b = abs_da / 2, error_c = abs_da / 2;
  for (uint32_t i = 0; i < end; ++i) {
    at(offset, z_mask);
    step(role_a, off_a);
    error_b += abs_db;
    error_c += abs_dc;
    if ((uint32_t)error_b >= abs_da) {
      step(role_b, off_b);
      error_b -= abs_da;
    }
    if ((uint32_t)error_c >= abs_da) {
      step(role_c, off_c);
      error_c -= abs_da;
    }
  }
  at(offset, z_mask);
}

__device__ __forceinline__ uint32_t cellDistance(double world_dist, double res) {  // costmap_2d.cpp:181-185
  double c = fmax(0.0, ceil(world_dist / res));
  return (uint32_t)c;
}
// VoxelGrid::bitsBelowThreshold (voxel_grid.h:151-164) == popcount(n) <= thr
__device__ __forceinline__ bool bitsBelowThreshold(uint32_t n, uint32_t thr) { return (uint32_t)__popc(n) <= thr; }

struct BoundsAcc {
  double min_x, min_y, max_x, max_y;
  __device__ __forceinline__ void touch(double x, double y) {  // CostmapLayer::touch: std::min(x, *min_x) ...
    min_x = (min_x < x) ? min_x : x;
    min_y = (min_y < y) ? min_y : y;
    max_x = (max_x < x) ? x : max_x;
    max_y = (max_y < y) ? y : max_y;
  }
};

__device__ __forceinline__ void enforceBounds(const Geom& g, double wx, double wy, int& mx, int& my) {  // costmap_2d.cpp:228-262
  if (wx < g.ox)
    mx = 0;
  else if (wx > g.res * (g.nx - 1) + g.ox)
    mx = g.nx - 1;
  else
    mx = (int)((wx - g.ox) / g.res);
  if (wy < g.oy)
    my = 0;
  else if (wy > g.res * (g.ny - 1) + g.oy)
    my = g.ny - 1;
  else
    my = (int)((wy - g.oy) / g.res);
}

// ------------------------------------------------------------------------------------------------
// k_obstacle: one 256-thread workgroup per robot instance.
//   ObstacleLayer::updateBounds   plugins/obstacle_layer.cpp:340-413 (+ raytraceFreespace :498-576,
//                                 updateRaytraceBounds :602-610, updateFootprint :415-425)
//   VoxelLayer::updateBounds      plugins/voxel_layer.cpp:116-213 (+ raytraceFreespace :266-383)
//   StaticLayer::updateBounds     plugins/static_layer.cpp:263-283
//   InflationLayer::updateBounds  plugins/inflation_layer.cpp:125-158
//   LayeredCostmap::updateMap     src/layered_costmap.cpp:96-135 (bounds -> cell box)
//   ObstacleLayer::updateCosts    plugins/obstacle_layer.cpp:432-435 (footprint polygon clearing,
//                                 Costmap2D::setConvexPolygonCost src/costmap_2d.cpp:315-428)
// Ordering kept from the reference: every clearing ray before any mark (workgroup barrier
// between the phases); all rays write FREE_SPACE so write-write races are benign.  The voxel
// clear phase is split in two (column bits with atomics, then the dependent 2-D byte from the
// final column state) which is equivalent because bits only ever get cleared during the phase.
// ------------------------------------------------------------------------------------------------
template <bool VOXEL>
__global__ __launch_bounds__(256) void k_obstacle(CostmapDev cm, uint32_t first, const double* bounds_io_in, double* bounds_io_out,
                                                  int only_bounds) {
  const uint32_t inst = first + blockIdx.x;
  const uint32_t tid = threadIdx.x;
  Geom g{cm.origin[2 * inst], cm.origin[2 * inst + 1], cm.res, cm.nx, cm.ny};
  uint8_t* layer = cm.obst + (size_t)inst * cm.cells_padded;
  uint32_t* vox = VOXEL ? cm.voxel + (size_t)inst * cm.cells_padded : nullptr;
  InstCostmapState* st = cm.state + inst;

  __shared__ double red[4][4];
  __shared__ int s_box_valid;
  __shared__ uint32_t colmin[1024], colmax[1024];
  __shared__ uint32_t s_vx[kMaxFootprint], s_vy[kMaxFootprint];
  __shared__ int s_poly_ok;

  BoundsAcc b{1e30, 1e30, -1e30, -1e30};
  if (tid == 0) {
    if (bounds_io_in) {
      b.min_x = bounds_io_in[4 * blockIdx.x + 0];
      b.min_y = bounds_io_in[4 * blockIdx.x + 1];
      b.max_x = bounds_io_in[4 * blockIdx.x + 2];
      b.max_y = bounds_io_in[4 * blockIdx.x + 3];
    }
    if (!only_bounds && (cm.layers & NAVGPU_LAYER_STATIC) && st->static_has_updated_data) {
      // mapToWorld(x_, y_) and (x_+width_, y_+height_) with the whole map as the updated window
      double wx = g.ox + (0 + 0.5) * g.res, wy = g.oy + (0 + 0.5) * g.res;
      b.min_x = (wx < b.min_x) ? wx : b.min_x;  // std::min(wx, *min_x)
      b.min_y = (wy < b.min_y) ? wy : b.min_y;
      wx = g.ox + (g.nx + 0.5) * g.res;
      wy = g.oy + (g.ny + 0.5) * g.res;
      b.max_x = (b.max_x < wx) ? wx : b.max_x;
      b.max_y = (b.max_y < wy) ? wy : b.max_y;
      st->static_has_updated_data = 0;
    }
    if (!only_bounds && cm.stat_roll) {
      // rolling window: StaticLayer::updateBounds has no early return (static_layer.cpp:265-268) and adds the extent
      // of the LAYER's grid - the static map's own geometry - every cycle
      double wx = cm.stat_ox + (0 + 0.5) * cm.stat_res, wy = cm.stat_oy + (0 + 0.5) * cm.stat_res;
      b.min_x = (wx < b.min_x) ? wx : b.min_x;
      b.min_y = (wy < b.min_y) ? wy : b.min_y;
      wx = cm.stat_ox + (cm.stat_nx + 0.5) * cm.stat_res;
      wy = cm.stat_oy + (cm.stat_ny + 0.5) * cm.stat_res;
      b.max_x = (b.max_x < wx) ? wx : b.max_x;
      b.max_y = (b.max_y < wy) ? wy : b.max_y;
    }
  }

  const bool has_obs_layer = (cm.layers & (NAVGPU_LAYER_OBSTACLE | NAVGPU_LAYER_VOXEL)) && cm.obs_enabled;
  if (tid == 0 && has_obs_layer && st->has_extra_bounds) {  // useExtraBounds (obstacle_layer.cpp:347, voxel_layer.cpp:123; costmap_layer.cpp:46-60)
    b.min_x = (st->extra[0] < b.min_x) ? st->extra[0] : b.min_x;
    b.min_y = (st->extra[1] < b.min_y) ? st->extra[1] : b.min_y;
    b.max_x = (b.max_x < st->extra[2]) ? st->extra[2] : b.max_x;
    b.max_y = (b.max_y < st->extra[3]) ? st->extra[3] : b.max_y;
    st->extra[0] = st->extra[1] = 1e6;
    st->extra[2] = st->extra[3] = -1e6;
    st->has_extra_bounds = 0;
  }
  const uint32_t ob = inst * cm.max_obs, oe = ob + cm.obs_count[inst];
  const float* inst_points = cm.points + (size_t)inst * cm.max_points * 3;
  const uint32_t unknown_thr = VOXEL ? (uint32_t)(cm.unknown_threshold + (16 - cm.z_voxels)) : 0;  // voxel_layer.cpp:89
  const uint32_t mark_thr = VOXEL ? (uint32_t)cm.mark_threshold : 0;
  const uint32_t size_z = VOXEL ? (uint32_t)(cm.z_voxels > 16 ? 16 : cm.z_voxels) : 0;

  if (has_obs_layer) {
    // ---------------- phase 1: clearing observations
    for (int pass = 0; pass < (VOXEL ? 2 : 1); ++pass) {
      for (uint32_t o = ob; o < oe; ++o) {
        const ObsCsr obs = cm.obs[o];
        if (!(obs.flags & NAVGPU_OBS_CLEARING)) continue;
        const float* pts = inst_points + (size_t)obs.first_point * 3;
        if (!VOXEL) {
          uint32_t x0, y0;
          if (!worldToMap(g, obs.ox, obs.oy, x0, y0)) continue;
          const double map_end_x = g.ox + g.nx * g.res, map_end_y = g.oy + g.ny * g.res;
          if (tid == 0) b.touch(obs.ox, obs.oy);
          const uint32_t cell_range = cellDistance(obs.raytrace_range, g.res);
          for (uint32_t p = tid; p < obs.n_points; p += blockDim.x) {
            double wx = pts[3 * p], wy = pts[3 * p + 1];
            const double ox = obs.ox, oy = obs.oy;
            double a = wx - ox, bb = wy - oy;
            if (wx < g.ox) {
              double t = (g.ox - ox) / a;
              wx = g.ox;
              wy = oy + bb * t;
            }
            if (wy < g.oy) {
              double t = (g.oy - oy) / bb;
              wx = ox + a * t;
              wy = g.oy;
            }
            if (wx > map_end_x) {
              double t = (map_end_x - ox) / a;
              wx = map_end_x - .001;
              wy = oy + bb * t;
            }
            if (wy > map_end_y) {
              double t = (map_end_y - oy) / bb;
              wx = ox + a * t;
              wy = map_end_y - .001;
            }
            uint32_t x1, y1;
            if (!worldToMap(g, wx, wy, x1, y1)) continue;
            raytrace2d(g.nx, x0, y0, x1, y1, cell_range, [&](uint32_t off) { layer[off] = kFree; });
            double dx = wx - ox, dy = wy - oy;
            double full = hyp2(dx, dy);
            double scale = fmin(1.0, obs.raytrace_range / full);
            b.touch(ox + dx * scale, oy + dy * scale);
          }
        } else {
          if (obs.n_points == 0) continue;
          const double ox = obs.ox, oy = obs.oy, oz = obs.oz;
          // worldToMap3DFloat (voxel_layer.h:107-118)
          if (ox < g.ox || oy < g.oy || oz < cm.origin_z) continue;
          const double sensor_x = (ox - g.ox) / g.res, sensor_y = (oy - g.oy) / g.res, sensor_z = (oz - cm.origin_z) / cm.z_resolution;
          if (!(sensor_x < g.nx && sensor_y < g.ny && sensor_z < size_z)) continue;
          const double map_end_x = g.ox + (g.nx - 1 + 0.5) * g.res;  // origin + getSizeInMetersX()
          const double map_end_y = g.oy + (g.ny - 1 + 0.5) * g.res;
          const uint32_t cell_range = cellDistance(obs.raytrace_range, g.res);
          for (uint32_t p = tid; p < obs.n_points; p += blockDim.x) {
            double wpx = pts[3 * p], wpy = pts[3 * p + 1], wpz = pts[3 * p + 2];
            double distance = sqrt((ox - wpx) * (ox - wpx) + (oy - wpy) * (oy - wpy) + (oz - wpz) * (oz - wpz));
            double scaling_fact = fmax(fmin(1.0, (distance - 2 * g.res) / distance), 0.0);
            wpx = scaling_fact * (wpx - ox) + ox;
            wpy = scaling_fact * (wpy - oy) + oy;
            wpz = scaling_fact * (wpz - oz) + oz;
            double a = wpx - ox, bb = wpy - oy, c = wpz - oz, t = 1.0;
            if (wpz > cm.max_obstacle_height)
              t = fmax(0.0, fmin(t, (cm.max_obstacle_height - 0.01 - oz) / c));
            else if (wpz < cm.origin_z)
              t = fmin(t, (cm.origin_z - oz) / c);
            if (wpx < g.ox) t = fmin(t, (g.ox - ox) / a);
            if (wpy < g.oy) t = fmin(t, (g.oy - oy) / bb);
            if (wpx > map_end_x) t = fmin(t, (map_end_x - ox) / a);
            if (wpy > map_end_y) t = fmin(t, (map_end_y - oy) / bb);
            wpx = ox + a * t;
            wpy = oy + bb * t;
            wpz = oz + c * t;
            if (wpx < g.ox || wpy < g.oy || wpz < cm.origin_z) continue;
            double px = (wpx - g.ox) / g.res, py = (wpy - g.oy) / g.res, pz = (wpz - cm.origin_z) / cm.z_resolution;
            if (!(px < g.nx && py < g.ny && pz < size_z)) continue;
            // clearVoxelLineInMap endpoint check (voxel_grid.cpp:131-136) is implied by the two tests above
            if (pass == 0) {
              raytrace3d(g.nx, sensor_x, sensor_y, sensor_z, px, py, pz, cell_range,
                         [&](uint32_t off, uint32_t zm) { atomicAnd(&vox[off], ~zm); });
              double dx = wpx - ox, dy = wpy - oy;
              double full = hyp2(dx, dy);
              double scale = fmin(1.0, obs.raytrace_range / full);
              b.touch(ox + dx * scale, oy + dy * scale);
            } else {
              raytrace3d(g.nx, sensor_x, sensor_y, sensor_z, px, py, pz, cell_range, [&](uint32_t off, uint32_t) {
                uint32_t col = vox[off];
                uint32_t unknown_bits = (uint16_t)(col >> 16) ^ (uint16_t)col;
                uint32_t marked_bits = col >> 16;
                if (bitsBelowThreshold(marked_bits, mark_thr))
                  layer[off] = bitsBelowThreshold(unknown_bits, unknown_thr) ? kFree : kNoInfo;
              });
            }
          }
        }
      }
      __syncthreads();
    }
    // ---------------- phase 2: marking observations
    // VoxelGrid::markVoxelInMap reports "marked" from the column's bit count AT THAT POINT of the sequential loop
    // (voxel_grid.h:100-117), so with mark_threshold > 0 which points touch the bounds depends on the point
    // order.  Exact two-step form: (1) every point records its cell, z and the column bits before any marking;
    // (2) a point is marked iff those bits, the bits of the EARLIER points of the same cell and its own exceed
    // the threshold.  (mark_threshold == 0: every accepted point is marked, the one-pass form below is exact.)
    if (VOXEL && mark_thr > 0) {
      uint2* seq = cm.mark_seq + (size_t)inst * cm.max_points;
      uint32_t n_seq = 0;
      for (uint32_t o = ob; o < oe; ++o) n_seq = max(n_seq, cm.obs[o].first_point + cm.obs[o].n_points);
      for (uint32_t s = tid; s < n_seq; s += blockDim.x) seq[s] = make_uint2(0u, 0u);
      __syncthreads();
      for (uint32_t o = ob; o < oe; ++o) {
        const ObsCsr obs = cm.obs[o];
        if (!(obs.flags & NAVGPU_OBS_MARKING)) continue;
        const float* pts = inst_points + (size_t)obs.first_point * 3;
        const double sq_obstacle_range = obs.obstacle_range * obs.obstacle_range;
        for (uint32_t p = tid; p < obs.n_points; p += blockDim.x) {
          const float fx = pts[3 * p], fy = pts[3 * p + 1], fz = pts[3 * p + 2];
          double px = fx, py = fy, pz = fz;
          if (pz > cm.max_obstacle_height) continue;
          double sq_dist = (px - obs.ox) * (px - obs.ox) + (py - obs.oy) * (py - obs.oy) + (pz - obs.oz) * (pz - obs.oz);
          if (sq_dist >= sq_obstacle_range) continue;
          double wz = (pz < cm.origin_z) ? cm.origin_z : pz;
          if (px < g.ox || py < g.oy || wz < cm.origin_z) continue;
          double fxm = (px - g.ox) / g.res, fym = (py - g.oy) / g.res, fzm = (wz - cm.origin_z) / cm.z_resolution;
          if (!(fxm < 2147483648.0) || !(fym < 2147483648.0) || !(fzm < 2147483648.0)) continue;
          uint32_t mx = (uint32_t)(int)fxm, my = (uint32_t)(int)fym, mz = (uint32_t)(int)fzm;
          if (!(mx < g.nx && my < g.ny && mz < (uint32_t)cm.z_voxels)) continue;
          if (mz >= size_z) continue;
          const uint32_t cell = my * g.nx + mx;
          seq[obs.first_point + p] = make_uint2(cell, 0x80000000u | ((vox[cell] >> 16) << 8) | mz);
        }
      }
      __syncthreads();
      for (uint32_t s = tid; s < n_seq; s += blockDim.x) {
        const uint2 e = seq[s];
        if (!(e.y & 0x80000000u)) continue;
        const uint32_t mz = e.y & 0xFFu;
        uint32_t acc = ((e.y >> 8) & 0xFFFFu) | (1u << mz);
        for (uint32_t q = 0; q < s; ++q) {
          const uint2 f = seq[q];
          if ((f.y & 0x80000000u) && f.x == e.x) acc |= 1u << (f.y & 0xFFu);
        }
        atomicOr(&vox[e.x], ((uint32_t)1 << mz << 16) | (1u << mz));
        if (!bitsBelowThreshold(acc, mark_thr)) {
          layer[e.x] = kLethal;
          const float* pt = inst_points + (size_t)s * 3;  // touched with the point's own coordinates (voxel_layer.cpp:181)
          b.touch((double)pt[0], (double)pt[1]);
        }
      }
    } else
    for (uint32_t o = ob; o < oe; ++o) {
      const ObsCsr obs = cm.obs[o];
      if (!(obs.flags & NAVGPU_OBS_MARKING)) continue;
      const float* pts = inst_points + (size_t)obs.first_point * 3;
      const double sq_obstacle_range = obs.obstacle_range * obs.obstacle_range;
      for (uint32_t p = tid; p < obs.n_points; p += blockDim.x) {
        const float fx = pts[3 * p], fy = pts[3 * p + 1], fz = pts[3 * p + 2];
        double px = fx, py = fy, pz = fz;
        if (pz > cm.max_obstacle_height) continue;
        double sq_dist = (px - obs.ox) * (px - obs.ox) + (py - obs.oy) * (py - obs.oy) + (pz - obs.oz) * (pz - obs.oz);
        if (sq_dist >= sq_obstacle_range) continue;
        if (!VOXEL) {
          uint32_t mx, my;
          if (!worldToMap(g, px, py, mx, my)) continue;
          layer[my * g.nx + mx] = kLethal;
          b.touch(px, py);
        } else {
          double wz = (pz < cm.origin_z) ? cm.origin_z : pz;  // voxel_layer.cpp:169-173
          if (px < g.ox || py < g.oy || wz < cm.origin_z) continue;
          double fxm = (px - g.ox) / g.res, fym = (py - g.oy) / g.res, fzm = (wz - cm.origin_z) / cm.z_resolution;
          if (!(fxm < 2147483648.0) || !(fym < 2147483648.0) || !(fzm < 2147483648.0)) continue;
          uint32_t mx = (uint32_t)(int)fxm, my = (uint32_t)(int)fym, mz = (uint32_t)(int)fzm;
          if (!(mx < g.nx && my < g.ny && mz < (uint32_t)cm.z_voxels)) continue;  // worldToMap3D uses size_z_ as configured
          if (mz >= size_z) continue;                                             // VoxelGrid::markVoxelInMap bound (voxel_grid.h:102)
          uint32_t full_mask = ((uint32_t)1 << mz << 16) | (1u << mz);
          uint32_t old = atomicOr(&vox[my * g.nx + mx], full_mask);
          uint32_t marked_bits = (old | full_mask) >> 16;
          if (!bitsBelowThreshold(marked_bits, mark_thr)) {
            layer[my * g.nx + mx] = kLethal;
            b.touch(px, py);
          }
        }
      }
    }
    // ---------------- updateFootprint: touch the transformed footprint
    if (cm.footprint_clearing) {
      uint32_t nfp = cm.fp_n[inst];
      if (tid < nfp) b.touch(cm.fp_world[((size_t)inst * kMaxFootprint + tid) * 2], cm.fp_world[((size_t)inst * kMaxFootprint + tid) * 2 + 1]);
    }
  }

  // ---------------- workgroup reduction of the bounds (min/max are exact, order-free)
  for (int off = 32; off > 0; off >>= 1) {
    double o0 = __shfl_down(b.min_x, off), o1 = __shfl_down(b.min_y, off), o2 = __shfl_down(b.max_x, off), o3 = __shfl_down(b.max_y, off);
    b.min_x = (o0 < b.min_x) ? o0 : b.min_x;
    b.min_y = (o1 < b.min_y) ? o1 : b.min_y;
    b.max_x = (b.max_x < o2) ? o2 : b.max_x;
    b.max_y = (b.max_y < o3) ? o3 : b.max_y;
  }
  if ((tid & 63) == 0) {
    red[tid >> 6][0] = b.min_x;
    red[tid >> 6][1] = b.min_y;
    red[tid >> 6][2] = b.max_x;
    red[tid >> 6][3] = b.max_y;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w) {
      b.min_x = (red[w][0] < b.min_x) ? red[w][0] : b.min_x;
      b.min_y = (red[w][1] < b.min_y) ? red[w][1] : b.min_y;
      b.max_x = (b.max_x < red[w][2]) ? red[w][2] : b.max_x;
      b.max_y = (b.max_y < red[w][3]) ? red[w][3] : b.max_y;
    }
    if (only_bounds) {
      bounds_io_out[4 * blockIdx.x + 0] = b.min_x;
      bounds_io_out[4 * blockIdx.x + 1] = b.min_y;
      bounds_io_out[4 * blockIdx.x + 2] = b.max_x;
      bounds_io_out[4 * blockIdx.x + 3] = b.max_y;
      // layer-granular use (costmap_2d::Layer adapters): the footprint polygon is cleared here, ahead of the
      // navgpu_obstacle_update_costs that the host's LayeredCostmap issues next (ObstacleLayer::updateCosts :431-434;
      // updateFootprint has put the polygon into the bounds, so that call always follows)
      s_box_valid = 1;
    } else {
      if (cm.layers & NAVGPU_LAYER_INFLATION) {  // InflationLayer::updateBounds
        if (st->need_reinflation) {
          st->last_min_x = b.min_x;
          st->last_min_y = b.min_y;
          st->last_max_x = b.max_x;
          st->last_max_y = b.max_y;
          b.min_x = -3.4028234663852886e38;  // -std::numeric_limits<float>::max()
          b.min_y = -3.4028234663852886e38;
          b.max_x = 3.4028234663852886e38;
          b.max_y = 3.4028234663852886e38;
          st->need_reinflation = 0;
        } else {
          double tminx = st->last_min_x, tminy = st->last_min_y, tmaxx = st->last_max_x, tmaxy = st->last_max_y;
          st->last_min_x = b.min_x;
          st->last_min_y = b.min_y;
          st->last_max_x = b.max_x;
          st->last_max_y = b.max_y;
          b.min_x = ((b.min_x < tminx) ? b.min_x : tminx) - cm.inflation_radius;  // std::min(tmp_min_x, *min_x)
          b.min_y = ((b.min_y < tminy) ? b.min_y : tminy) - cm.inflation_radius;
          b.max_x = ((tmaxx < b.max_x) ? b.max_x : tmaxx) + cm.inflation_radius;
          b.max_y = ((tmaxy < b.max_y) ? b.max_y : tmaxy) + cm.inflation_radius;
        }
      }
      st->bounds[0] = b.min_x;
      st->bounds[1] = b.min_y;
      st->bounds[2] = b.max_x;
      st->bounds[3] = b.max_y;
      int x0, xn, y0, yn;
      enforceBounds(g, b.min_x, b.min_y, x0, y0);
      enforceBounds(g, b.max_x, b.max_y, xn, yn);
      x0 = x0 > 0 ? x0 : 0;
      xn = ((int)g.nx < xn + 1) ? (int)g.nx : xn + 1;
      y0 = y0 > 0 ? y0 : 0;
      yn = ((int)g.ny < yn + 1) ? (int)g.ny : yn + 1;
      st->box[0] = x0;
      st->box[1] = xn;
      st->box[2] = y0;
      st->box[3] = yn;
      st->box_valid = !(xn < x0 || yn < y0);
      s_box_valid = st->box_valid;
    }
  }
  __syncthreads();

  // ---------------- ObstacleLayer::updateCosts head: clear the robot footprint polygon in the layer grid.
  // setConvexPolygonCost == per column x, every cell between the lowest and the highest outline
  // cell of that column (the bubble sort + pairwise walk of convexFillCells yields exactly this
  // because each outline column holds >= 2 entries of a closed outline; tests/test_polygon_fill).
  if (has_obs_layer && cm.footprint_clearing && s_box_valid) {
    const uint32_t nfp = cm.fp_n[inst];
    if (tid == 0) s_poly_ok = nfp >= 3;
    __syncthreads();
    if (tid < nfp) {
      uint32_t mx, my;
      bool ok = worldToMap(g, cm.fp_world[((size_t)inst * kMaxFootprint + tid) * 2], cm.fp_world[((size_t)inst * kMaxFootprint + tid) * 2 + 1], mx, my);
      s_vx[tid] = mx;
      s_vy[tid] = my;
      if (!ok) atomicAnd(&s_poly_ok, 0);
    }
    __syncthreads();
    if (s_poly_ok) {
      uint32_t minx = 0xFFFFFFFFu, maxx = 0;
      for (uint32_t i = 0; i < nfp; ++i) {
        minx = s_vx[i] < minx ? s_vx[i] : minx;
        maxx = s_vx[i] > maxx ? s_vx[i] : maxx;
      }
      const uint32_t span = maxx - minx + 1;
      if (span <= 1024) {
        for (uint32_t c = tid; c < span; c += blockDim.x) {
          colmin[c] = 0xFFFFFFFFu;
          colmax[c] = 0;
        }
        __syncthreads();
        if (tid < nfp) {
          uint32_t e1 = (tid + 1 == nfp) ? 0 : tid + 1;
          raytrace2d(g.nx, s_vx[tid], s_vy[tid], s_vx[e1], s_vy[e1], 0xFFFFFFFFu, [&](uint32_t off) {
            uint32_t y = off / g.nx, x = off - y * g.nx;
            atomicMin(&colmin[x - minx], y);
            atomicMax(&colmax[x - minx], y);
          });
        }
        __syncthreads();
        for (uint32_t c = tid >> 4; c < span; c += blockDim.x >> 4) {
          uint32_t lo = colmin[c], hi = colmax[c];
          if (lo == 0xFFFFFFFFu) continue;
          for (uint32_t y = lo + (tid & 15); y <= hi; y += 16) layer[y * g.nx + minx + c] = kFree;
        }
      }
    }
  }
}

// k_reset_bounding_box: CostmapLayer::resetBoundingBox (costmap_layer.cpp:30-43) on the obstacle / voxel layer's 2-D grid:
// worldToMapEnforceBounds of both corners, Costmap2D::resetMap (rows [y0, yn), columns [x0, xn): the end cell is NOT
// included, costmap_2d.cpp:93-99), addExtraBounds for the next updateBounds.  A VoxelLayer's columns stay as they are
// (it overrides resetMaps, not resetMap).
__global__ __launch_bounds__(256) void k_reset_bounding_box(CostmapDev cm, uint32_t first, const double* boxes_world) {
  const uint32_t inst = first + blockIdx.x;
  const Geom g{cm.origin[2 * inst], cm.origin[2 * inst + 1], cm.res, cm.nx, cm.ny};
  const double* bw = boxes_world + 4 * blockIdx.x;
  int x0, y0, xn, yn;
  enforceBounds(g, bw[0], bw[1], x0, y0);
  enforceBounds(g, bw[2], bw[3], xn, yn);
  uint8_t* layer = cm.obst + (size_t)inst * cm.cells_padded;
  if (xn > x0 && yn > y0) {
    const uint32_t w = (uint32_t)(xn - x0), n = w * (uint32_t)(yn - y0);
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) layer[(uint32_t)(y0 + i / w) * g.nx + (uint32_t)x0 + i % w] = cm.obstacle_default;
  }
  if (threadIdx.x == 0) {
    InstCostmapState* st = cm.state + inst;
    if (!st->has_extra_bounds) {  // (a fresh state is all zeros: the reference starts from 1e6 / -1e6)
      st->extra[0] = st->extra[1] = 1e6;
      st->extra[2] = st->extra[3] = -1e6;
    }
    st->extra[0] = (bw[0] < st->extra[0]) ? bw[0] : st->extra[0];
    st->extra[1] = (bw[1] < st->extra[1]) ? bw[1] : st->extra[1];
    st->extra[2] = (st->extra[2] < bw[2]) ? bw[2] : st->extra[2];
    st->extra[3] = (st->extra[3] < bw[3]) ? bw[3] : st->extra[3];
    st->has_extra_bounds = 1;
  }
}
// Costmap2D::resetMap(x0, y0, xn, yn) (costmap_2d.cpp:93-99) of one byte grid per instance
__global__ __launch_bounds__(256) void k_reset_window(uint8_t* grid, size_t stride, uint32_t nx, uint32_t x0, uint32_t y0, uint32_t xn, uint32_t yn, uint8_t value) {
  uint8_t* gp = grid + (size_t)blockIdx.x * stride;
  const uint32_t w = xn - x0, n = w * (yn - y0);
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) gp[(y0 + i / w) * nx + x0 + i % w] = value;
}
void launch_reset_window(uint8_t* grid, size_t stride, uint32_t count, uint32_t nx, uint32_t x0, uint32_t y0, uint32_t xn, uint32_t yn, uint8_t value, hipStream_t s) {
  hipLaunchKernelGGL(k_reset_window, dim3(count), dim3(256), 0, s, grid, stride, nx, x0, y0, xn, yn, value);
}
void launch_reset_bounding_box(const CostmapDev& cm, uint32_t first, uint32_t count, const double* boxes_world, hipStream_t s) {
  hipLaunchKernelGGL(k_reset_bounding_box, dim3(count), dim3(256), 0, s, cm, first, boxes_world);
}

void launch_obstacle(const CostmapDev& cm, uint32_t first, uint32_t count, const double* bounds_in, int only_bounds, hipStream_t s) {
  // bounds_in doubles as the in/out buffer in only_bounds mode
  double* bounds_out = const_cast<double*>(bounds_in);
  if (cm.layers & NAVGPU_LAYER_VOXEL)
    hipLaunchKernelGGL(k_obstacle<true>, dim3(count), dim3(256), 0, s, cm, first, bounds_in, bounds_out, only_bounds);
  else
    hipLaunchKernelGGL(k_obstacle<false>, dim3(count), dim3(256), 0, s, cm, first, bounds_in, bounds_out, only_bounds);
}

// ------------------------------------------------------------------------------------------------
// k_merge: Costmap2D::resetMap (costmap_2d.cpp:93-99) over the box, then StaticLayer::updateCosts
// (static_layer.cpp:285-299: updateWithTrueOverwrite | updateWithMax) and ObstacleLayer::updateCosts
// (obstacle_layer.cpp:437-447: updateWithOverwrite | updateWithMax, costmap_layer.cpp:62-124),
// fused per byte.  16 cells per thread: three 16-byte reads and one 16-byte write, fully coalesced.
// layer_only: ObstacleLayer::updateCosts alone, as a costmap_2d::Layer plugin runs it - the master
// grid it is handed already holds what the layers before it wrote (no reset, no static merge).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_merge(CostmapDev cm, uint32_t first, const int32_t* boxes, int static_received, int layer_only) {
  const uint32_t inst = first + blockIdx.y;
  int x0, xn, y0, yn;
  if (boxes) {
    x0 = boxes[4 * blockIdx.y + 0];
    y0 = boxes[4 * blockIdx.y + 1];
    xn = boxes[4 * blockIdx.y + 2];
    yn = boxes[4 * blockIdx.y + 3];
  } else {
    const InstCostmapState* st = cm.state + inst;
    if (!st->box_valid) return;
    x0 = st->box[0];
    xn = st->box[1];
    y0 = st->box[2];
    yn = st->box[3];
  }
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;  // 16-cell group
  const uint32_t base = q * 16;
  if (base >= cm.cells) return;
  // rows touched by this group
  const int row_first = base / cm.nx, row_last = (min(base + 15, cm.cells - 1)) / cm.nx;
  if (row_last < y0 || row_first >= yn) return;
  bool inside = false;  // all 16 cells in the box: resetMap overwrites every byte, the master's old ones need not be read
  if (row_first == row_last) {  // a group inside one row: nothing to do left or right of the box either (3.4x -> 1.3x the box's bytes)
    const int gx = base - row_first * cm.nx;
    if (gx + 15 < x0 || gx >= xn) return;
    inside = gx >= x0 && gx + 15 < xn && base + 15 < cm.cells && !layer_only;
  }
  const size_t off = (size_t)inst * cm.cells_padded + base;
  uint4 mv = make_uint4(0, 0, 0, 0);
  if (!inside) mv = *reinterpret_cast<const uint4*>(cm.master + off);
  const bool has_static = (cm.layers & NAVGPU_LAYER_STATIC) && static_received && !layer_only && !cm.stat_roll && cm.stat;
  const bool roll_static = cm.stat_roll != nullptr && !layer_only;
  double tfm[8] = {1, 0, 0, 0, 0, 1, 0, 0};  // rows x and y of the transform: basis[0..2], origin.x, basis[3..5], origin.y
  double m_ox = 0, m_oy = 0;
  if (roll_static) {
    const double* T = cm.stat_tf + (size_t)inst * 12;
    tfm[0] = T[0]; tfm[1] = T[1]; tfm[2] = T[2]; tfm[3] = T[9];
    tfm[4] = T[3]; tfm[5] = T[4]; tfm[6] = T[5]; tfm[7] = T[10];
    m_ox = cm.origin[2 * inst];
    m_oy = cm.origin[2 * inst + 1];
  }
  const bool has_obst = (cm.layers & (NAVGPU_LAYER_OBSTACLE | NAVGPU_LAYER_VOXEL)) && cm.obs_enabled;
  uint4 sv = has_static ? *reinterpret_cast<const uint4*>(cm.stat + off) : make_uint4(0, 0, 0, 0);
  uint4 lv = has_obst ? *reinterpret_cast<const uint4*>(cm.obst + off) : make_uint4(0, 0, 0, 0);
  uint32_t mw[4] = {mv.x, mv.y, mv.z, mv.w}, sw[4] = {sv.x, sv.y, sv.z, sv.w}, lw[4] = {lv.x, lv.y, lv.z, lv.w};
  int y = row_first;
  int x = base - row_first * cm.nx;
  bool changed = false;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (base + k < cm.cells && x >= x0 && x < xn && y >= y0 && y < yn) {
      const int sh = (k & 3) * 8;
      uint8_t m = layer_only ? (uint8_t)((mw[k >> 2] >> sh) & 0xFF) : cm.master_default;  // resetMap
      if (has_static) {
        uint8_t sc = (sw[k >> 2] >> sh) & 0xFF;
        if (!cm.static_use_maximum)
          m = sc;
        else if (sc != kNoInfo && (m == kNoInfo || m < sc))
          m = sc;
      }
      if (roll_static) {
        // static_layer.cpp:318-330: mapToWorld(i, j) of the master, tf::Transform::operator() (basis row . point + origin,
        // z = 0), the static map's own worldToMap; plain max with use_maximum, plain copy without
        const double wx = m_ox + (x + 0.5) * cm.res, wy = m_oy + (y + 0.5) * cm.res;
        const double px = tfm[0] * wx + tfm[1] * wy + tfm[2] * 0.0 + tfm[3];
        const double py = tfm[4] * wx + tfm[5] * wy + tfm[6] * 0.0 + tfm[7];
        if (!(px < cm.stat_ox || py < cm.stat_oy)) {
          const uint32_t smx = (uint32_t)(int)((px - cm.stat_ox) / cm.stat_res), smy = (uint32_t)(int)((py - cm.stat_oy) / cm.stat_res);
          if (smx < cm.stat_nx && smy < cm.stat_ny) {
            const uint8_t sc = cm.stat_roll[(size_t)smy * cm.stat_nx + smx];
            m = cm.static_use_maximum ? (sc > m ? sc : m) : sc;
          }
        }
      }
      if (has_obst) {
        uint8_t lc = (lw[k >> 2] >> sh) & 0xFF;
        if (lc != kNoInfo) {
          if (cm.combination_method == 0)
            m = lc;
          else if (cm.combination_method == 1 && (m == kNoInfo || m < lc))
            m = lc;
        }
      }
      mw[k >> 2] = (mw[k >> 2] & ~(0xFFu << sh)) | ((uint32_t)m << sh);
      changed = true;
    }
    if (++x == (int)cm.nx) {
      x = 0;
      ++y;
    }
  }
  if (changed) *reinterpret_cast<uint4*>(cm.master + off) = make_uint4(mw[0], mw[1], mw[2], mw[3]);
}

void launch_merge(const CostmapDev& cm, uint32_t first, uint32_t count, const int32_t* boxes, hipStream_t s, bool layer_only) {
  uint32_t groups = (cm.cells + 15) / 16;
  dim3 grid((groups + 255) / 256, count);
  hipLaunchKernelGGL(k_merge, grid, dim3(256), 0, s, cm, first, boxes, cm.static_received, layer_only ? 1 : 0);
}

// ------------------------------------------------------------------------------------------------
// k_inflate: InflationLayer::updateCosts (plugins/inflation_layer.cpp:172-266) restated as an
// order-independent windowed exact Euclidean transform: a cell takes
//     max over LETHAL seeds s inside the grown box, |c - s| <= R, of cached_costs_[|dx|][|dy|]
// ( == the cost of the nearest seed, the tables being monotone in distance ), merged with the
// old value by the reference rule (:243-247).  Separable: pass A finds, per row, the nearest
// seed column distance; pass B takes the max over the 2R+1 rows through the cost table.
// Tile = 64x64 output cells per 256-thread workgroup, (64+2R)^2 seed flags staged in LDS.
// Bit-exact against oracle updateCostsExact(); >= the reference's priority-queue result, which
// itself depends on std::priority_queue tie order (SURVEY §7 hard part 1).
// ------------------------------------------------------------------------------------------------
constexpr int kTile = 64;
__global__ __launch_bounds__(256) void k_inflate(CostmapDev cm, uint32_t first, const int32_t* boxes) {
  extern __shared__ __align__(16) uint8_t lds[];
  const uint32_t inst = first + blockIdx.z;
  const int R = (int)cm.R;
  int min_i, min_j, max_i, max_j;
  if (boxes) {
    min_i = boxes[4 * blockIdx.z + 0];
    min_j = boxes[4 * blockIdx.z + 1];
    max_i = boxes[4 * blockIdx.z + 2];
    max_j = boxes[4 * blockIdx.z + 3];
  } else {
    const InstCostmapState* st = cm.state + inst;
    if (!st->box_valid) return;
    min_i = st->box[0];
    max_i = st->box[1];
    min_j = st->box[2];
    max_j = st->box[3];
  }
  // grown + clamped box (:205-213)
  min_i = max(0, min_i - R);
  min_j = max(0, min_j - R);
  max_i = min((int)cm.nx, max_i + R);
  max_j = min((int)cm.ny, max_j + R);
  const int tx0 = blockIdx.x * kTile, ty0 = blockIdx.y * kTile;
  // seeds that can reach this tile lie in [tx0-R, tx0+kTile+R) x [ty0-R, ty0+kTile+R)
  if (tx0 - R >= max_i || tx0 + kTile + R <= min_i || ty0 - R >= max_j || ty0 + kTile + R <= min_j) return;

  const int W = kTile + 2 * R;           // halo'd tile edge
  const int WS = (W + 3) & ~3;           // padded row stride of the seed flags
  uint8_t* seed = lds;                   // [W][WS]
  uint8_t* hd = seed + W * WS;           // [W][kTile] nearest seed |dx| per row, 255 = none within R
  uint8_t* lut = hd + W * kTile;         // [(R+2)^2]
  __shared__ int s_any;
  const uint32_t tid = threadIdx.x;
  uint8_t* master = cm.master + (size_t)inst * cm.cells_padded;
  if (tid == 0) s_any = 0;
  for (int i = tid; i < (R + 2) * (R + 2); i += blockDim.x) lut[i] = cm.lut[i];
  __syncthreads();
  int any = 0;
  for (int i = tid; i < W * W; i += blockDim.x) {
    int ly = i / W, lx = i - ly * W;
    int gx = tx0 - R + lx, gy = ty0 - R + ly;
    uint8_t f = 0;
    if (gx >= min_i && gx < max_i && gy >= min_j && gy < max_j) f = master[gy * cm.nx + gx] == kLethal;
    seed[ly * WS + lx] = f;
    any |= f;
  }
  if (any) s_any = 1;
  __syncthreads();
  if (!s_any) return;
  // pass A
  for (int i = tid; i < W * kTile; i += blockDim.x) {
    int ly = i / kTile, c = i - ly * kTile;
    const uint8_t* row = seed + ly * WS + c + R;
    uint8_t best = 255;
    for (int d = 0; d <= R; ++d) {
      if (row[d] | row[-d]) {
        best = (uint8_t)d;
        break;
      }
    }
    hd[i] = best;
  }
  __syncthreads();
  // pass B + merge
  const int n = R + 2;
  for (int i = tid; i < kTile * kTile; i += blockDim.x) {
    int ry = i / kTile, c = i - ry * kTile;
    int gx = tx0 + c, gy = ty0 + ry;
    if (gx >= (int)cm.nx || gy >= (int)cm.ny) continue;
    uint8_t cost = 0;
    for (int dy = -R; dy <= R; ++dy) {
      uint8_t h = hd[(ry + R + dy) * kTile + c];
      if (h != 255) {
        int ady = dy < 0 ? -dy : dy;
        uint8_t cc = lut[h * n + ady];
        cost = cc > cost ? cc : cost;  // lut holds 0 beyond the radius
      }
    }
    if (cost == 0) continue;  // max(old, 0) == old and the 255-rule needs cost >= 253
    uint8_t old = master[gy * cm.nx + gx];
    uint8_t nv = (old == kNoInfo && cost >= kInscribed) ? cost : (old > cost ? old : cost);
    if (nv != old) master[gy * cm.nx + gx] = nv;
  }
}

// ------------------------------------------------------------------------------------------------
// k_inflate_bits: the same windowed exact EDT, bit-parallel (inflation radius <= 14 cells).
//   * seeds of a halo'd tile row are four 32-bit words built with wave ballots (no per-cell LDS
//     traffic); the nearest seed column distance of a cell is two bit scans (ctz / clz) on a 2R+1
//     bit window of that row;
//   * per-row distances are 4-bit nibbles packed 8 rows per word, column-major, so one lane
//     (= one column) reads 5 words and holds every row it needs in registers;
//   * a cell's squared distance is min over dy of h(dy)^2 + dy^2 (3 VALU ops per dy, constants
//     folded when R is the template value) and its cost one lookup in a table indexed by d^2
//     (host-built from the reference's pairwise table; monotone in d^2, checked on the host).
// Tile = 64 columns (one per lane) x 32 rows per 256-thread workgroup.
// ------------------------------------------------------------------------------------------------
constexpr int kBX = 64, kBY = 32;
template <int RT>
__global__ __launch_bounds__(256) void k_inflate_bits(CostmapDev cm, uint32_t first, uint32_t count, const int32_t* boxes) {
  // XCD-aware tile order: workgroups b and b + 8 of a launch (linear id, x fastest) share an XCD and its L2.  The grid is
  // (8 * tiles_x, tiles_y, robots / 8) and the robot 8 * z + x % 8, so a robot's tiles follow one another on ONE XCD: the 2R
  // halo a tile reads around its 64 x 32 cells is its neighbours' interior and comes from that L2 instead of from the fabric
  // once per XCD (FETCH_SIZE x 2 + WRITE_SIZE per launch: 84.8 -> 19.4 MB for the benchmark's 256 windows)
  const uint32_t tbx = blockIdx.x >> 3, tby = blockIdx.y, rz = blockIdx.z * 8u + (blockIdx.x & 7u);
  if (rz >= count) return;
  const uint32_t inst = first + rz;
  const int R = RT > 0 ? RT : (int)cm.R;
  int min_i, min_j, max_i, max_j;
  if (boxes) {
    min_i = boxes[4 * rz + 0];
    min_j = boxes[4 * rz + 1];
    max_i = boxes[4 * rz + 2];
    max_j = boxes[4 * rz + 3];
  } else {
    const InstCostmapState* st = cm.state + inst;
    if (!st->box_valid) return;
    min_i = st->box[0];
    max_i = st->box[1];
    min_j = st->box[2];
    max_j = st->box[3];
  }
  min_i = max(0, min_i - R);
  min_j = max(0, min_j - R);
  max_i = min((int)cm.nx, max_i + R);
  max_j = min((int)cm.ny, max_j + R);
  const int tx0 = (int)tbx * kBX, ty0 = (int)tby * kBY;
  if (tx0 - R >= max_i || tx0 + kBX + R <= min_i || ty0 - R >= max_j || ty0 + kBY + R <= min_j) return;

  constexpr int kMaxHR = kBY + 2 * 14;      // halo'd rows
  constexpr int kOct = (kMaxHR + 7) / 8;    // row octets
  __shared__ uint32_t s_rows[kMaxHR][4];    // seed bits of columns [tx0-32, tx0+96)
  __shared__ uint32_t s_hd[kOct + 1][kBX];  // nibble-packed nearest-seed |dx| per (row octet, column)
  __shared__ uint8_t s_lut2[256];           // cost by squared distance
  __shared__ int s_any;
  __shared__ uint32_t s_rowany[2];                // bit r: halo row r of the tile holds a seed (pass B skips the others)
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int HR = kBY + 2 * R;
  uint8_t* master = cm.master + (size_t)inst * cm.cells_padded;
  if (tid == 0) s_any = 0;
  if (tid < 2) s_rowany[tid] = 0;
  s_lut2[tid] = cm.lut2[tid];
  __syncthreads();
  // ---- seed bitmaps by ballot: wave w takes halo rows w, w+4, ...
  // (all of a wave's loads are issued before the first ballot needs one: in a rolled loop every load is waited for on
  // its own, 28 latencies in a row - that was most of this kernel's 57 us)
  int any = 0;
  uint32_t rows_lo = 0, rows_hi = 0;  // (wave-uniform) this wave's halo rows that hold a seed
  constexpr int kSeedIt = (kMaxHR + 3) / 4;
  // The loads are unconditional (clamped coordinates; a conditional load is waited for inside its branch) and the
  // conditions are applied to what they return.
  uint8_t cell[kSeedIt][2];
  const int gxa = tx0 - 32 + (int)lane, gxb = gxa + 64;
  // (clamped into the columns / rows that can hold a seed this tile needs: what lies outside is masked below, and loading it
  // would fetch lines of the map nobody uses)
  const int lo_x = max(min_i, tx0 - R), lo_y = max(min_j, ty0 - R);  // (min_* >= 0; an inverted box leaves hi < lo: still on the map)
  const int hi_x = max(lo_x, min(max_i, tx0 + kBX + R) - 1), hi_y = max(lo_y, min(max_j, ty0 + kBY + R) - 1);
  const int cxa = min(min(max(gxa, lo_x), hi_x), (int)cm.nx - 1), cxb = min(min(max(gxb, lo_x), hi_x), (int)cm.nx - 1);
  const bool xa_ok = gxa >= min_i && gxa < max_i && gxa >= tx0 - R && gxa < tx0 + kBX + R;
  const bool xb_ok = gxb >= min_i && gxb < max_i && gxb >= tx0 - R && gxb < tx0 + kBX + R;
#pragma unroll
  for (int it = 0; it < kSeedIt; ++it) {
    const int gy = ty0 - R + (int)wave + 4 * it;
    const int cy = min(min(max(gy, lo_y), hi_y), (int)cm.ny - 1);
    cell[it][0] = master[cy * cm.nx + cxa];
    cell[it][1] = master[cy * cm.nx + cxb];
  }
#pragma unroll
  for (int it = 0; it < kSeedIt; ++it) {
    const int r = (int)wave + 4 * it, gy = ty0 - R + r;
    if (r < HR) {  // (wave-uniform)
      const bool row_ok = gy >= min_j && gy < max_j;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const unsigned long long m = __ballot(row_ok && (half ? xb_ok : xa_ok) && cell[it][half] == kLethal);
        if (lane == 0) {
          s_rows[r][2 * half] = (uint32_t)m;
          s_rows[r][2 * half + 1] = (uint32_t)(m >> 32);
        }
        any |= m != 0ull;
        if (m != 0ull) {
          if (r < 32)
            rows_lo |= 1u << r;
          else
            rows_hi |= 1u << (r - 32);
        }
      }
    }
  }
  if (any && lane == 0) {
    s_any = 1;
    if (rows_lo) atomicOr(&s_rowany[0], rows_lo);
    if (rows_hi) atomicOr(&s_rowany[1], rows_hi);
  }
  __syncthreads();
  if (!s_any) return;
  // ---- pass A: nearest seed |dx| (nibble, 15 = none within R) for every (halo row, column)
  const uint32_t lowmask = (1u << R) - 1u;
  for (int o = wave; o * 8 < HR; o += 4) {
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = o * 8 + k;
      uint32_t h = 15;
      // (a row without seeds needs no bit scans: wave-uniform, the row words are the same for every lane)
      if (r < HR && (s_rows[r][0] | s_rows[r][1] | s_rows[r][2] | s_rows[r][3]) != 0u) {
        // cell column c = lane sits at bit 32 + c of the 128-bit row; window = bits [32+c-R, 32+c+R]
        const uint32_t sft = 32u + lane - (uint32_t)R;
        const uint32_t wi = sft >> 5, bs = sft & 31u;
        const unsigned long long two = ((unsigned long long)s_rows[r][wi + 1 < 4 ? wi + 1 : 3] << 32) | s_rows[r][wi];
        const uint32_t q = (uint32_t)(two >> bs);
        const uint32_t qr = q >> R;        // bit k: seed at dx = +k (k = 0..R after masking below)
        const uint32_t ql = q & lowmask;   // bit j: seed at dx = j - R
        uint32_t dr = 15, dl = 15;
        if (qr & ((2u << R) - 1u)) dr = (uint32_t)__builtin_ctz(qr & ((2u << R) - 1u));
        if (ql) dl = (uint32_t)R - (31u - (uint32_t)__builtin_clz(ql));
        h = dr < dl ? dr : dl;
      }
      packed |= h << (4 * k);
    }
    s_hd[o][lane] = packed;
  }
  __syncthreads();
  // ---- pass B: wave w owns output rows [8w, 8w+8) of the tile; halo rows [8w, 8w+8+2R) = 5 words
  uint32_t hw[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) hw[k] = (wave + k) * 8 < (uint32_t)HR ? s_hd[wave + k][lane] : 0xFFFFFFFFu;
  // nibble 15 = no seed within R in that row: a wave all of whose columns see none in any of its 8 + 2R rows writes nothing
  if (__ballot((hw[0] & hw[1] & hw[2] & hw[3] & hw[4]) != 0xFFFFFFFFu) == 0ull) return;
  const int gx = tx0 + (int)lane;
  const int R2 = R * R;
  // R known at compile time: the halo rows outermost, and only those that hold a seed (wave-uniform: s_rowany) - a row's h^2 is worked
  // out once and goes into the (up to 8) output rows it is within R of: 2 instructions per (row, output row) pair instead of 3 per
  // pair of ALL 8 + 2R rows
  uint32_t bestr[8] = {0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu};
  if (RT > 0) {
    const unsigned long long rowany = (((unsigned long long)s_rowany[1] << 32) | s_rowany[0]) >> (8u * wave);  // bit p: this wave's halo row p
    const uint32_t ra_lo = __builtin_amdgcn_readfirstlane((uint32_t)rowany), ra_hi = __builtin_amdgcn_readfirstlane((uint32_t)(rowany >> 32));
#pragma unroll
    for (int p = 0; p < 8 + 2 * RT; ++p) {
      if (((p < 32 ? ra_lo >> (p & 31) : ra_hi >> (p & 31)) & 1u) != 0u) {  // (wave-uniform)
        const uint32_t h = (hw[p >> 3] >> (4 * (p & 7))) & 15u;
        const uint32_t hh = h * h;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int dy = p - RT - j;
          if (dy >= -RT && dy <= RT) {
            const uint32_t d2 = hh + (uint32_t)(dy * dy);
            bestr[j] = d2 < bestr[j] ? d2 : bestr[j];
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int gy = ty0 + (int)wave * 8 + j;
    uint32_t best = bestr[j];
    if (RT == 0) {
      for (int dy = -R; dy <= R; ++dy) {
        const int p = j + R + dy;
        const uint32_t wsel = p >> 3;
        const uint32_t word = wsel == 0 ? hw[0] : (wsel == 1 ? hw[1] : (wsel == 2 ? hw[2] : (wsel == 3 ? hw[3] : hw[4])));
        const uint32_t h = (word >> (4 * (p & 7))) & 15u;
        const uint32_t d2 = h * h + (uint32_t)(dy * dy);
        best = d2 < best ? d2 : best;
      }
    }
    if (best > (uint32_t)R2 || gx >= (int)cm.nx || gy >= (int)cm.ny) continue;
    const uint8_t cost = s_lut2[best];
    if (cost == 0) continue;
    const uint8_t old = master[gy * cm.nx + gx];
    const uint8_t nv = (old == kNoInfo && cost >= kInscribed) ? cost : (old > cost ? old : cost);
    if (nv != old) master[gy * cm.nx + gx] = nv;
  }
}

// ------------------------------------------------------------------------------------------------
// k_inflate_pq: InflationLayer::updateCosts + enqueue (plugins/inflation_layer.cpp:172-293) AS WRITTEN - the
// reference-order mode (navgpu_inflation_params::priority_queue_order).  The reference's result depends on the
// order in which std::priority_queue<CellData> pops equal distances (SURVEY 7 hard part 1), i.e. on libstdc++'s
// push_heap / pop_heap (bits/stl_heap.h: __push_heap, __adjust_heap), which are deterministic sift-up / sift-down
// loops; they are restated below verbatim in structure, with the reference's comparison (inflation_layer.h:82-85:
// a < b  <=>  a.distance_ > b.distance_).  The keys are the RANKS of the reference's cached_distances_ (host hypot())
// among their distinct values - same order, same ties, two bytes instead of a double - so a heap entry is 8 bytes:
// cell index, rank, and the source as a signed offset from the cell.
// The walk is sequential by nature: one lane per robot does it.  What makes it usable in a control loop is where its
// state lives (a pop is a chain of ~log2(n) dependent reads: 50 ns each from LDS, 0.5 us from HBM): the two tables,
// seen_ as a bitmap of the window and the window's cost bytes (loaded and written back by the whole wave) in LDS, and
// whatever is left of 150 KB for the top of the heap (14 000 entries for a control cycle's window; every sift passes
// through the top levels), the rest of it in HBM.  The window is the
// box + 2 R + 1 cells: seeds come from box + R (:204-212) and reach R further; a neighbour's seen_ flag is read before
// its distance is tested (:280-286), one cell beyond that.  Larger windows fall back to global memory piece by piece
// (bitmap up to 163 840 cells, bytes up to 65 536).  Byte-identical to the reference.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kPqSeenWords = 5120, kPqWinBytes = 65536;  // largest window whose seen_ bitmap / cost bytes live in LDS
constexpr size_t kPqLdsBytes = 150 * 1024;                    // one workgroup per CU: the rest of it is the heap's top
constexpr uint32_t kPqTabEntries = 66 * 66;
__global__ __launch_bounds__(64) void k_inflate_pq(CostmapDev cm, uint32_t first, const int32_t* boxes) {
  extern __shared__ __align__(16) uint8_t pq_sm[];
  const uint32_t inst = first + blockIdx.x;
  const uint32_t tid = threadIdx.x;
  int min_i, min_j, max_i, max_j;
  if (boxes) {
    min_i = boxes[4 * blockIdx.x + 0];
    min_j = boxes[4 * blockIdx.x + 1];
    max_i = boxes[4 * blockIdx.x + 2];
    max_j = boxes[4 * blockIdx.x + 3];
  } else {
    const InstCostmapState* st = cm.state + inst;
    if (!st->box_valid) return;
    min_i = st->box[0];
    max_i = st->box[1];
    min_j = st->box[2];
    max_j = st->box[3];
  }
  uint8_t* master = cm.master + (size_t)inst * cm.cells_padded;
  uint8_t* seen_g = cm.pq_seen + (size_t)inst * cm.cells_padded;
  PqCell* heap_g = cm.pq_heap + (size_t)inst * cm.pq_cap;
  const uint32_t size_x = cm.nx, size_y = cm.ny;
  const int R = (int)cm.R;
  const uint32_t n = cm.R + 2;  // stride of the caches
  min_i = max(0, min_i - R);  // :204-212
  min_j = max(0, min_j - R);
  max_i = min((int)size_x, max_i + R);
  max_j = min((int)size_y, max_j + R);
  if (max_i <= min_i || max_j <= min_j) return;  // (uniform) no seed row / column: the loops below never run
  // everything the walk can touch
  const int wx0 = max(0, min_i - R - 1), wy0 = max(0, min_j - R - 1), wx1 = min((int)size_x, max_i + R + 1), wy1 = min((int)size_y, max_j + R + 1);
  const uint32_t ww = (uint32_t)(wx1 - wx0), wh = (uint32_t)(wy1 - wy0), wcells = ww * wh;
  const bool seen_lds = wcells <= kPqSeenWords * 32u, win_lds = wcells <= kPqWinBytes;
  // LDS: rank and cost tables | seen_ bitmap | window bytes | heap top (whatever is left)
  uint16_t* rank = reinterpret_cast<uint16_t*>(pq_sm);
  uint8_t* lut_l = pq_sm + kPqTabEntries * 2;
  const uint32_t seen_off = (kPqTabEntries * 3 + 15u) & ~15u;
  uint32_t* seen_l = reinterpret_cast<uint32_t*>(pq_sm + seen_off);
  const uint32_t win_off = seen_off + (seen_lds ? (((wcells + 31) / 32 * 4 + 15u) & ~15u) : 0u);
  uint8_t* win_l = pq_sm + win_off;
  const uint32_t heap_off = win_off + (win_lds ? ((wcells + 15u) & ~15u) : 0u);
  PqCell* heap_l = reinterpret_cast<PqCell*>(pq_sm + heap_off);
  const long heap_lds = (long)((kPqLdsBytes - heap_off) / sizeof(PqCell));
  for (uint32_t i = tid; i < n * n; i += 64) {
    rank[i] = reinterpret_cast<const uint16_t*>(cm.dist_lut)[i];
    lut_l[i] = cm.lut[i];
  }
  // memset(seen_, false, ...) (:198), and the window's bytes
  if (seen_lds) {
    for (uint32_t i = tid; i < (wcells + 31) / 32; i += 64) seen_l[i] = 0;
  } else {
    for (uint32_t i = tid * 16; i < cm.cells_padded; i += 64 * 16) *reinterpret_cast<uint4*>(seen_g + i) = make_uint4(0, 0, 0, 0);
  }
  if (win_lds)
    for (uint32_t i = tid; i < wcells; i += 64) {
      const uint32_t y = i / ww, x = i - y * ww;
      win_l[i] = master[(uint32_t)(wy0 + (int)y) * size_x + (uint32_t)(wx0 + (int)x)];
    }
  __syncthreads();
  if (tid == 0) {
    long len = 0;  // inflation_queue_.size()
    auto hget = [&](long i) -> PqCell { return i < heap_lds ? heap_l[i] : heap_g[i]; };
    auto hset = [&](long i, const PqCell& v) {
      if (i < heap_lds)
        heap_l[i] = v;
      else
        heap_g[i] = v;
    };
    auto widx = [&](uint32_t mx, uint32_t my) { return (my - (uint32_t)wy0) * ww + (mx - (uint32_t)wx0); };
    auto seenGet = [&](uint32_t index, uint32_t mx, uint32_t my) -> bool {
      if (seen_lds) {
        const uint32_t w = widx(mx, my);
        return (seen_l[w >> 5] >> (w & 31)) & 1u;
      }
      return seen_g[index] != 0;
    };
    auto seenSet = [&](uint32_t index, uint32_t mx, uint32_t my) {
      if (seen_lds) {
        const uint32_t w = widx(mx, my);
        seen_l[w >> 5] |= 1u << (w & 31);
      } else {
        seen_g[index] = 1;
      }
    };
    // std::push_heap: __push_heap(first, holeIndex, topIndex = 0, value, comp)
    auto pushHeap = [&](long hole, const PqCell& value) {
      long parent = (hole - 1) / 2;
      while (hole > 0) {
        const PqCell pv = hget(parent);
        if (!(pv.rank > value.rank)) break;
        hset(hole, pv);
        hole = parent;
        parent = (hole - 1) / 2;
      }
      hset(hole, value);
    };
    // enqueue (:277-293) in two halves: what it READS (the neighbour's seen_ flag, the cached distance) does not depend on
    // the queue, so the four neighbours' reads are issued together; the push itself stays in the reference's order
    auto probe = [&](bool in_map, uint32_t index, uint32_t mx, uint32_t my, uint32_t sx, uint32_t sy) -> uint32_t {  // rank, or 0xFFFF = no push
      if (!in_map) return 0xFFFFu;
      const uint32_t dx = mx > sx ? mx - sx : sx - mx, dy = my > sy ? my - sy : sy - my;
      const uint32_t r = rank[dx * n + dy];  // 0xFFFF: distance > cell_inflation_radius_ (:286)
      return seenGet(index, mx, my) ? 0xFFFFu : r;
    };
    auto push = [&](uint32_t r, uint32_t index, uint32_t mx, uint32_t my, uint32_t sx, uint32_t sy) {
      if (r == 0xFFFFu) return;
      if ((uint64_t)len >= cm.pq_cap) return;  // cannot happen: capacity = 4 pushes per cell + the seeds
      PqCell c;
      c.index = index;
      c.rank = (uint16_t)r;
      c.sdx = (int8_t)((int)sx - (int)mx);
      c.sdy = (int8_t)((int)sy - (int)my);
      pushHeap(len++, c);  // priority_queue::push = push_back + push_heap
    };
    auto enqueue = [&](uint32_t index, uint32_t mx, uint32_t my, uint32_t sx, uint32_t sy) { push(probe(true, index, mx, my, sx, sy), index, mx, my, sx, sy); };
    for (int j = min_j; j < max_j; j++)  // :214-226
      for (int i = min_i; i < max_i; i++) {
        const uint32_t index = (uint32_t)j * size_x + (uint32_t)i;
        const uint8_t c0 = win_lds ? win_l[widx((uint32_t)i, (uint32_t)j)] : master[index];
        if (c0 == kLethal) enqueue(index, i, j, i, j);
      }
    while (len > 0) {  // :228-266
      const PqCell cur = hget(0);  // top()
      // priority_queue::pop = pop_heap + pop_back; std::pop_heap acts only on more than one element
      if (len > 1) {
        const PqCell value = hget(len - 1);
        const long l = len - 1;  // __adjust_heap(first, 0, l, value)
        long hole = 0, child = 0;
        while (child < (l - 1) / 2) {
          child = 2 * (child + 1);
          const PqCell a = hget(child), b = hget(child - 1);
          if (a.rank > b.rank) {  // comp(first + secondChild, first + (secondChild - 1))
            child--;
            hset(hole, b);
          } else {
            hset(hole, a);
          }
          hole = child;
        }
        if ((l & 1) == 0 && child == (l - 2) / 2) {
          child = 2 * (child + 1);
          hset(hole, hget(child - 1));
          hole = child - 1;
        }
        pushHeap(hole, value);
      }
      --len;
      const uint32_t index = cur.index;
      const uint32_t my = index / size_x, mx = index - my * size_x;
      if (seenGet(index, mx, my)) continue;
      seenSet(index, mx, my);
      const uint32_t sx = (uint32_t)((int)mx + cur.sdx), sy = (uint32_t)((int)my + cur.sdy);
      const uint32_t dx = mx > sx ? mx - sx : sx - mx, dy = my > sy ? my - sy : sy - my;
      const uint8_t cost = lut_l[dx * n + dy];  // costLookup (within the radius the table is cached_costs_)
      uint8_t* cellp = win_lds ? &win_l[widx(mx, my)] : &master[index];
      const uint8_t old_cost = *cellp;
      if (old_cost == kNoInfo && cost >= kInscribed)
        *cellp = cost;
      else
        *cellp = old_cost > cost ? old_cost : cost;
      const uint32_t r0 = probe(mx > 0, index - 1, mx - 1, my, sx, sy), r1 = probe(my > 0, index - size_x, mx, my - 1, sx, sy),
                     r2 = probe(mx < size_x - 1, index + 1, mx + 1, my, sx, sy), r3 = probe(my < size_y - 1, index + size_x, mx, my + 1, sx, sy);
      push(r0, index - 1, mx - 1, my, sx, sy);
      push(r1, index - size_x, mx, my - 1, sx, sy);
      push(r2, index + 1, mx + 1, my, sx, sy);
      push(r3, index + size_x, mx, my + 1, sx, sy);
    }
  }
  __syncthreads();
  if (win_lds)
    for (uint32_t i = tid; i < wcells; i += 64) {
      const uint32_t y = i / ww, x = i - y * ww;
      master[(uint32_t)(wy0 + (int)y) * size_x + (uint32_t)(wx0 + (int)x)] = win_l[i];
    }
}

void launch_inflate(const CostmapDev& cm, uint32_t first, uint32_t count, const int32_t* boxes, hipStream_t s) {
  if (!cm.infl_enabled) return;
  const int R = (int)cm.R;
  if (cm.infl_pq) {
    if (!cm.pq_heap || !cm.pq_seen || !cm.dist_lut) return;  // (a reconfigure that could not allocate them keeps the previous mode: navgpu_inflation_configure)
    hipFuncSetAttribute((const void*)k_inflate_pq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPqLdsBytes);
    hipLaunchKernelGGL(k_inflate_pq, dim3(count), dim3(64), kPqLdsBytes, s, cm, first, boxes);
    return;
  }
  if (cm.lut2_ok && R >= 1 && R <= 14) {
    dim3 grid(8u * ((cm.nx + kBX - 1) / kBX), (cm.ny + kBY - 1) / kBY, (count + 7u) / 8u);  // (robot = 8 z + x % 8: see the kernel)
    if (R == 11)
      hipLaunchKernelGGL(k_inflate_bits<11>, grid, dim3(256), 0, s, cm, first, count, boxes);
    else
      hipLaunchKernelGGL(k_inflate_bits<0>, grid, dim3(256), 0, s, cm, first, count, boxes);
    return;
  }
  const int W = kTile + 2 * R, WS = (W + 3) & ~3;
  size_t lds = (size_t)W * WS + (size_t)W * kTile + (size_t)(R + 2) * (R + 2);
  dim3 grid((cm.nx + kTile - 1) / kTile, (cm.ny + kTile - 1) / kTile, count);
  hipLaunchKernelGGL(k_inflate, grid, dim3(256), lds, s, cm, first, boxes);
}

}  // namespace navgpu
