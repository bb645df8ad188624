// Device helpers shared by the planner kernels' translation units (planner_samples.hip, planner_bfs*.hip,
// planner_score*.hip, planner_select.hip).  gfx950 only.
#pragma once
#include <cstdlib>

#include "navgpu_device.h"

namespace navgpu {

constexpr int kMaxAxis = 128;  // per-axis sample capacity staged in LDS (vsamples + 1 <= 128)

__device__ __forceinline__ Geom geomOf(const PlannerDev& pl, uint32_t inst) {
  return Geom{pl.origin[2 * inst], pl.origin[2 * inst + 1], pl.res, pl.nx, pl.ny};
}

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

// The traversable-cell bitmap the wavefront of grid `which` sweeps, [ny][W] words: the costmap's (k_free_bits), or - for the legacy
// TrajectoryPlanner's path_map_ - the same with the MapCell::within_robot cells set (trajectory_planner.cpp:918-930; obstacle
// cells under the robot's own footprint propagate like free cells, map_grid.cpp:109-115): k_free_bits ORs the free bits
// into pl.within for those launches.
__device__ __forceinline__ const uint32_t* bfsFreeBitmap(const PlannerDev& pl, int which, uint32_t inst, uint32_t words) {
  return ((pl.within != nullptr && which == 0) ? pl.within : pl.bfs_free) + (size_t)inst * words;
}
// ---- launchers that stay inside the planner's translation units
constexpr int kRowsHalo = 7;                       // k_bfs_rows: rows a wave copies from either neighbour = levels between two exchanges
constexpr int kRowsPerWave = 64 - 2 * kRowsHalo;   // rows a wave owns
__host__ __device__ inline uint32_t bfs_rows_waves(uint32_t ny) { return (ny + kRowsPerWave - 1) / kRowsPerWave; }
bool launch_bfs_rows(const PlannerDev& pl, uint32_t first, uint32_t count, hipStream_t s, const uint32_t* order);   // false: not this map's kernel
bool launch_bfs_rows2(const PlannerDev& pl, uint32_t first, uint32_t count, hipStream_t s, const uint32_t* order);  // false: not this map's kernel
uint32_t bfs_cu_count();
bool bfs_rows_fits(uint32_t nx, uint32_t ny);  // the map is k_bfs_rows' (no scratch)

}  // namespace navgpu
