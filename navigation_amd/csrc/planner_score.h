// Shared by planner_score.hip (the general k_score kernels and the per-robot image k_score_prep* builds) and
// planner_score_sweep.hip (the product kernel k_score_sweep, which loads that image).
#pragma once
#include "planner_common.h"

namespace navgpu {

// TABLES (use_dwa && discretize_by_time): the heading sequence theta_k of a sample depends only on its
// v_theta and the step (theta += v_theta*dt, rounded to float each step), so sincos(theta_k),
// sincos(pi/2+theta_k), the rotated footprint vertices and the forward-point offset are computed once
// per (v_theta sample, step) by the workgroup into LDS and shared by all (vx, vy) samples, and lanes
// are mapped so that a wave shares one v_theta: identical edge shapes => convergent Bresenham loops.
// The arithmetic per value is unchanged (same operations, same rounding), only deduplicated.
__host__ __device__ inline size_t score_bits_bytes(int win) {  // [win][nw][4] words (part of the LDS image), 16-byte aligned
  return (((size_t)4 * win * ((win + 31) >> 5) * 4) + 15) & ~(size_t)15;
}
// the two [win][nw][2] word arrays the dilation passes work in: behind the image, only where the image is BUILT (PREP != 2)
__host__ __device__ inline size_t score_scratch_bytes(int win) { return score_bits_bytes(win); }
// PREP: 0 = build the LDS image (window, bitmaps, tables) in this workgroup; 1 = build it and store it to
// pl.prep (k_score_prep*, one workgroup per robot); 2 = load the stored image (the scoring workgroups of
// a robot all use the same one: 74 of them in the 32x32x16 configuration)
// CHUNK: cells of a footprint edge fetched per LDS round trip; the launcher picks the smallest of 6 / 9 / 12 / 16 that
// covers the longest edge (a 0.4 m square at 0.05 m: 9), longer edges take several chunks
// AGG: the MapGridCostFunction options DWAPlanner itself never sets - aggregation Sum / Product and a sideways shift
// (map_grid_cost_function.cpp:75-129) - as navgpu_planner_set_map_grid_options configures them: every live critic looks
// its own cell up at every point (no screen, no shared cell); the product kernels are compiled without it.
// n / d for 0 <= n < 2^22, 1 <= d < 2^22: float quotient + one correction step either way (the generic 32-bit division is
// ~40 vector instructions, and every lane of a scoring workgroup makes two of them)
__device__ __forceinline__ int divSmall(int n, int d) {
  int q = (int)((float)n * __builtin_amdgcn_rcpf((float)d));
  int r = n - q * d;
  if (r < 0) {
    --q;
    r += d;
  }
  if (r >= d) ++q;
  return q;
}

// behind a robot's image in pl.prep (k_score_prep_tab writes them, k_score_sweep reads them): kScoreAuxBytes of per-robot scalars
// and one byte per (vx, vy) pair, at the END of the robot's slot
constexpr uint32_t kScoreAuxBytes = 64;
__host__ __device__ inline uint32_t score_prep_reject_bytes(const PlannerDev& pl) { return (pl.max_axis * pl.max_axis + 255u) & ~255u; }
__host__ __device__ inline uint32_t score_prep_reject_offset(const PlannerDev& pl) { return pl.prep_stride - score_prep_reject_bytes(pl); }
size_t score_table_row_bytes(const PlannerDev& pl);
size_t score_table_lds_bytes(const PlannerDev& pl);
// k_score_sweep (planner_score_sweep.hip): the launch for use_dwa && discretize_by_time with DWAPlanner's own MapGrid options
bool score_sweep_applies(const PlannerDev& pl);
uint32_t launch_score_sweep(const PlannerDev& pl, uint32_t first, uint32_t count, hipStream_t s);  // returns blocks per instance

}  // namespace navgpu
