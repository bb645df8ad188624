// Internal header shared by the HIP kernels and the C-ABI host code (gfx950 only).
// Data layout in HBM (all SoA over instances, instance-major):
//   master/static/obstacle : uint8  [n][cells_padded]   row-major, index = my*size_x + mx
//   voxel                  : uint32 [n][cells_padded]
//   path / goal / goal_front: uint32 [n][cells]   MapGrid target_dist of path_costs_ (shared by
//                            alignment_costs_, same target poses), goal_costs_, goal_front_costs_
// MapGrid distances are stored as uint32 (the reference keeps doubles that only ever hold
// integers <= size_x*size_y+1; map_cell.h:44-64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/navgpu.h"

namespace navgpu {

constexpr uint8_t kNoInfo = 255, kLethal = 254, kInscribed = 253, kFree = 0;
constexpr int kMaxFootprint = 32;
constexpr int kCareRows = 128, kCareWords = 4;  // bounded wavefronts: extent of the per-robot pocket mask (k_samples)
#ifndef NAVGPU_SCORE_TAB_THREADS
#define NAVGPU_SCORE_TAB_THREADS 256  // samples per k_score_tab workgroup (512 with a 52 KB image: 0.523 ms; 256 with 26 KB: 0.516 ms, configs[4] 2.17 -> 1.82 ms)
#endif
#ifndef NAVGPU_SCORE_TAB_LDS_KB
#define NAVGPU_SCORE_TAB_LDS_KB 14    // first LDS budget tried for a k_score_tab workgroup's image (window + screens + table rows): 3 v_theta rows at configs[2] (26 KB = 9 rows: k_score 0.526 -> 0.506 ms)
#endif
#ifndef NAVGPU_SCORE_PREP_THREADS
#define NAVGPU_SCORE_PREP_THREADS 256  // threads of the workgroup that builds a robot's image (k_score_prep_*): alone 512 is 3 us faster, but beside the other stream groups' kernels its 8 waves and 53 KB wait longer for a CU (0.872 -> 0.862 ms per step)
#endif
#ifndef NAVGPU_SCORE_TAB_WAVES
#define NAVGPU_SCORE_TAB_WAVES 6      // waves per SIMD k_score_tab is compiled for (6: 80 registers)
#endif
#ifndef NAVGPU_SCORE_THREADS
#define NAVGPU_SCORE_THREADS 256
#endif
// A/B and trace switches read from the environment exist in tool builds only (make EXTRA=-DNAVGPU_DEBUG_SWITCHES, tools/):
// the library that is loaded into move_base reads no environment variable.
#ifdef NAVGPU_DEBUG_SWITCHES
#define NAVGPU_DEBUG_ENV(name) getenv(name)
#else
#define NAVGPU_DEBUG_ENV(name) ((const char*)nullptr)
#endif
constexpr int kScoreThreads = NAVGPU_SCORE_THREADS;  // samples per k_score workgroup  // vertices kept in registers/LDS by the kernels

struct InstCostmapState {  // per-instance state that persists across update cycles (device resident)
  double last_min_x, last_min_y, last_max_x, last_max_y;  // InflationLayer::last_* (inflation_layer.cpp:63-66)
  int32_t need_reinflation;                               // InflationLayer::need_reinflation_
  int32_t static_has_updated_data;                        // StaticLayer::has_updated_data_
  int32_t box[4];                                         // x0, xn, y0, yn of the last updateMap
  int32_t box_valid;                                      // 0 when xn < x0 || yn < y0 (layered_costmap.cpp:128-135)
  int32_t pad;
  double bounds[4];                                       // min_x, min_y, max_x, max_y after all updateBounds
  double extra[4];                                        // CostmapLayer::extra_min_x_, _min_y_, _max_x_, _max_y_ of the obstacle layer
  int32_t has_extra_bounds, pad2;                         // (costmap_layer.cpp:21-60)
};

struct ObsCsr {  // one observation, device form
  uint32_t first_point, n_points, flags, pad;
  double ox, oy, oz;
  double obstacle_range, raytrace_range;
};

// CellData of inflation_layer.h:55-80 as the heap stores it: the key is the RANK of distance_ among the distinct cached
// distances (same order, same ties), the source a signed offset from the cell (x_, y_ follow from index_)
struct PqCell {
  uint32_t index;
  uint16_t rank;
  int8_t sdx, sdy;
};
static_assert(sizeof(PqCell) == 8, "heap entry");
struct CostmapDev {
  uint32_t nx, ny, cells, cells_padded;
  double res;
  int32_t layers, track_unknown;
  uint8_t master_default, obstacle_default;
  // layer params
  int32_t obs_enabled, footprint_clearing, combination_method, static_use_maximum, static_received;
  double max_obstacle_height;
  // voxel
  int32_t z_voxels, unknown_threshold, mark_threshold;
  double origin_z, z_resolution;
  // inflation
  int32_t infl_enabled;
  uint32_t R;  // cell_inflation_radius_
  double inflation_radius;
  // buffers
  double* origin;      // [n][2]
  uint8_t *master, *stat, *obst;
  uint32_t* voxel;
  uint8_t* lut;        // (R+2)^2 cost table with 0 where cached distance > R
  uint8_t* lut2;       // [256] cost by squared cell distance (k_inflate_bits), valid when lut2_ok
  int32_t lut2_ok;
  // reference-order mode (navgpu_inflation_params::priority_queue_order): InflationLayer's own priority-queue walk
  int32_t infl_pq;
  double* dist_lut;    // (R+2)^2 uint16 ranks of cached_distances_ = hypot(i, j) (host libm) among their distinct values; 0xFFFF beyond the radius
  uint8_t* pq_seen;    // [n][cells] seen_
  PqCell* pq_heap;     // [n][pq_cap] the binary heap std::priority_queue<CellData> keeps
  uint64_t pq_cap;
  // StaticLayer of a rolling-window costmap: one static map with its own geometry, a transform per robot
  uint8_t* stat_roll;   // [stat_ny][stat_nx] interpreted costs (nullptr: not a rolling static layer / no map yet)
  uint32_t stat_nx, stat_ny;
  double stat_res, stat_ox, stat_oy;
  double* stat_tf;      // [n][12] tf::Transform map_frame <- global_frame: basis row-major, origin
  InstCostmapState* state;  // [n]
  // staged cycle inputs
  double* pose;        // [n][3]
  double* fp_world;    // [n][kMaxFootprint][2] transformed footprint (host fp64 libm, footprint.cpp:103-118)
  uint32_t* fp_n;      // [n]
  ObsCsr* obs;         // [n][max_obs] observations of the staged cycle (first_point relative to the instance block)
  uint32_t* obs_count; // [n]
  float* points;       // [n][max_points][3] xyz
  uint32_t max_obs, max_points;
  int32_t* shift;      // [n][2] pending rolling-window shift in cells (cell_ox, cell_oy)
  uint8_t *master_alt, *obst_alt;  // ping-pong targets of the shift
  uint32_t* voxel_alt;
  uint2* mark_seq;     // [n][max_points] voxel marking with mark_threshold > 0: (cell, valid | column bits before marking | z) in point order
};

struct PlannerDev {
  uint32_t nx, ny, cells;
  double res, inv_res;
  double* origin;  // [n][2] (shared with the costmap)
  const uint8_t* master;
  uint32_t cells_padded;
  navgpu_dwa_config cfg;
  // derived once per configure (dwa_planner.cpp:64-75)
  double scale_path, scale_goal, scale_obstacle;
  uint32_t max_plan, max_sim_steps, max_axis;  // max_axis = per-axis sample capacity
  uint32_t max_samples, score_blocks;
  // staged inputs
  navgpu_robot_state* state;  // [n]
  double* plan;               // [n][max_plan][2]
  uint32_t* plan_count;       // [n]
  double* front_last;         // [n][2] last pose of front_global_plan (nose goal)
  int32_t* align_on;          // [n] alignment_costs_ scale != 0
  double* fp_spec;            // [n][kMaxFootprint][2] robot-frame footprint
  uint32_t* fp_n;             // [n]
  // per-cycle work buffers
  float* axis_samples;        // [n][3][max_axis]
  int32_t* axis_count;        // [n][4]  (nx, ny, nth, total)
  uint32_t *path, *goal, *goal_front;  // [n][cells] each
  uint32_t* bfs_scratch;      // k_bfs_global bitmaps (only for grids too large for LDS)
  int32_t* bfs_box;           // [n][8] x0, x1, y0, y1 (cells, inclusive) of the robot's region = box of its MapGrid look-ups + 2 cells; x1 < x0 = none; [4] = bfs_care valid (k_samples)
  uint32_t* bfs_care;         // [n][kCareRows][kCareWords] box cells outside pockets, by region row and word from the region's first
  uint32_t* bfs_reach;        // [n] staged half edge of that box in cells, 0 = search the whole grid
  uint32_t bfs_bounded;       // launch switch: stop a wavefront once its robot's box is settled
  unsigned long long* bfs_trace;  // [items][8] wall-clock stamps of the phases of every wavefront (NAVGPU_DEBUG_BFS_TRACE=<file>, tools/trace_bfs.py), else null
  uint32_t* bfs_next_item;    // work counter of the persistent k_bfs_wave launch
  uint32_t* bfs_free;         // [n][ny][W] traversable-cell bitmap of the costmaps (k_free_bits, per launch_bfs)
  uint32_t* bfs_levels;       // [n][3] levels the last wavefront of (robot, grid) ran: predicts the next one's length
  uint32_t* bfs_order;        // [n * 3] items of a launch sorted longest first, stored at first * 3 (k_samples)
  uint32_t bfs_grids;         // wavefronts per robot: 3 (DWA: path, goal, goal_front) or 2 (legacy TrajectoryPlanner)
  uint32_t* within;           // legacy planner: [n][ny][W] MapCell::within_robot bits of path_map_ on entry to launch_bfs, whose k_free_bits ORs the costmap's free bits in (= path_map_'s traversable-cell bitmap); else null
  uint32_t win;               // edge (cells) of the costmap window staged in LDS by k_score
  uint32_t fp_rcells;         // Chebyshev radius (cells) that contains every footprint cell around the centre cell
  uint8_t fp_halfw[32];       // by |dy| <= fp_rcells: the largest |dx| at which a footprint cell can lie dy rows from the centre cell (a disc, planWindow); 0xFF: no such row
  uint32_t fp_chunk;          // cells of the longest footprint edge (+1): picks the k_score<CHUNK> instantiation
  uint32_t use_tables, tab_steps, tab_nfp, tab_nth;
  // MapGridCostFunction options beyond DWAPlanner's own wiring (navgpu_planner_set_map_grid_options), indexed
  // 0 path, 1 goal, 2 goal_front, 3 alignment: aggregation 0 Last | 1 Sum | 2 Product, sideways shift in metres
  int32_t mg_agg[4];
  double mg_yshift[4];
  int32_t mg_generic;         // any of them set: the scoring launches take the general step (score_body<AGG>)
  uint32_t tab_rows;          // v_theta rows per row group = rows of the tables a k_score_tab workgroup keeps in LDS
  uint32_t tab_bytes;         // score_table_bytes(): the tables' share of the LDS image (0 without tables)
  double tab_dt;              // sim_time / tab_steps
  uint8_t* prep;              // [n][prep_stride] LDS image of k_score (window, reach bitmaps, heading tables), built per cycle by k_score_prep*
  uint32_t prep_stride, prep_bytes;
  // k_score<TABLES>: shared per-(v_theta, step) tables in LDS
  double* sample_cost;        // [n][max_samples] or null
  int32_t* sample_status;     // [n][max_samples] or null
  double* part_cost;          // [n][score_blocks]
  int32_t* part_index;        // [n][score_blocks]
  int32_t* counters;          // [n][2] scored, valid
  uint32_t* osc_flags;        // [n]
  float* osc_prev;            // [n][3]
  navgpu_plan_result* result; // [n]
  double* traj;               // [n][max_sim_steps][3]
};

// legacy TrajectoryPlanner (tp_kernels.hip)
struct TpOut {  // per generateTrajectory call
  double cost;
  double ex, ey, eth;  // Trajectory::getEndpoint
  double ahead;        // goal_map_ at the heading_lookahead point of the endpoint
  int32_t n_points;
  int32_t ahead_ok;    // that point is on the map
};
struct TpDev {
  navgpu_tp_config cfg;
  uint32_t max_samples;   // capacity per robot
  double* samples;        // [n][max_samples][3] vx, vy, vtheta
  uint32_t* n_samples;    // [n]
  double* start;          // [n][6] x, y, theta, vx, vy, vtheta (fp64 promotions of the Vector3f)
  TpOut* out;             // [n][max_samples]
  int32_t* winner;        // [n] sample whose points k_tp_rollout stores (second pass), -1 none
  double* points;         // [n][max_sim_steps][3]
  uint32_t* within_cells; // [n][max_within] cell indices under the robot
  uint32_t* within_count; // [n]
  uint32_t max_within;
  uint32_t* within_bits;  // [n][ny][W]
};
void launch_tp_within(const PlannerDev& pl, const TpDev& tp, uint32_t first, uint32_t count, hipStream_t s);
void launch_tp_rollout(const PlannerDev& pl, const TpDev& tp, uint32_t first, uint32_t count, int store_points, hipStream_t s);

// ---- launchers (defined in the .hip files) ---------------------------------------------------
void launch_obstacle(const CostmapDev& cm, uint32_t first, uint32_t count, const double* bounds_in, int only_bounds,
                     hipStream_t s);
void launch_reset_window(uint8_t* grid, size_t stride, uint32_t count, uint32_t nx, uint32_t x0, uint32_t y0, uint32_t xn, uint32_t yn, uint8_t value, hipStream_t s);
void launch_reset_bounding_box(const CostmapDev& cm, uint32_t first, uint32_t count, const double* boxes_world, hipStream_t s);
void launch_merge(const CostmapDev& cm, uint32_t first, uint32_t count, const int32_t* boxes, hipStream_t s, bool layer_only = false);
void launch_inflate(const CostmapDev& cm, uint32_t first, uint32_t count, const int32_t* boxes, hipStream_t s);
void launch_static_interpret(uint8_t* dst, const int8_t* occ, uint32_t cells, uint32_t cells_padded, uint32_t count,
                             int track_unknown_space, int trinary, int lethal_threshold, int unknown_cost_value, hipStream_t s);
void launch_export_window(const uint8_t* master, uint32_t nx, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, int8_t* out, hipStream_t s);
void launch_fill_u8(uint8_t* dst, uint8_t v, size_t n, hipStream_t s);
void launch_fill_u32(uint32_t* dst, uint32_t v, size_t n, hipStream_t s);
void launch_shift_u8(const uint8_t* src, uint8_t* dst, const CostmapDev& cm, uint32_t first, uint32_t count, uint8_t fill, hipStream_t s);
void launch_shift_u32(const uint32_t* src, uint32_t* dst, const CostmapDev& cm, uint32_t first, uint32_t count, uint32_t fill, hipStream_t s);

void launch_cell_costs(const PlannerDev& pl, uint32_t inst, float4* out, hipStream_t s);
void launch_samples(const PlannerDev& pl, uint32_t first, uint32_t count, hipStream_t s);
void launch_bfs(const PlannerDev& pl, uint32_t first, uint32_t count, hipStream_t s, const uint32_t* order = nullptr, bool free_ready = false);
uint32_t launch_score(const PlannerDev& pl, uint32_t first, uint32_t count, const float* explicit_sample, hipStream_t s);  // returns blocks per instance
void launch_select(const PlannerDev& pl, uint32_t first, uint32_t count, uint32_t n_blocks, hipStream_t s);
// the poses of up to 64 robots as one kernel-argument block (< 4 KB)
constexpr uint32_t kPoseChunk = 64;
struct PoseChunk {
  navgpu_robot_state* state;  // device arrays, indexed by absolute instance
  double* front_last;
  int32_t* align_on;
  uint32_t* bfs_reach;
  uint32_t first, count;
  navgpu_robot_state st[kPoseChunk];
  double front[2 * kPoseChunk];
  int32_t align[kPoseChunk];
  uint32_t reach[kPoseChunk];
};
static_assert(sizeof(PoseChunk) <= 4096, "kernel arguments are limited to 4 KB");
void launch_stage_poses(const PoseChunk& c, hipStream_t s);
void launch_sincos(const double* th, uint32_t n, double* sn, double* cs, hipStream_t s);
size_t score_table_bytes(const PlannerDev& pl);
size_t score_window_bytes(uint32_t win);
size_t score_prep_bytes(const PlannerDev& pl);
size_t score_prep_slot_bytes(const PlannerDev& pl);
size_t bfs_scratch_words(uint32_t nx, uint32_t ny);
uint32_t score_table_rows(const PlannerDev& pl, uint32_t win);

// ---- device helpers ---------------------------------------------------------------------------
struct Geom {
  double ox, oy, res;
  uint32_t nx, ny;
};

// Costmap2D::worldToMap (costmap_2d.cpp:208-220).  The explicit NaN / range guards reproduce what
// the x86 `(int)double` conversion yields for out-of-range values (0x80000000 -> "not in map").
__device__ __forceinline__ bool worldToMap(const Geom& g, double wx, double wy, uint32_t& mx, uint32_t& my) {
  if (wx < g.ox || wy < g.oy) return false;
  double fx = (wx - g.ox) / g.res;
  double fy = (wy - g.oy) / g.res;
  if (!(fx < 2147483648.0) || !(fy < 2147483648.0)) return false;
  mx = (uint32_t)(int)fx;
  my = (uint32_t)(int)fy;
  return mx < g.nx && my < g.ny;
}

// |(x, y)| for generic doubles: sqrt of the correctly summed squares (within 1 ulp of libm hypot)
__device__ __forceinline__ double hyp2(double x, double y) {
  double a = x * x, b = y * y;
  double s = a + b;
  double h = sqrt(s);
  if (h == 0.0) return 0.0;
  double bb = s - a;
  double e = (a - (s - bb)) + (b - bb);
  double r = __builtin_fma(-h, h, s);
  return h + (r + e) / (2.0 * h);
}

// navfn::NavFn arrays of a batch of plans (navfn_kernels.hip, navgpu_navfn.cpp)
struct NavfnDev {
  int nx, ny, ns;
  uint32_t ns_padded, path_cap;
  uint8_t *costarr, *pending;   // [n][ns_padded]
  float *potarr, *gradx, *grady;  // [n][ns_padded]
  int* pb;                      // [n][3][PRIORITYBUFSIZE]
  float* path;                  // [n][2][path_cap]: pathx, pathy
  navgpu_navfn_result* results; // [n]
  // tiled wavefront expansion (navgpu_navfn_plan_wavefront), allocated on first use
  float* potalt;                // [n][ns_padded] the second potential array (rounds alternate)
  uint32_t* wf_act;             // [n][2][tiles] marks of this / the next round: 1 copy across, 2 relax
  uint32_t *wf_nchg, *wf_min;   // [n][wf_max_rounds] tiles that changed in a round / smallest value it wrote (float bits)
  struct NavfnWfStatus* wf_status;  // [n]
  int wf_tiles_x, wf_tiles_y, wf_max_rounds;
};
struct NavfnWfStatus {
  int32_t done, final_array, rounds, pad;
};
// which update rule the tiled wavefront relaxes: NavFn::updateCell's (navfn.cpp:466-535) or DijkstraExpansion::updateCell's with
// its cost translation and either potential calculator (dijkstra.cpp:170-229, dijkstra.h:78-87)
struct NavfnWfRule {
  int32_t global_planner, quadratic, outline, allow_unknown;
  int32_t lethal_cost, neutral_cost;
  float cost_factor;
  int32_t max_sweeps;  // red / black sweeps of a tile per round
};
void launch_navfn_wf_init(const NavfnDev& nv, uint32_t first, uint32_t count, const NavfnWfRule& rule, const int32_t* seed_cells, const float* seed_vals,
                          hipStream_t s);
void launch_navfn_wf_round(const NavfnDev& nv, uint32_t first, uint32_t count, const NavfnWfRule& rule, const int32_t* stop_cells, int at_start, int round,
                           hipStream_t s);
void launch_gp_wf_finish(const NavfnDev& nv, uint32_t first, uint32_t count, const navgpu_global_planner_params& gp, const double* starts,
                         const double* goals, const int32_t* goal_cells, hipStream_t s);
void launch_navfn_wf_path(const NavfnDev& nv, uint32_t first, uint32_t count, const int32_t* goals, const int32_t* starts, hipStream_t s);
void launch_navfn_costmap(const NavfnDev& nv, uint32_t first, uint32_t count, const uint8_t* cmap, size_t stride, int cost_mode, int allow_unknown,
                          hipStream_t s);
void launch_gp_plan(const NavfnDev& nv, uint32_t first, uint32_t count, const navgpu_global_planner_params& gp, const double* starts,
                    const double* goals, const int32_t* goal_cells, void* heaps, hipStream_t s);
void launch_navfn_plan(const NavfnDev& nv, uint32_t first, uint32_t count, const int32_t* goals, const int32_t* starts, int astar, int at_start,
                       hipStream_t s);

}  // namespace navgpu
