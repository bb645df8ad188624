/*
 * navgpu.h — C-ABI of the MI355X-native costmap + DWA hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch/ROS types.  The thin C++
 * plugin adapters (navigation_amd/plugin/, `nav_core::BaseLocalPlanner` and `costmap_2d::Layer`
 * subclasses) and the fleet harness (bench.py, tests) call exactly these entry points.  Every
 * entry point cites the reference interface it replaces (paths relative to the reference tree).
 *
 * Model: a *fleet* is N independent robot instances on one GPU, each with its own layered
 * costmap (master grid + layer grids) and DWA planner state, all of one grid size.  A single
 * robot is a fleet of 1.  Calls take an instance range [first, first+count) and are enqueued on
 * the fleet's HIP stream; `navgpu_sync` (or any *_results/_download call) waits for them.
 *
 * Conventions: return 0 on success, negative navgpu_status on error (no exceptions, like the
 * reference's bool / negative-cost error channel); the caller owns every buffer it passes.
 * There is NO CPU fallback: without a usable HIP device `navgpu_fleet_create` fails.
 *
 * Threading: every entry point that takes a fleet (or a navgpu_navfn handle) holds that handle's
 * own recursive mutex for its whole body and makes the handle's GPU current on the calling thread,
 * so calls on ONE handle from several host threads are serialised inside the library: a
 * reconfigure (navgpu_planner_configure, navgpu_inflation_configure, navgpu_obstacle_configure,
 * navgpu_set_footprint ...) issued by a second thread while another is inside a stage / update /
 * cycle waits for that call to return, drains the stream before it frees or re-allocates a device
 * table, and a call that fails leaves the previous configuration in force.  This is the role of
 * DWAPlanner::configuration_mutex_ (dwa_local_planner/src/dwa_planner.cpp:55,301) and
 * InflationLayer::inflation_access_ (costmap_2d/plugins/inflation_layer.cpp:68,112,175).  What the
 * library cannot know is which calls form ONE control cycle: a caller that must not see a
 * reconfigure land between its stage and its cycle holds its own lock around the pair, as the
 * adapters do (navgpu::DWAPlannerROS::configuration_mutex_, the layers' gpu_access_ /
 * inflation_access_), next to the master-costmap mutex the reference already holds around both
 * virtual calls (costmap_2d/src/layered_costmap.cpp:83, move_base/src/move_base.cpp:947).
 * navgpu_fleet_destroy / navgpu_navfn_destroy must not race any other call on the same handle.
 * Distinct handles are independent.  navgpu_last_error is per thread.
 */
#ifndef NAVGPU_H_
#define NAVGPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  NAVGPU_OK = 0,
  NAVGPU_ERR_INVALID = -1,   /* bad argument / range */
  NAVGPU_ERR_NO_DEVICE = -2, /* no HIP device, or kernels not loadable on it */
  NAVGPU_ERR_HIP = -3,       /* HIP runtime error (see navgpu_last_error) */
  NAVGPU_ERR_CAPACITY = -4,  /* input exceeds a capacity given at creation */
  NAVGPU_ERR_STATE = -5      /* call sequence violated (e.g. cycle before configure) */
} navgpu_status;

/* costmap_2d/include/costmap_2d/cost_values.h:42-45 */
#define NAVGPU_NO_INFORMATION 255
#define NAVGPU_LETHAL_OBSTACLE 254
#define NAVGPU_INSCRIBED_INFLATED_OBSTACLE 253
#define NAVGPU_FREE_SPACE 0

/* layer plugins present in an instance's LayeredCostmap, in the reference's plugin order
 * static -> obstacle|voxel -> inflation (costmap_2d/src/costmap_2d_ros.cpp:189-260) */
#define NAVGPU_LAYER_STATIC 1
#define NAVGPU_LAYER_OBSTACLE 2
#define NAVGPU_LAYER_VOXEL 4 /* replaces OBSTACLE: VoxelLayer derives from ObstacleLayer */
#define NAVGPU_LAYER_INFLATION 8

/* device-resident grids of an instance (navgpu_grid_upload / _download / _device) */
typedef enum {
  NAVGPU_GRID_MASTER = 0,     /* LayeredCostmap::costmap_, uint8 [size_y][size_x]                        */
  NAVGPU_GRID_STATIC = 1,     /* StaticLayer's costmap_, uint8                                           */
  NAVGPU_GRID_OBSTACLE = 2,   /* ObstacleLayer/VoxelLayer's costmap_, uint8                              */
  NAVGPU_GRID_VOXEL = 3,      /* VoxelGrid::data_, uint32 per column (voxel_grid.h:66-434)               */
  NAVGPU_GRID_PATH = 4,       /* MapGrid target_dist of path_costs_ (== alignment_costs_), uint32        */
  NAVGPU_GRID_GOAL = 5,       /* MapGrid target_dist of goal_costs_, uint32                              */
  NAVGPU_GRID_GOAL_FRONT = 6  /* MapGrid target_dist of goal_front_costs_, uint32                        */
} navgpu_grid_id;

typedef struct navgpu_fleet navgpu_fleet;

typedef struct {
  uint32_t n_instances;
  uint32_t size_x, size_y;  /* cells; Costmap2D::size_x_/size_y_ (costmap_2d.h:60-466)                   */
  double resolution;        /* m/cell                                                                   */
  int32_t layers;           /* NAVGPU_LAYER_* bitmask                                                   */
  int32_t track_unknown;    /* LayeredCostmap(track_unknown): master default 255 else 0 (layered_costmap.cpp:50-57) */
  int32_t device;           /* HIP device ordinal                                                       */
  uint32_t max_points;      /* capacity: cloud points per instance per update                           */
  uint32_t max_observations;/* capacity: observations per instance per update                           */
  uint32_t max_plan;        /* capacity: poses of the local plan handed to the planner                  */
  uint32_t max_footprint;   /* capacity: footprint polygon vertices                                     */
  uint32_t max_sim_steps;   /* capacity: trajectory points (ceil(sim_time/sim_granularity) or the
                               per-sample bound when discretize_by_time = 0)                            */
  int32_t keep_sample_costs;/* 1: keep every sample's total cost + status for navgpu_planner_samples    */
  int32_t rolling_window;   /* LayeredCostmap(rolling_window): every update re-centres the grids on the robot
                               (Costmap2D::updateOrigin, costmap_2d.cpp:264-313; not with NAVGPU_LAYER_STATIC,
                               whose rolling branch needs tf)                                            */
} navgpu_fleet_desc;

/* One sensor observation of one instance — costmap_2d::Observation (observation.h:46-103):
 * origin_ (double xyz, global frame), cloud_ (float xyz, global frame), the two ranges.  This is
 * where the reference's own tests inject data (ObstacleLayer::addStaticObservation,
 * plugins/obstacle_layer.cpp:450-464; testing_helper.h:75-90). */
typedef struct {
  uint32_t instance;        /* absolute instance index                                                  */
  uint32_t first_point;     /* index of this cloud's first point in the packed points_xyz array         */
  uint32_t n_points;
  uint32_t flags;           /* bit0 marking, bit1 clearing                                              */
  double origin_x, origin_y, origin_z;
  double obstacle_range, raytrace_range;
} navgpu_observation;
#define NAVGPU_OBS_MARKING 1
#define NAVGPU_OBS_CLEARING 2

/* ObstacleLayer / VoxelLayer parameters (cfg/ObstaclePlugin.cfg:7-18, cfg/VoxelPlugin.cfg:10-15) */
typedef struct {
  int32_t enabled;
  int32_t footprint_clearing_enabled;
  int32_t combination_method; /* 0 overwrite, 1 max (costmap_layer.cpp:62-124) */
  int32_t z_voxels;           /* voxel only */
  double max_obstacle_height;
  double origin_z, z_resolution; /* voxel only */
  int32_t unknown_threshold;  /* voxel only, as configured (the +16-z_voxels of voxel_layer.cpp:89 is applied inside) */
  int32_t mark_threshold;     /* voxel only */
} navgpu_obstacle_params;

/* InflationLayer parameters (cfg/InflationPlugin.cfg:8-9) + the footprint's inscribed radius
 * (LayeredCostmap::getInscribedRadius, set through InflationLayer::onFootprintChanged) */
typedef struct {
  int32_t enabled;
  int32_t priority_queue_order; /* 0 (default): the order-independent windowed exact Euclidean transform - every cell takes
                                   the cost of its nearest LETHAL cell - on the parallel kernels.  1: InflationLayer::updateCosts
                                   as written (inflation_layer.cpp:226-293), byte for byte: a cell keeps the source carried by
                                   whichever neighbour std::priority_queue popped first (ties in libstdc++ heap order); one
                                   sequential walk per robot, far slower.  The two differ in ~6e-5 of the cells of a 400x400
                                   map at 1 % obstacles; 0 is never lower than 1 */
  double inflation_radius;
  double cost_scaling_factor;
  double inscribed_radius;
} navgpu_inflation_params;

/* base_local_planner::LocalPlannerLimits (local_planner_limits.h:44-124) + DWAPlannerConfig
 * (dwa_local_planner/cfg/DWAPlanner.cfg:15-36) + the plain params of DWAPlanner's ctor
 * (dwa_planner.cpp:131-181).  Field order is ABI. */
typedef struct {
  double max_trans_vel, min_trans_vel;
  double max_vel_x, min_vel_x, max_vel_y, min_vel_y;
  double max_rot_vel, min_rot_vel;
  double acc_lim_x, acc_lim_y, acc_lim_theta;
  double sim_time, sim_granularity, angular_sim_granularity, sim_period;
  double path_distance_bias, goal_distance_bias, occdist_scale;
  double forward_point_distance, cheat_factor;
  double oscillation_reset_dist, oscillation_reset_angle;
  int32_t vx_samples, vy_samples, vth_samples;
  int32_t use_dwa;            /* only 1 (DWA window, no continued acceleration) is accelerated   */
  int32_t discretize_by_time; /* SimpleTrajectoryGenerator::initialise(..., discretize_by_time)  */
  int32_t sum_scores;         /* ObstacleCostFunction::setSumScores                               */
  int32_t allow_unknown;      /* explicit (reference reads an uninitialised member)  */
  /* Which function computeNewPositions' unqualified cos(pos[2]) / sin(pos[2]) names for its FLOAT argument
   * (simple_trajectory_generator.cpp:253-260) depends on the reference's build, not on its source:
   *   0  ::cos(double) - only the C declarations in scope: the fork's own Kinetic / GCC 5 toolchain (default);
   *   1  the float overload - libstdc++ >= 6 puts std::cos(float) into the global namespace once <math.h> is in scope, and
   *      vel[0] * cos(pos[2]) becomes a float product (the M_PI_2 + pos[2] terms stay double).
   * INTEGRATION.md has a three-line probe that tells which one a workspace builds. */
  int32_t rollout_trig;
} navgpu_dwa_config;

/* Robot state for one planner cycle, already narrowed to float the way DWAPlanner::findBestPath
 * builds its Eigen::Vector3f pos / vel (dwa_planner.cpp:303-304). */
typedef struct {
  float pos[3];             /* x, y, yaw in the costmap's global frame */
  float vel[3];             /* vx, vy, vtheta                           */
  uint32_t plan_first;      /* first pose of this instance in the packed plan_xy array */
  uint32_t plan_count;      /* poses (>= 1); the transformed+pruned local plan          */
} navgpu_robot_state;

/* Result of DWAPlanner::findBestPath (dwa_planner.cpp:292-371) for one instance. */
typedef struct {
  int32_t best_index;       /* sample slot of the winner (x-outer, y, theta-inner order), -1 if none */
  int32_t n_samples;        /* sample slots generated this cycle                                      */
  int32_t n_scored;         /* slots the generator accepted                                           */
  int32_t n_valid;          /* slots with total cost >= 0                                             */
  int32_t n_points;         /* points of the winning trajectory                                       */
  uint32_t oscillation_flags;/* the 12 sticky flags after updateOscillationFlags (bit order: navgpu.h) */
  float xv, yv, thetav;     /* result_traj_.{xv_,yv_,thetav_}                                         */
  float reserved;
  double cost;              /* result_traj_.cost_ (-7 pre-set when nothing is valid)                  */
  double drive[3];          /* drive_velocities: (xv, yv, thetav) or zeros when cost < 0             */
} navgpu_plan_result;

/* oscillation flag bits (OscillationCostFunction members, oscillation_cost_function.cpp:81-97) */
#define NAVGPU_OSC_STRAFE_POS_ONLY (1u << 0)
#define NAVGPU_OSC_STRAFE_NEG_ONLY (1u << 1)
#define NAVGPU_OSC_STRAFING_POS (1u << 2)
#define NAVGPU_OSC_STRAFING_NEG (1u << 3)
#define NAVGPU_OSC_ROT_POS_ONLY (1u << 4)
#define NAVGPU_OSC_ROT_NEG_ONLY (1u << 5)
#define NAVGPU_OSC_ROTATING_POS (1u << 6)
#define NAVGPU_OSC_ROTATING_NEG (1u << 7)
#define NAVGPU_OSC_FORWARD_POS_ONLY (1u << 8)
#define NAVGPU_OSC_FORWARD_NEG_ONLY (1u << 9)
#define NAVGPU_OSC_FORWARD_POS (1u << 10)
#define NAVGPU_OSC_FORWARD_NEG (1u << 11)

/* per-sample status written when keep_sample_costs = 1 */
#define NAVGPU_SAMPLE_REJECTED 0 /* generateTrajectory returned false (simple_trajectory_generator.cpp:193-200,250) */
#define NAVGPU_SAMPLE_SCORED 1

/* ------------------------------------------------------------------------------------------ */
/* lifetime                                                                                   */
/* ------------------------------------------------------------------------------------------ */
const char* navgpu_version(void);
const char* navgpu_strerror(int status);
const char* navgpu_last_error(void); /* text of the last HIP failure on this thread */
int navgpu_device_count(void);

/* replaces: `new Costmap2DROS(...)` + plugin createInstance/initialize for the hot-path layers
 * (costmap_2d/src/costmap_2d_ros.cpp:63-167) and `DWAPlannerROS::initialize`
 * (dwa_local_planner/src/dwa_planner_ros.cpp:94-129), for n_instances robots at once. */
int navgpu_fleet_create(const navgpu_fleet_desc* desc, navgpu_fleet** out);
int navgpu_fleet_destroy(navgpu_fleet* fleet);
int navgpu_sync(navgpu_fleet* fleet);
void* navgpu_stream(navgpu_fleet* fleet); /* the fleet's hipStream_t */
/* Fault injection for tests (the counterpart of the reference's ObstacleLayer::addStaticObservation test hook,
 * costmap_2d/plugins/obstacle_layer.cpp:450-464): device allocations of this fleet larger than max_bytes fail with
 * NAVGPU_ERR_HIP from now on; 0 = no limit (default).  Used to check that a failed reconfigure leaves the previous
 * configuration in force. */
int navgpu_fleet_set_alloc_limit(navgpu_fleet* fleet, uint64_t max_bytes);

/* Costmap2D origin per instance (costmap_2d.h origin_x_/origin_y_); origins_xy = count x {x,y}.
 * replaces: LayeredCostmap::resizeMap origin arguments (layered_costmap.cpp:67-77) */
int navgpu_fleet_set_origin(navgpu_fleet* fleet, uint32_t first, uint32_t count, const double* origins_xy);
/* Costmap2D::getOriginX/Y — with a rolling window the origins move every navgpu_costmap_stage */
int navgpu_fleet_get_origin(navgpu_fleet* fleet, uint32_t first, uint32_t count, double* origins_xy);

/* raw grid access.  host buffers are count x size_y x size_x elements of the grid's type.
 * replaces: Costmap2D::getCharMap() (costmap_2d.cpp:187-190), VoxelGrid::getData() */
int navgpu_grid_upload(navgpu_fleet* fleet, int grid, uint32_t first, uint32_t count, const void* host);
int navgpu_grid_download(navgpu_fleet* fleet, int grid, uint32_t first, uint32_t count, void* host);
int navgpu_grid_device(navgpu_fleet* fleet, int grid, void** device_ptr, size_t* instance_stride_bytes);
/* replaces: Costmap2DPublisher::prepareGrid / the OccupancyGridUpdate window of publishCostmap
 * (costmap_2d/src/costmap_2d_publisher.cpp:57-74,103-115,146-156): cells [x0, xn) x [y0, yn) of the master grid
 * of one instance through the publisher's cost translation table (0, 1..98, 99, 100, -1), row-major into `out`
 * ((xn - x0) * (yn - y0) int8).  The message origin is the costmap origin (navgpu_fleet_get_origin). */
int navgpu_costmap_export(navgpu_fleet* fleet, uint32_t instance, uint32_t x0, uint32_t y0, uint32_t xn, uint32_t yn, int8_t* out);
/* Costmap2D::resetMaps() on the given grid (default value of that grid) */
int navgpu_grid_reset(navgpu_fleet* fleet, int grid, uint32_t first, uint32_t count);
/* replaces: Costmap2D::resetMap(x0, y0, xn, yn) (costmap_2d.cpp:93-99: rows [y0, yn), columns [x0, xn)) on
 * NAVGPU_GRID_MASTER or NAVGPU_GRID_OBSTACLE (a voxel layer's 2-D grid; its columns are untouched, as in the reference) */
int navgpu_grid_reset_window(navgpu_fleet* fleet, int grid, uint32_t first, uint32_t count, uint32_t x0, uint32_t y0, uint32_t xn, uint32_t yn);
/* replaces: CostmapLayer::resetBoundingBox(min, max) (costmap_layer.cpp:30-43; what Costmap2DROS::resetBoundingBox calls on
 * every CostmapLayer, costmap_2d_ros.cpp:574-611) on the obstacle / voxel layer: boxes = count x {min_x, min_y, max_x,
 * max_y} in world coordinates; the layer grid is reset inside (worldToMapEnforceBounds of both corners, resetMap) and the
 * box joins the bounds of the next navgpu_costmap_update / navgpu_obstacle_update_bounds (addExtraBounds / useExtraBounds). */
int navgpu_layer_reset_bounding_box(navgpu_fleet* fleet, uint32_t first, uint32_t count, const double* boxes);

/* ------------------------------------------------------------------------------------------ */
/* costmap layers                                                                             */
/* ------------------------------------------------------------------------------------------ */
/* replaces: StaticLayer::incomingMap (plugins/static_layer.cpp:167-228): occupancy is the
 * nav_msgs/OccupancyGrid int8 data of one map, interpreted with interpretValue (:149-163) on the
 * device and broadcast to [first, first+count). */
int navgpu_static_set_map(navgpu_fleet* fleet, uint32_t first, uint32_t count, const int8_t* occupancy,
                          int32_t track_unknown_space, int32_t use_maximum, int32_t trinary_costmap,
                          int32_t lethal_cost_threshold, int32_t unknown_cost_value);

/* replaces: StaticLayer under a rolling window (plugins/static_layer.cpp:187-193 incomingMap resizes the layer only;
 * :262-283 updateBounds adds the layer's extent every cycle; :300-333 updateCosts maps every master cell of the update
 * window to world, through the map_frame <- global_frame transform, into the static map).  One static map (its own
 * size, resolution and origin) is shared by all robots of a rolling_window fleet created with NAVGPU_LAYER_STATIC. */
int navgpu_static_set_rolling_map(navgpu_fleet* fleet, const int8_t* occupancy, uint32_t size_x, uint32_t size_y, double resolution,
                                  double origin_x, double origin_y, int32_t track_unknown_space, int32_t use_maximum,
                                  int32_t trinary_costmap, int32_t lethal_cost_threshold, int32_t unknown_cost_value);
/* replaces: tf_->lookupTransform(map_frame_, global_frame_, ...) (static_layer.cpp:311): per robot 12 doubles, the
 * tf::Transform's 3x3 basis row-major followed by its origin.  Identity until set. */
int navgpu_static_set_transform(navgpu_fleet* fleet, uint32_t first, uint32_t count, const double* basis_origin);

/* replaces: ObstacleLayer::reconfigureCB / VoxelLayer::reconfigureCB */
int navgpu_obstacle_configure(navgpu_fleet* fleet, const navgpu_obstacle_params* params);
/* replaces: InflationLayer::setInflationParameters + onFootprintChanged + computeCaches
 * (plugins/inflation_layer.cpp:160-170,295-328,362-376).  The (R+2)^2 distance/cost tables are
 * built on the host in fp64 with libm exactly as the reference does, then uploaded. */
int navgpu_inflation_configure(navgpu_fleet* fleet, const navgpu_inflation_params* params);
/* footprint helpers, pure host functions (no fleet, no GPU).  xy = n x {x, y} in the robot frame.
 * replaces: costmap_2d::calculateMinAndMaxDistances (costmap_2d/src/footprint.cpp:41-67; DBL_MAX / 0 for n <= 2),
 * padFootprint (:138-147, in place), makeFootprintFromRadius (:150-167, 16 vertices). */
int navgpu_footprint_radii(const double* xy, uint32_t n, double* inscribed_radius, double* circumscribed_radius);
int navgpu_footprint_pad(double* xy, uint32_t n, double padding);
int navgpu_footprint_from_radius(double radius, double* xy16);
/* replaces: LayeredCostmap::setFootprint (layered_costmap.cpp:164-174) for [first,first+count).
 * footprint_xy = n_vertices x {x,y} in the robot frame.  Does NOT change inscribed_radius of the
 * inflation layer (pass it through navgpu_inflation_configure, as the adapter does). */
int navgpu_set_footprint(navgpu_fleet* fleet, uint32_t first, uint32_t count, const double* footprint_xy,
                         uint32_t n_vertices);

/* H2D staging of one update cycle's observations (host -> HBM).  poses = count x {x,y,yaw}.
 * replaces: ObstacleLayer::getMarkingObservations / getClearingObservations (:466-496) */
int navgpu_costmap_stage(navgpu_fleet* fleet, uint32_t first, uint32_t count, const double* robot_poses,
                         const navgpu_observation* observations, uint32_t n_observations,
                         const float* points_xyz, uint32_t n_points_total);
/* replaces: LayeredCostmap::updateMap(robot_x, robot_y, robot_yaw) (layered_costmap.cpp:79-150):
 * updateBounds of every layer (raytrace clearing, marking, footprint touch, inflation box union),
 * window reset, updateCosts of every layer — all on the device, boxes never visit the host. */
int navgpu_costmap_update(navgpu_fleet* fleet, uint32_t first, uint32_t count);
/* boxes = count x {x0, xn, y0, yn}: LayeredCostmap::getBounds (layered_costmap.h:131-137) */
int navgpu_costmap_bounds(navgpu_fleet* fleet, uint32_t first, uint32_t count, int32_t* boxes);

/* layer-granular calls for the costmap_2d::Layer adapters (layer.h:50-130).  boxes = count x
 * {min_i, min_j, max_i, max_j} as handed to Layer::updateCosts, or NULL to use the boxes the
 * last navgpu_costmap_update computed on the device.
 * replaces: InflationLayer::updateCosts (plugins/inflation_layer.cpp:172-266) */
int navgpu_inflate(navgpu_fleet* fleet, uint32_t first, uint32_t count, const int32_t* boxes);
/* replaces: ObstacleLayer::updateBounds / VoxelLayer::updateBounds on the staged observations;
 * bounds_inout = count x {min_x, min_y, max_x, max_y} */
int navgpu_obstacle_update_bounds(navgpu_fleet* fleet, uint32_t first, uint32_t count, double* bounds_inout);
/* replaces: ObstacleLayer::updateCosts (plugins/obstacle_layer.cpp:427-448): updateWithOverwrite /
 * updateWithMax (costmap_layer.cpp:62-124) of the layer grid into the master grid AS IT STANDS — what
 * the layers before this one wrote (upload it with navgpu_grid_upload) is kept; no window reset and no
 * static merge here, those belong to navgpu_costmap_update's fused LayeredCostmap::updateMap. */
int navgpu_obstacle_update_costs(navgpu_fleet* fleet, uint32_t first, uint32_t count, const int32_t* boxes);

/* ------------------------------------------------------------------------------------------ */
/* DWA local planner                                                                          */
/* ------------------------------------------------------------------------------------------ */
/* replaces: DWAPlanner::reconfigure (dwa_planner.cpp:52-116) */
int navgpu_planner_configure(navgpu_fleet* fleet, const navgpu_dwa_config* config);
/* replaces: DWAPlanner::setPlan (dwa_planner.cpp:204-207): resets the oscillation flags */
int navgpu_planner_set_plan(navgpu_fleet* fleet, uint32_t first, uint32_t count);
/* H2D staging of one control cycle (host -> HBM): robot states and the packed local plans
 * (plan_xy = n_plan_total x {x,y}).  Also performs DWAPlanner::updatePlanAndLocalCosts
 * (dwa_planner.cpp:240-286): nose goal and alignment on/off are evaluated here on the host in
 * fp64 libm, the same arithmetic the reference runs once per cycle. */
int navgpu_planner_stage(navgpu_fleet* fleet, uint32_t first, uint32_t count, const navgpu_robot_state* states,
                         const double* plan_xy, uint32_t n_plan_total);
/* A control cycle whose local plan has not changed since the last navgpu_planner_stage (move_base hands a new plan
 * at planner_frequency, the pose changes at controller_frequency): stages pose and velocity only, 24 B per robot
 * (pos_xyth, vel_xyth = count x 3 floats), and re-derives the nose goal / alignment switch of
 * DWAPlanner::updatePlanAndLocalCosts (dwa_planner.cpp:254-285) from the resident plan.  Needs a staged plan. */
int navgpu_planner_stage_poses(navgpu_fleet* fleet, uint32_t first, uint32_t count, const float* pos_xyth, const float* vel_xyth);
/* replaces: DWAPlanner::findBestPath (dwa_planner.cpp:292-371) =
 * SimpleTrajectoryGenerator::initialise + SimpleScoredSamplingPlanner::findBestTrajectory
 * (4 x MapGridCostFunction::prepare, rollout + six critics per sample, first-strict-minimum) +
 * OscillationCostFunction::updateOscillationFlags. */
int navgpu_planner_cycle(navgpu_fleet* fleet, uint32_t first, uint32_t count);
/* Bounded MapGrid wavefronts (no counterpart in the reference, which always runs computeTargetDistance over the whole
 * costmap, map_grid.cpp:262-310).  The critics read a MapGrid only at trajectory and forward points
 * (map_grid_cost_function.cpp:75-129), all inside a box around the robot of half edge
 * hypot(max |v_x|, max |v_y|) * sim_time + forward_point_distance; a level-synchronous wavefront has every cell it has
 * reached final, so the search of a cycle may stop once that box is settled.  enable = 1 (the default): it does, and the
 * grids of that cycle are exact inside the box and completed on demand before anything else reads them
 * (navgpu_grid_download / _device of the three MapGrids, navgpu_planner_cost_cloud, a checked trajectory that leaves
 * the box) — a request that comes after the costmap or the plan of that cycle has changed fails with
 * NAVGPU_ERR_STATE.  Robots within two such half edges of the end of their plan always get whole grids (the
 * stop-and-rotate controller keeps using them across cycles).  enable = 0: every cycle searches the whole grid, as
 * the reference does.  Planner results are identical either way. */
int navgpu_planner_set_bounded_map_grids(navgpu_fleet* fleet, int32_t enable);
/* replaces: the MapGridCostFunction constructor arguments DWAPlanner never passes (map_grid_cost_function.h:64-69,
 * map_grid_cost_function.cpp:42-53, 75-129): aggregationType (0 Last - what DWAPlanner's four critics use -, 1 Sum,
 * 2 Product) and yshift (metres, sideways) of one critic: 0 path_costs_, 1 goal_costs_, 2 goal_front_costs_,
 * 3 alignment_costs_.  xshift stays DWAPlanner's (forward_point_distance for 2 and 3).  With any option set the scoring
 * launches take a general per-point step (no screen, no heading tables) and the wavefronts cover the whole map. */
int navgpu_planner_set_map_grid_options(navgpu_fleet* fleet, int32_t critic, int32_t aggregation, double yshift);
/* introspection: the number of wavefront levels the last cycle ran for the path / goal / goal_front grid of each instance
 * (levels = count x 3).  A whole-grid search runs until nothing new is reached, a bounded one stops earlier. */
int navgpu_planner_wavefront_levels(navgpu_fleet* fleet, uint32_t first, uint32_t count, uint32_t* levels);
/* introspection: the cell box {x0, x1, y0, y1} (inclusive) the last cycle's bounded wavefronts settled per instance
 * (boxes = count x 4); the whole map for an instance whose grids were searched whole */
int navgpu_planner_wavefront_boxes(navgpu_fleet* fleet, uint32_t first, uint32_t count, int32_t* boxes);
int navgpu_planner_results(navgpu_fleet* fleet, uint32_t first, uint32_t count, navgpu_plan_result* results);
/* Two control cycles in flight on the fleet's stream (no counterpart in the reference, whose cycle is a blocking call:
 * move_base.cpp:947 runs computeVelocityCommands to completion; an option of this library for callers that drive many
 * robots).  cycles = 2: navgpu_costmap_stage / navgpu_planner_stage of cycle k + 1 wait only until the copies out of
 * their pinned mirrors have run (a marker behind them), not for the stream, so cycle k + 1 can be handed over and queued
 * while cycle k runs; navgpu_planner_cycle then writes its results into the slot the cycle before it did not use, and
 * navgpu_planner_results_previous returns the results of the cycle BEFORE the latest queued one as soon as that cycle
 * has finished (NAVGPU_ERR_STATE when there is none).  navgpu_planner_results keeps its meaning: the latest queued cycle,
 * after the stream has drained.  With cycles = 2 a cycle covers the whole fleet (NAVGPU_ERR_STATE for a sub-range: the
 * result slot alternates per call).  cycles = 1 (default): every call as before.  The call itself drains the stream. */
int navgpu_planner_set_cycles_in_flight(navgpu_fleet* fleet, int32_t cycles);
int navgpu_planner_results_previous(navgpu_fleet* fleet, uint32_t first, uint32_t count, navgpu_plan_result* results);
/* winning trajectory of one instance: xyth = n_points x {x,y,theta}; returns n_points or <0 */
int navgpu_planner_trajectory(navgpu_fleet* fleet, uint32_t instance, double* xyth, uint32_t capacity_points);
/* every sample slot of one instance (needs keep_sample_costs): total cost with all critics summed
 * (no early-out; negative = the first failing critic's code) and NAVGPU_SAMPLE_* status */
int navgpu_planner_samples(navgpu_fleet* fleet, uint32_t instance, double* costs, int32_t* status,
                           float* velocities_xyz, uint32_t capacity);
/* replaces: DWAPlanner::checkTrajectory (dwa_planner.cpp:213-237) for one instance; uses the
 * staged state of that instance.  *ok = 1 when the single sample scores >= 0. */
int navgpu_planner_check_trajectory(navgpu_fleet* fleet, uint32_t instance, const float vel_samples[3], int32_t* ok);
/* replaces: DWAPlanner::getCellCosts (dwa_planner.cpp:185-202) over the whole map + MapGridVisualizer::publishCostCloud
 * (base_local_planner/src/map_grid_visualizer.cpp:55-83): the cost cloud of one instance from the grids of its last
 * cycle.  points = up to `capacity` x {x, y, z, path_cost, goal_cost, occ_cost, total_cost} (MapGridCostPoint), in the
 * reference's order (cx outer, cy inner, cells for which getCellCosts returns false skipped).  Returns the
 * number of points of the full cloud. */
int navgpu_planner_cost_cloud(navgpu_fleet* fleet, uint32_t instance, float* points, uint32_t capacity);
/* OscillationCostFunction state access (persists across cycles per instance) */
int navgpu_planner_get_oscillation(navgpu_fleet* fleet, uint32_t first, uint32_t count, uint32_t* flags,
                                   float* prev_stationary_pos_xyz);
int navgpu_planner_set_oscillation(navgpu_fleet* fleet, uint32_t first, uint32_t count, const uint32_t* flags,
                                   const float* prev_stationary_pos_xyz);

/* ------------------------------------------------------------------------------------------ */
/* DWAPlannerROS control cycle: the steps around findBestPath         */
/* ------------------------------------------------------------------------------------------ */
/* Host-side mirror, free of ROS types, of DWAPlannerROS::setPlan / computeVelocityCommands /
 * isGoalReached (dwa_local_planner/src/dwa_planner_ros.cpp:130-158,176-300), of
 * LocalPlannerUtil::getLocalPlan (base_local_planner/src/local_planner_util.cpp:105-123) with
 * goal_functions.cpp's transformGlobalPlan / prunePlan / getGoalPose / stopped (:69-174,175-255) and of
 * LatchedStopRotateController (src/latched_stop_rotate_controller.cpp:37-273).  Poses are (x, y, yaw)
 * triples; the tf lookup is replaced by an optional planar plan->global transform handed in by the
 * caller (identity when NULL).  tf's own 3-D arithmetic and the `angles` package are not part of the
 * reference tree: their formulas are restated (angles::normalize_angle: fmod form), parity unpinned. */
typedef struct {
  double xy_goal_tolerance, yaw_goal_tolerance;  /* LocalPlannerLimits (local_planner_limits.h)            */
  double rot_stopped_vel, trans_stopped_vel;
  double max_rot_vel, min_rot_vel;
  double acc_lim_x, acc_lim_y, acc_lim_theta;    /* limits.getAccLimits()                                   */
  double sim_period;                             /* DWAPlanner::getSimPeriod()                              */
  int32_t prune_plan;                            /* LocalPlannerLimits::prune_plan                          */
  int32_t latch_xy_goal_tolerance;               /* ~/latch_xy_goal_tolerance                               */
} navgpu_local_limits;

typedef struct {
  double pose[3];      /* costmap_ros_->getRobotPose(): x, y, yaw in the costmap's global frame             */
  double odom_vel[3];  /* OdometryHelperRos: twist.linear.x, twist.linear.y, twist.angular.z               */
  int32_t have_pose;   /* 0: getRobotPose failed -> computeVelocityCommands returns false                   */
  int32_t reserved;
} navgpu_robot_input;

typedef enum {
  NAVGPU_BRANCH_NONE = 0,     /* returned before dispatching (no pose, no plan, empty local plan)           */
  NAVGPU_BRANCH_DWA = 1,      /* dwaComputeVelocityCommands                                                 */
  NAVGPU_BRANCH_STOP = 2,     /* stop-rotate: stopWithAccLimits                                             */
  NAVGPU_BRANCH_ROTATE = 3,   /* stop-rotate: rotateToGoal                                                  */
  NAVGPU_BRANCH_AT_GOAL = 4   /* stop-rotate: goal orientation reached, zero command                        */
} navgpu_branch;

typedef struct {
  double cmd_vel[3];          /* geometry_msgs::Twist linear.x, linear.y, angular.z                         */
  int32_t ok;                 /* return value of computeVelocityCommands                                    */
  int32_t branch;             /* navgpu_branch                                                              */
  int32_t local_plan_points;  /* poses of the transformed + pruned plan handed to updatePlanAndLocalCosts   */
  int32_t trajectory_points;  /* points of the published local plan (winning trajectory; 0 unless DWA ok)   */
} navgpu_cmd_result;

/* transformGlobalPlan (goal_functions.cpp:88-174) followed, when `prune` is set, by prunePlan
 * (:69-86) on a plan of n (x, y, yaw) triples.  dist_threshold = max(size_x, size_y) * resolution / 2
 * (:119-120).  Writes the local plan to out_xyyaw (capacity poses), its length to *n_out and the number
 * of poses prunePlan erased from the FRONT OF THE STORED GLOBAL PLAN to *n_erased (it erases both plans in
 * lockstep from their beginnings).  Pure host function: needs no fleet and no GPU.
 * Returns NAVGPU_OK, NAVGPU_ERR_INVALID (n == 0: "Received plan with zero length") or NAVGPU_ERR_CAPACITY. */
int navgpu_local_plan_window(const double* plan_xyyaw, uint32_t n, const double pose[3], const double* plan_to_global,
                             double dist_threshold, int32_t prune, double* out_xyyaw, uint32_t capacity,
                             uint32_t* n_out, uint32_t* n_erased);
/* angles::shortest_angular_distance(from, to) as restated here (exposed for the parity tests) */
double navgpu_shortest_angular_distance(double from, double to);

/* LocalPlannerUtil::reconfigureCB limits + LatchedStopRotateController parameters, for the whole fleet */
int navgpu_local_planner_configure(navgpu_fleet* fleet, const navgpu_local_limits* limits);
/* DWAPlannerROS::setPlan: stores the global plan of one instance (n (x, y, yaw) triples in the plan's own
 * frame), clears the goal-tolerance latch and resets the oscillation flags.  plan_to_global = NULL: the
 * plan is already expressed in the costmap's global frame. */
int navgpu_local_planner_set_plan(navgpu_fleet* fleet, uint32_t instance, const double* plan_xyyaw, uint32_t n,
                                  const double* plan_to_global);
/* DWAPlannerROS::computeVelocityCommands for instances [first, first+count): getLocalPlan ->
 * updatePlanAndLocalCosts -> isPositionReached ? computeVelocityCommandsStopRotate (checkTrajectory on
 * the GPU as the obstacle check, MapGrids NOT refreshed: the reference does not call prepare() there)
 * : dwaComputeVelocityCommands (navgpu_planner_cycle over the instances on that branch). */
int navgpu_local_planner_compute_velocity_commands(navgpu_fleet* fleet, uint32_t first, uint32_t count,
                                                   const navgpu_robot_input* in, navgpu_cmd_result* out);
/* DWAPlannerROS::isGoalReached -> LatchedStopRotateController::isGoalReached */
int navgpu_local_planner_is_goal_reached(navgpu_fleet* fleet, uint32_t first, uint32_t count,
                                         const navgpu_robot_input* in, int32_t* reached);
/* the stored global plan of one instance after pruning; returns its length (poses) or < 0 */
int navgpu_local_planner_get_plan(navgpu_fleet* fleet, uint32_t instance, double* xyyaw, uint32_t capacity);

/* ------------------------------------------------------------------------------------------ */
/* Legacy base_local_planner::TrajectoryPlanner ("Trajectory Rollout")              */
/* ------------------------------------------------------------------------------------------ */
/* The second nav_core::BaseLocalPlanner of the reference (TrajectoryPlannerROS).  Same structure as the
 * DWA path - two MapGrid wavefronts (path_map_ with the cells under the robot's own footprint marked
 * within_robot, goal_map_), a rollout per velocity sample with footprint and grid look-ups - but fp64
 * state with acceleration-limited velocities, a different sample enumeration and a sequential, stateful
 * selection (in-place rotation / strafing / backing up with oscillation and escape flags).  The GPU rolls
 * out every candidate sample of createTrajectories; the host replays the reference's selection over the
 * per-sample results.  heading_scoring (headingDiff's line-of-sight scan over the plan, :372-386) and simple_attractor
 * (:310-315) are options of the same rollout. */
typedef struct {
  double acc_lim_x, acc_lim_y, acc_lim_theta;
  double sim_time, sim_granularity, angular_sim_granularity;
  double pdist_scale, gdist_scale, occdist_scale; /* after the meter_scoring multiplication, if any */
  double heading_lookahead, oscillation_reset_dist, escape_reset_dist, escape_reset_theta;
  double max_vel_x, min_vel_x, max_vel_th, min_vel_th, min_in_place_vel_th;
  double backup_vel;                              /* escape_vel */
  double sim_period;
  double heading_scoring_timestep;                /* BaseLocalPlanner.cfg: 0.1 (TrajectoryPlannerROS's own param default: 0.8) */
  double y_vels[8];
  int32_t n_y_vels;
  int32_t vx_samples, vtheta_samples;
  int32_t holonomic_robot, dwa, allow_unknown;
  int32_t heading_scoring, simple_attractor;
} navgpu_tp_config;

/* TrajectoryPlanner members that persist between cycles (trajectory_planner.h:290-300) */
#define NAVGPU_TP_STUCK_LEFT (1u << 0)
#define NAVGPU_TP_STUCK_RIGHT (1u << 1)
#define NAVGPU_TP_ROTATING_LEFT (1u << 2)
#define NAVGPU_TP_ROTATING_RIGHT (1u << 3)
#define NAVGPU_TP_STUCK_LEFT_STRAFE (1u << 4)
#define NAVGPU_TP_STUCK_RIGHT_STRAFE (1u << 5)
#define NAVGPU_TP_STRAFE_LEFT (1u << 6)
#define NAVGPU_TP_STRAFE_RIGHT (1u << 7)
#define NAVGPU_TP_ESCAPING (1u << 8)
typedef struct {
  uint32_t flags;
  uint32_t reserved;
  double prev_x, prev_y, escape_x, escape_y, escape_theta;
} navgpu_tp_state;

typedef struct {
  double xv, yv, thetav, cost;   /* the returned Trajectory                                             */
  double drive[3];               /* drive_velocities (zeros when cost < 0)                              */
  int32_t n_points;              /* points of that trajectory                                           */
  int32_t n_samples;             /* generateTrajectory calls the reference would have made this cycle   */
  int32_t best_sample;           /* index of the winner among them (call order)                         */
  int32_t reserved;
} navgpu_tp_result;

/* one generateTrajectory call as the reference makes it (call order) */
typedef struct {
  double vx, vy, vtheta;         /* the sample                                                          */
  double cost;                   /* traj.cost_: >= 0, -1 (off map / collision) or -2 (no path to goal)  */
  int32_t n_points;
  int32_t reserved;
} navgpu_tp_sample;

/* replaces: TrajectoryPlanner::TrajectoryPlanner / reconfigure (trajectory_planner.cpp:58-172); footprint
 * from navgpu_set_footprint.  Also sizes the per-robot sample buffers. */
int navgpu_tp_configure(navgpu_fleet* fleet, const navgpu_tp_config* config);
/* replaces: TrajectoryPlanner::updatePlan(new_plan, compute_dists) (:474-500) for one instance; plan_xy =
 * n x {x, y} in the costmap's global frame. */
int navgpu_tp_update_plan(navgpu_fleet* fleet, uint32_t instance, const double* plan_xy, uint32_t n, int32_t compute_dists);
/* replaces: TrajectoryPlanner::findBestPath (:908-984) for instances [first, first+count): pos / vel of
 * `states` are the Eigen::Vector3f the reference builds (plan fields ignored). */
int navgpu_tp_find_best_path(navgpu_fleet* fleet, uint32_t first, uint32_t count, const navgpu_robot_state* states,
                             navgpu_tp_result* results);
/* the winning trajectory's points (x, y, theta); returns n_points or < 0 */
int navgpu_tp_trajectory(navgpu_fleet* fleet, uint32_t instance, double* xyth, uint32_t capacity_points);
/* the generateTrajectory calls of the last cycle of one instance, in the reference's call order */
int navgpu_tp_samples(navgpu_fleet* fleet, uint32_t instance, navgpu_tp_sample* samples, uint32_t capacity);
/* replaces: TrajectoryPlanner::scoreTrajectory / checkTrajectory (:502-531) against the current grids */
int navgpu_tp_score_trajectory(navgpu_fleet* fleet, uint32_t instance, const double pose[3], const double vel[3],
                               const double vel_samples[3], double* cost);
int navgpu_tp_get_state(navgpu_fleet* fleet, uint32_t first, uint32_t count, navgpu_tp_state* states);
int navgpu_tp_set_state(navgpu_fleet* fleet, uint32_t first, uint32_t count, const navgpu_tp_state* states);

/* ------------------------------------------------------------------------------------------ */
/* measurement                                                                                */
/* ------------------------------------------------------------------------------------------ */
typedef enum {
  NAVGPU_K_OBSTACLE = 0, /* raytrace + mark + bounds                      */
  NAVGPU_K_MERGE = 1,    /* window reset + static/obstacle merge          */
  NAVGPU_K_INFLATE = 2,  /* inflation                                     */
  NAVGPU_K_BFS = 3,      /* MapGrid wavefronts                            */
  NAVGPU_K_SCORE = 4,    /* rollout + critics                             */
  NAVGPU_K_SELECT = 5,   /* argmin + result + oscillation update          */
  NAVGPU_K_COUNT = 6
} navgpu_kernel_id;
/* HIP-event timing of the kernels on the fleet's stream.  While enabled every launch of the
 * listed kernels is bracketed by two hipEventRecord calls; read() synchronises and returns the
 * accumulated device time (ms) and launch count since the last reset. */
int navgpu_profile_enable(navgpu_fleet* fleet, int32_t enable);
/* restrict the bracketing to some kernels: bit k = navgpu_kernel_id k (default: all).  Every pair of events costs the
 * stream a few microseconds, so a throughput measurement selects the kernel it reports on. */
int navgpu_profile_select(navgpu_fleet* fleet, uint32_t kernel_mask);
int navgpu_profile_reset(navgpu_fleet* fleet);
int navgpu_profile_read(navgpu_fleet* fleet, int32_t kernel, double* total_ms, uint64_t* launches);
const char* navgpu_kernel_name(int32_t kernel);

/* The floating-point contract, checkable: sin / cos of `n` headings exactly as the scoring kernels evaluate them on `device`
 * (double `sincos`, the reference's `cos(theta)` / `sin(theta)` of simple_trajectory_generator.cpp:253-258 on the host's
 * libm).  Host pointers.  tests/ compare the result with the host's libm bit for bit and bound the difference. */
int navgpu_device_sincos(int32_t device, const double* theta, uint32_t n, double* sin_out, double* cos_out);

/* ------------------------------------------------------------------------------------------ */
/* navfn::NavFn - global-planner potential expansion and path extraction       */
/* ------------------------------------------------------------------------------------------ */
/* A batch of independent plans on maps of one size (one plan per robot of a fleet).  What is computed is exactly what
 * navfn::NavFn computes - potentials, priority-buffer order, the early stop at the start cell, the interpolated path -
 * bit for bit (the expansion is an order-dependent sequential process: one GPU lane walks each plan, the batch is the
 * parallel dimension).  Coordinates are cells, origin upper left, as in the reference (navfn.h:108-112). */
typedef struct navgpu_navfn navgpu_navfn;
typedef struct {
  int32_t found;          /* calcNavFnDijkstra / calcNavFnAstar return value (navfn.cpp:293-345)           */
  int32_t path_length;    /* NavFn::getPathLen(), 0 when no path was found                                 */
  int32_t cycles;         /* propagation cycles used                                                       */
  float start_potential;  /* potarr[start] (NavFn::getLastPathCost after calcNavFnAstar)                   */
} navgpu_navfn_result;
/* replaces: NavFn::NavFn / setNavArr (navfn.cpp:110-215) for n_plans plans */
int navgpu_navfn_create(uint32_t nx, uint32_t ny, uint32_t n_plans, int32_t device, navgpu_navfn** out);
int navgpu_navfn_destroy(navgpu_navfn* nav);
/* replaces: NavFn::setCostmap(cmap, isROS, allow_unknown) (navfn.cpp:222-283).  cmap = count x ny x nx bytes (or ONE map shared
 * by all plans when shared != 0).  cost_mode 1: isROS = true (costmap_2d values), 2: isROS = false (a plain PGM, 7-cell
 * borders stay obstacles), 0: the bytes are costarr itself (navfn/test/path_calc_test.cpp:52) */
int navgpu_navfn_set_costmap(navgpu_navfn* nav, uint32_t first, uint32_t count, const uint8_t* cmap, int32_t shared, int32_t cost_mode,
                             int32_t allow_unknown);
/* the same from the master grids of a fleet on the same GPU (NavfnROS::makePlan hands costmap_->getCharMap() over, navfn_ros.cpp:265-268):
 * plan first + k takes the master grid of fleet instance fleet_first + k; nothing crosses PCIe */
int navgpu_navfn_set_costmap_from_fleet(navgpu_navfn* nav, uint32_t first, uint32_t count, navgpu_fleet* fleet, uint32_t fleet_first,
                                        int32_t allow_unknown);
/* replaces: NavFn::setGoal / setStart + calcNavFnDijkstra(at_start) | calcNavFnAstar() (navfn.cpp:145-171, 293-345).
 * goals_xy, starts_xy = count x {x, y} cells.  (NavfnROS passes the robot as "goal" and the goal as "start",
 * navfn_ros.cpp:270-281: the potential is grown from the robot.) */
int navgpu_navfn_plan(navgpu_navfn* nav, uint32_t first, uint32_t count, const int32_t* goals_xy, const int32_t* starts_xy, int32_t astar,
                      int32_t at_start, navgpu_navfn_result* results);
/* The same call as a device algorithm (no counterpart in the reference): NavFn::updateCell's update rule (navfn.cpp:466-535,
 * same float / double arithmetic) relaxed to its fixed point by 32 x 32 tiles in LDS, round by round, instead of walked
 * through the three priority buffers on one lane; at_start stops once the start cell and everything below its potential
 * is final (the counterpart of :692-694).  The potentials are those the reference's process converges to where it is
 * allowed to finish: <= the reference's everywhere (its 10 000-entry buffers drop cells, its push tests skip updates, its
 * early stop leaves the last block half done), equal along most of the path; calcPath (the same code) then gives a path
 * within a fraction of a cell of the reference's (tests/test_navfn.py: Hausdorff distance <= 1 cell on the reference's
 * willow_costmap searches).  Results do not depend on scheduling (rounds are Jacobi across tiles, tiles are swept
 * red / black).  results[k].cycles = rounds run.  Dijkstra only.  The bit-exact mode stays navgpu_navfn_plan. */
int navgpu_navfn_plan_wavefront(navgpu_navfn* nav, uint32_t first, uint32_t count, const int32_t* goals_xy, const int32_t* starts_xy,
                                int32_t at_start, navgpu_navfn_result* results);
/* replaces: NavFn::getPathX / getPathY / getPathLen: xy = up to capacity_points x {x, y}; returns the path length */
int navgpu_navfn_path(navgpu_navfn* nav, uint32_t plan, float* xy, uint32_t capacity_points);
/* NavFn::potarr of one plan (ny x nx floats, POT_HIGH = 1e10 where unassigned) */
int navgpu_navfn_potential(navgpu_navfn* nav, uint32_t plan, float* potarr);

/* global_planner::GlobalPlanner's expansion and traceback on the same arrays (the other half of SURVEY 8 f-4).
 * Parameters as planner_core.cpp:105-152 reads them and GlobalPlanner.cfg sets them. */
typedef struct {
  int32_t use_dijkstra;       /* 1: DijkstraExpansion (dijkstra.cpp), 0: AStarExpansion (astar.cpp)                          */
  int32_t use_quadratic;      /* 1: QuadraticCalculator, 0: PotentialCalculator                                              */
  int32_t use_grid_path;      /* 1: GridPath, 0: GradientPath                                                                */
  int32_t old_navfn_behavior; /* 1: integer start / goal, no precise start, no clearEndpoint (planner_core.cpp:108-127,299)  */
  int32_t allow_unknown;      /* Expander::setHasUnknown                                                                     */
  int32_t lethal_cost, neutral_cost; /* GlobalPlanner.cfg: 253, 50                                                           */
  float cost_factor;          /* GlobalPlanner.cfg: 3.0                                                                      */
  int32_t outline_map;        /* 1: GlobalPlanner::outlineMap(costs, nx, ny, LETHAL_OBSTACLE) first, as makePlan does (:296).
                               * With 0, or with lethal_cost 255 under A*, border cells can be expanded; the reference then
                               * reads potential[] / costs[] one row outside its arrays.  Here such neighbours read as
                               * unreached lethal cells (never outside device memory); results on maps whose expansion
                               * stays off the border are unchanged                                                          */
  int32_t reserved;
} navgpu_global_planner_params;
/* replaces: the body of GlobalPlanner::makePlan between worldToMap and the plan assembly (planner_core.cpp:250-311):
 * outlineMap, planner_->calculatePotentials(costs, start, goal, nx * ny * 2, potential), clearEndpoint, path_maker_->getPath.
 * The cost bytes are those of navgpu_navfn_set_costmap with cost_mode 0 (the costmap itself: getCost translates on the fly).
 * starts_xy / goals_xy = count x {x, y} MAP coordinates as makePlan computes them (the cell index, or (w - origin) / resolution
 * - 0.5 without old_navfn_behavior); goal_cells_xy = count x {goal_x_i, goal_y_i}.  navgpu_navfn_path then returns the
 * traceback's own point list (goal first; getPlanFromPotential reverses it), navgpu_navfn_potential the potential array.
 * Limits: starts closer than 2 cells and goals closer than 1 cell to the map border are rejected (the reference reads
 * outside its arrays there); a traceback longer than max(nx * ny / 2, 4 nx) + 4 points - the reference allows 4 nx ny -
 * reports found = 0. */
int navgpu_global_planner_plan(navgpu_navfn* nav, uint32_t first, uint32_t count, const navgpu_global_planner_params* params,
                               const double* starts_xy, const double* goals_xy, const int32_t* goal_cells_xy, navgpu_navfn_result* results);
/* The same call with DijkstraExpansion run as a device algorithm (see navgpu_navfn_plan_wavefront: the update rule of
 * dijkstra.cpp:170-229 - getCost, PotentialCalculator or QuadraticCalculator - relaxed to its fixed point by LDS tiles from the
 * start cell(s), stopped once the goal cell and everything below its potential is final); clearEndpoint and the traceback are
 * the same code as in navgpu_global_planner_plan.  use_dijkstra must be 1.  results[k].cycles = rounds run. */
int navgpu_global_planner_plan_wavefront(navgpu_navfn* nav, uint32_t first, uint32_t count, const navgpu_global_planner_params* params,
                                         const double* starts_xy, const double* goals_xy, const int32_t* goal_cells_xy, navgpu_navfn_result* results);

#ifdef __cplusplus
}
#endif
#endif /* NAVGPU_H_ */
