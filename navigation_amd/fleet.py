"""Fleet: thin Python handle over the navgpu C-ABI (one GPU, N robot instances).

Method names follow the reference interfaces they front:
  update_map            LayeredCostmap::updateMap            (costmap_2d/src/layered_costmap.cpp:79-150)
  add_static_map        StaticLayer::incomingMap             (plugins/static_layer.cpp:167-228)
  set_footprint         LayeredCostmap::setFootprint         (layered_costmap.cpp:164-174)
  set_plan              DWAPlanner::setPlan                  (dwa_local_planner/src/dwa_planner.cpp:204-207)
  find_best_path        DWAPlanner::updatePlanAndLocalCosts + findBestPath (dwa_planner.cpp:240-371)
  check_trajectory      DWAPlanner::checkTrajectory          (dwa_planner.cpp:213-237)
All compute happens in libnavgpu.so on the GPU; this file only marshals numpy buffers.
"""
import ctypes as C

import numpy as np

from . import _lib as N
from ._lib import (DwaConfig, FleetDesc, InflationParams, NavgpuError, Observation, ObstacleParams, PlanResult,
                   RobotState, check, lib)

_GRID_DTYPE = {N.GRID_MASTER: np.uint8, N.GRID_STATIC: np.uint8, N.GRID_OBSTACLE: np.uint8,
               N.GRID_VOXEL: np.uint32, N.GRID_PATH: np.uint32, N.GRID_GOAL: np.uint32,
               N.GRID_GOAL_FRONT: np.uint32}


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Fleet:
    def __init__(self, n_instances, size_x, size_y, resolution, layers=N.LAYER_OBSTACLE | N.LAYER_INFLATION,
                 track_unknown=False, device=0, max_points=1024, max_observations=4, max_plan=256,
                 max_footprint=16, max_sim_steps=64, keep_sample_costs=False, rolling_window=False):
        self.L = lib()
        d = FleetDesc(n_instances, size_x, size_y, resolution, layers, int(track_unknown), device, max_points,
                      max_observations, max_plan, max_footprint, max_sim_steps, int(keep_sample_costs), int(rolling_window))
        self.desc = d
        self.n, self.nx, self.ny, self.res = n_instances, size_x, size_y, resolution
        self.layers = layers
        h = C.c_void_p()
        check(self.L.navgpu_fleet_create(C.byref(d), C.byref(h)), "navgpu_fleet_create")
        self.h = h
        self.cfg = None
        self._footprint = None

    def close(self):
        if getattr(self, "h", None):
            self.L.navgpu_fleet_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _range(self, first, count):
        return first, (self.n - first if count is None else count)

    # ---------------------------------------------------------------- grids
    def sync(self):
        check(self.L.navgpu_sync(self.h), "navgpu_sync")

    def set_alloc_limit(self, max_bytes):
        """fault injection for tests: device allocations larger than max_bytes fail from now on (0 = no limit)"""
        check(self.L.navgpu_fleet_set_alloc_limit(self.h, int(max_bytes)), "fleet_set_alloc_limit")

    def origins(self):
        """Current Costmap2D origins (they move with a rolling window)."""
        o = np.zeros((self.n, 2), np.float64)
        check(self.L.navgpu_fleet_get_origin(self.h, 0, self.n, _ptr(o)), "get_origin")
        return o

    def set_origin(self, origins_xy, first=0):
        o = np.ascontiguousarray(origins_xy, np.float64).reshape(-1, 2)
        check(self.L.navgpu_fleet_set_origin(self.h, first, len(o), _ptr(o)), "set_origin")

    def upload(self, grid, cells, first=0):
        a = np.ascontiguousarray(cells, _GRID_DTYPE[grid]).reshape(-1, self.ny, self.nx)
        check(self.L.navgpu_grid_upload(self.h, grid, first, len(a), _ptr(a)), "grid_upload")

    def download(self, grid, first=0, count=None):
        first, count = self._range(first, count)
        out = np.empty((count, self.ny, self.nx), _GRID_DTYPE[grid])
        check(self.L.navgpu_grid_download(self.h, grid, first, count, _ptr(out)), "grid_download")
        return out

    def master(self, first=0, count=None):
        return self.download(N.GRID_MASTER, first, count)

    def export_occupancy(self, instance, x0=0, y0=0, xn=None, yn=None):
        """Costmap2DPublisher's int8 occupancy view of a window of the master grid."""
        xn = self.desc.size_x if xn is None else xn
        yn = self.desc.size_y if yn is None else yn
        out = np.zeros((yn - y0, xn - x0), np.int8)
        check(self.L.navgpu_costmap_export(self.h, instance, x0, y0, xn, yn, _ptr(out)), "costmap_export")
        return out

    def reset(self, grid, first=0, count=None):
        first, count = self._range(first, count)
        check(self.L.navgpu_grid_reset(self.h, grid, first, count), "grid_reset")

    def grid_device(self, grid):
        p, s = C.c_void_p(), C.c_size_t()
        check(self.L.navgpu_grid_device(self.h, grid, C.byref(p), C.byref(s)), "grid_device")
        return p.value, s.value

    def reset_window(self, grid, x0, y0, xn, yn, first=0, count=None):
        """Costmap2D::resetMap(x0, y0, xn, yn) on GRID_MASTER or GRID_OBSTACLE."""
        first, count = self._range(first, count)
        check(self.L.navgpu_grid_reset_window(self.h, grid, first, count, x0, y0, xn, yn), "grid_reset_window")

    def reset_bounding_box(self, boxes, first=0, count=None):
        """CostmapLayer::resetBoundingBox on the obstacle / voxel layer: boxes (count, 4) = min_x, min_y, max_x, max_y (world)."""
        first, count = self._range(first, count)
        b = np.ascontiguousarray(np.broadcast_to(np.asarray(boxes, np.float64).reshape(-1, 4), (count, 4)))
        check(self.L.navgpu_layer_reset_bounding_box(self.h, first, count, _ptr(b)), "layer_reset_bounding_box")

    # ---------------------------------------------------------------- layers
    def add_static_map(self, occupancy, first=0, count=None, track_unknown_space=True, use_maximum=False,
                       trinary_costmap=True, lethal_cost_threshold=100, unknown_cost_value=-1):
        first, count = self._range(first, count)
        occ = np.ascontiguousarray(occupancy, np.int8).reshape(self.ny, self.nx)
        check(self.L.navgpu_static_set_map(self.h, first, count, _ptr(occ), int(track_unknown_space), int(use_maximum),
                                           int(trinary_costmap), lethal_cost_threshold, unknown_cost_value),
              "static_set_map")

    def set_rolling_static_map(self, occupancy, resolution, origin_x, origin_y, track_unknown_space=True, use_maximum=False,
                               trinary_costmap=True, lethal_cost_threshold=100, unknown_cost_value=-1):
        """StaticLayer of a rolling-window fleet: one static map with its own geometry (static_layer.cpp:300-333)."""
        occ = np.ascontiguousarray(occupancy, np.int8)
        sy, sx = occ.shape
        check(self.L.navgpu_static_set_rolling_map(self.h, _ptr(occ), sx, sy, float(resolution), float(origin_x), float(origin_y),
                                                   int(track_unknown_space), int(use_maximum), int(trinary_costmap), lethal_cost_threshold,
                                                   unknown_cost_value), "static_set_rolling_map")

    def set_static_transform(self, basis, origin, first=0, count=None):
        """map_frame <- global_frame per robot: 3x3 basis (row-major) and origin, broadcast when a single one is given."""
        first, count = self._range(first, count)
        b = np.asarray(basis, np.float64).reshape(-1, 9)
        o = np.asarray(origin, np.float64).reshape(-1, 3)
        m = np.ascontiguousarray(np.broadcast_to(np.concatenate([b, o], 1), (count, 12)))
        check(self.L.navgpu_static_set_transform(self.h, first, count, _ptr(m)), "static_set_transform")

    def configure_obstacle(self, enabled=True, footprint_clearing_enabled=True, combination_method=1,
                           max_obstacle_height=2.0, z_voxels=10, origin_z=0.0, z_resolution=0.2,
                           unknown_threshold=15, mark_threshold=0):
        p = ObstacleParams(int(enabled), int(footprint_clearing_enabled), combination_method, z_voxels,
                           max_obstacle_height, origin_z, z_resolution, unknown_threshold, mark_threshold)
        check(self.L.navgpu_obstacle_configure(self.h, C.byref(p)), "obstacle_configure")

    def configure_inflation(self, inflation_radius, cost_scaling_factor, inscribed_radius, enabled=True, priority_queue_order=False):
        """priority_queue_order: reproduce InflationLayer::updateCosts' own priority-queue walk byte for byte (slow)."""
        p = InflationParams(int(enabled), int(priority_queue_order), inflation_radius, cost_scaling_factor, inscribed_radius)
        check(self.L.navgpu_inflation_configure(self.h, C.byref(p)), "inflation_configure")

    def set_footprint(self, xy, first=0, count=None):
        first, count = self._range(first, count)
        fp = np.ascontiguousarray(xy, np.float64).reshape(-1, 2)
        self._footprint = fp
        check(self.L.navgpu_set_footprint(self.h, first, count, _ptr(fp), len(fp)), "set_footprint")

    # ---------------------------------------------------------------- costmap cycle
    def stage_observations(self, poses, observations, first=0):
        """poses: (count,3).  observations: list of dicts {instance, points(k,3) float32, origin(3),
        obstacle_range, raytrace_range, marking, clearing}."""
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 3)
        count = len(poses)
        obs_arr = (Observation * max(1, len(observations)))()
        pts = []
        off = 0
        for k, o in enumerate(observations):
            p = np.ascontiguousarray(o["points"], np.float32).reshape(-1, 3)
            flags = (N.OBS_MARKING if o.get("marking", True) else 0) | (N.OBS_CLEARING if o.get("clearing", True) else 0)
            org = o.get("origin", (0.0, 0.0, 1.0))
            obs_arr[k] = Observation(o["instance"], off, len(p), flags, org[0], org[1], org[2],
                                     o.get("obstacle_range", 2.5), o.get("raytrace_range", 3.0))
            pts.append(p)
            off += len(p)
        allp = np.concatenate(pts) if pts else np.zeros((0, 3), np.float32)
        allp = np.ascontiguousarray(allp, np.float32)
        check(self.L.navgpu_costmap_stage(self.h, first, count, _ptr(poses), C.cast(obs_arr, C.c_void_p),
                                          len(observations), _ptr(allp) if len(allp) else None, len(allp)),
              "costmap_stage")

    def stage_observations_raw(self, poses, obs_array, n_obs, points, first=0):
        """Pre-built ctypes Observation array + packed float32 points (bench path, no Python loops)."""
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 3)
        check(self.L.navgpu_costmap_stage(self.h, first, len(poses), _ptr(poses), C.cast(obs_array, C.c_void_p), n_obs,
                                          _ptr(points), len(points)), "costmap_stage")

    def update_map(self, first=0, count=None):
        first, count = self._range(first, count)
        check(self.L.navgpu_costmap_update(self.h, first, count), "costmap_update")

    def bounds(self, first=0, count=None):
        first, count = self._range(first, count)
        b = np.zeros((count, 4), np.int32)
        check(self.L.navgpu_costmap_bounds(self.h, first, count, _ptr(b)), "costmap_bounds")
        return b

    def inflate(self, boxes=None, first=0, count=None):
        first, count = self._range(first, count)
        if boxes is None:
            check(self.L.navgpu_inflate(self.h, first, count, None), "inflate")
        else:
            b = np.ascontiguousarray(boxes, np.int32).reshape(count, 4)
            check(self.L.navgpu_inflate(self.h, first, count, _ptr(b)), "inflate")

    def obstacle_update_bounds(self, bounds, first=0):
        b = np.ascontiguousarray(bounds, np.float64).reshape(-1, 4).copy()
        check(self.L.navgpu_obstacle_update_bounds(self.h, first, len(b), _ptr(b)), "obstacle_update_bounds")
        return b

    def obstacle_update_costs(self, boxes, first=0):
        b = np.ascontiguousarray(boxes, np.int32).reshape(-1, 4)
        check(self.L.navgpu_obstacle_update_costs(self.h, first, len(b), _ptr(b)), "obstacle_update_costs")

    # ---------------------------------------------------------------- planner
    def configure_planner(self, cfg):
        self.cfg = cfg
        check(self.L.navgpu_planner_configure(self.h, C.byref(cfg)), "planner_configure")

    def set_plan(self, first=0, count=None):
        first, count = self._range(first, count)
        check(self.L.navgpu_planner_set_plan(self.h, first, count), "planner_set_plan")

    def stage_planner(self, pos, vel, plans, first=0):
        """pos, vel: (count,3) float32.  plans: (count, K, 2) array or list of (k_i,2) arrays."""
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        vel = np.ascontiguousarray(vel, np.float32).reshape(-1, 3)
        count = len(pos)
        if isinstance(plans, np.ndarray) and plans.ndim == 3:
            packed = np.ascontiguousarray(plans, np.float64).reshape(-1, 2)
            counts = [plans.shape[1]] * count
        else:
            counts = [len(p) for p in plans]
            packed = np.ascontiguousarray(np.concatenate([np.asarray(p, np.float64).reshape(-1, 2) for p in plans]))
        states = (RobotState * count)()
        off = 0
        for i in range(count):
            states[i].pos[:] = pos[i].tolist()
            states[i].vel[:] = vel[i].tolist()
            states[i].plan_first = off
            states[i].plan_count = counts[i]
            off += counts[i]
        check(self.L.navgpu_planner_stage(self.h, first, count, C.cast(states, C.c_void_p), _ptr(packed), len(packed)),
              "planner_stage")

    def stage_planner_raw(self, states, n, packed_plans, first=0):
        check(self.L.navgpu_planner_stage(self.h, first, n, C.cast(states, C.c_void_p), _ptr(packed_plans),
                                          len(packed_plans)), "planner_stage")

    def stage_poses(self, pos, vel, first=0):
        """A control cycle whose plan is unchanged: pose + velocity only (pos, vel: (count, 3) float32, C-contiguous)."""
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        vel = np.ascontiguousarray(vel, np.float32).reshape(-1, 3)
        check(self.L.navgpu_planner_stage_poses(self.h, first, len(pos), _ptr(pos), _ptr(vel)), "planner_stage_poses")

    def wavefront_boxes(self, first=0, count=None):
        """Cell boxes (x0, x1, y0, y1, inclusive) the last cycle's bounded wavefronts settled, [count, 4]."""
        first, count = self._range(first, count)
        out = np.zeros((count, 4), np.int32)
        check(self.L.navgpu_planner_wavefront_boxes(self.h, first, count, _ptr(out)), "wavefront_boxes")
        return out

    MAP_GRID_CRITICS = {"path": 0, "goal": 1, "goal_front": 2, "alignment": 3}
    AGGREGATIONS = {"last": 0, "sum": 1, "product": 2}

    def set_map_grid_options(self, critic, aggregation="last", yshift=0.0):
        """MapGridCostFunction's aggregationType / yshift of one of DWAPlanner's four map-grid critics."""
        check(self.L.navgpu_planner_set_map_grid_options(self.h, self.MAP_GRID_CRITICS[critic], self.AGGREGATIONS[aggregation], float(yshift)),
              "set_map_grid_options")

    def set_bounded_map_grids(self, enable):
        """navgpu_planner_set_bounded_map_grids: wavefronts stop once the robot's box is settled (default on)."""
        check(self.L.navgpu_planner_set_bounded_map_grids(self.h, 1 if enable else 0), "set_bounded_map_grids")

    def wavefront_levels(self, first=0, count=None):
        """Levels the last cycle's path / goal / goal_front wavefronts ran, [count, 3]."""
        first, count = self._range(first, count)
        out = np.zeros((count, 3), np.uint32)
        check(self.L.navgpu_planner_wavefront_levels(self.h, first, count, _ptr(out)), "wavefront_levels")
        return out

    def planner_cycle(self, first=0, count=None):
        first, count = self._range(first, count)
        check(self.L.navgpu_planner_cycle(self.h, first, count), "planner_cycle")

    def results(self, first=0, count=None):
        first, count = self._range(first, count)
        r = (PlanResult * count)()
        check(self.L.navgpu_planner_results(self.h, first, count, C.cast(r, C.c_void_p)), "planner_results")
        return list(r)

    def results_into(self, buf, first=0):
        """Same as results() but into a caller-owned (PlanResult * count) array: no Python allocation on
        the per-cycle path (a list of 256 fresh ctypes objects per call makes the interpreter's garbage
        collector the slowest part of a cycle once a large package such as torch is imported)."""
        check(self.L.navgpu_planner_results(self.h, first, len(buf), C.cast(buf, C.c_void_p)), "planner_results")
        return buf

    def set_cycles_in_flight(self, cycles):
        """navgpu_planner_set_cycles_in_flight: 2 = cycle k + 1 is handed over and queued while cycle k runs."""
        check(self.L.navgpu_planner_set_cycles_in_flight(self.h, int(cycles)), "set_cycles_in_flight")

    def results_previous_into(self, buf, first=0):
        """Results of the cycle before the latest queued one (two cycles in flight), into a (PlanResult * count) array."""
        check(self.L.navgpu_planner_results_previous(self.h, first, len(buf), C.cast(buf, C.c_void_p)), "planner_results_previous")
        return buf

    def find_best_path(self, pos, vel, plans, first=0):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        self.stage_planner(pos, vel, plans, first)
        self.planner_cycle(first, len(pos))
        return self.results(first, len(pos))

    def trajectory(self, instance):
        cap = self.desc.max_sim_steps
        out = np.zeros((cap, 3), np.float64)
        n = check(self.L.navgpu_planner_trajectory(self.h, instance, _ptr(out), cap), "planner_trajectory")
        return out[:n].copy()

    def samples(self, instance):
        cfg = self.cfg
        cap = (max(cfg.vx_samples, 2) + 1) * (max(cfg.vy_samples, 2) + 1) * (max(cfg.vth_samples, 2) + 1)
        cost = np.zeros(cap, np.float64)
        status = np.zeros(cap, np.int32)
        vel = np.zeros((cap, 3), np.float32)
        n = check(self.L.navgpu_planner_samples(self.h, instance, _ptr(cost), _ptr(status), _ptr(vel), cap),
                  "planner_samples")
        return cost[:n].copy(), status[:n].copy(), vel[:n].copy()

    def check_trajectory(self, instance, vel_samples):
        v = np.ascontiguousarray(vel_samples, np.float32)
        ok = C.c_int32()
        check(self.L.navgpu_planner_check_trajectory(self.h, instance, _ptr(v), C.byref(ok)), "check_trajectory")
        return bool(ok.value)

    def cost_cloud(self, instance):
        """MapGridVisualizer's cost cloud of one robot: (n, 7) float32 x, y, z, path, goal, occ, total."""
        cap = self.desc.size_x * self.desc.size_y
        buf = np.zeros((cap, 7), np.float32)
        n = check(self.L.navgpu_planner_cost_cloud(self.h, instance, _ptr(buf), cap), "cost_cloud")
        return buf[:n].copy()

    def oscillation(self, first=0, count=None):
        first, count = self._range(first, count)
        flags = np.zeros(count, np.uint32)
        prev = np.zeros((count, 3), np.float32)
        check(self.L.navgpu_planner_get_oscillation(self.h, first, count, _ptr(flags), _ptr(prev)), "get_oscillation")
        return flags, prev

    def set_oscillation(self, flags, prev=None, first=0):
        flags = np.ascontiguousarray(flags, np.uint32)
        p = None if prev is None else _ptr(np.ascontiguousarray(prev, np.float32))
        check(self.L.navgpu_planner_set_oscillation(self.h, first, len(flags), _ptr(flags), p), "set_oscillation")

    # ---------------------------------------------------------------- DWAPlannerROS control cycle
    def configure_local_planner(self, **limits):
        """LocalPlannerLimits + LatchedStopRotateController parameters (navgpu_local_limits)."""
        lim = N.LocalLimits(**limits)
        check(self.L.navgpu_local_planner_configure(self.h, C.byref(lim)), "local_planner_configure")

    def set_global_plan(self, instance, plan_xyyaw, plan_to_global=None):
        """DWAPlannerROS::setPlan for one robot."""
        plan = np.ascontiguousarray(plan_xyyaw, np.float64).reshape(-1, 3)
        T = None if plan_to_global is None else _ptr(np.ascontiguousarray(plan_to_global, np.float64))
        check(self.L.navgpu_local_planner_set_plan(self.h, instance, _ptr(plan), len(plan), T), "local_planner_set_plan")

    def global_plan(self, instance):
        n = check(self.L.navgpu_local_planner_get_plan(self.h, instance, None, 0), "local_planner_get_plan")
        out = np.zeros((n, 3))
        if n:
            check(self.L.navgpu_local_planner_get_plan(self.h, instance, _ptr(out), n), "local_planner_get_plan")
        return out

    def _robot_inputs(self, poses, odom_vels, have_pose):
        poses = np.asarray(poses, np.float64).reshape(-1, 3)
        vels = np.asarray(odom_vels, np.float64).reshape(-1, 3)
        arr = (N.RobotInput * len(poses))()
        for k in range(len(poses)):
            arr[k].pose[:] = [float(v) for v in poses[k]]
            arr[k].odom_vel[:] = [float(v) for v in vels[k]]
            arr[k].have_pose = 1 if have_pose is None else int(bool(have_pose[k]))
        return arr

    def compute_velocity_commands(self, poses, odom_vels, first=0, have_pose=None):
        """DWAPlannerROS::computeVelocityCommands for len(poses) robots -> list of CmdResult."""
        arr = self._robot_inputs(poses, odom_vels, have_pose)
        out = (N.CmdResult * len(arr))()
        check(self.L.navgpu_local_planner_compute_velocity_commands(self.h, first, len(arr), arr, out), "compute_velocity_commands")
        return list(out)

    def is_goal_reached(self, poses, odom_vels, first=0, have_pose=None):
        arr = self._robot_inputs(poses, odom_vels, have_pose)
        out = (C.c_int32 * len(arr))()
        check(self.L.navgpu_local_planner_is_goal_reached(self.h, first, len(arr), arr, out), "is_goal_reached")
        return [bool(v) for v in out]

    # ---------------------------------------------------------------- legacy TrajectoryPlanner
    def configure_trajectory_planner(self, cfg):
        check(self.L.navgpu_tp_configure(self.h, C.byref(cfg)), "tp_configure")

    def tp_update_plan(self, instance, plan_xy, compute_dists=False):
        plan = np.ascontiguousarray(plan_xy, np.float64).reshape(-1, 2)
        check(self.L.navgpu_tp_update_plan(self.h, instance, _ptr(plan) if len(plan) else None, len(plan), int(compute_dists)),
              "tp_update_plan")

    def tp_find_best_path(self, pos, vel, first=0):
        """TrajectoryPlanner::findBestPath for len(pos) robots -> list of TpResult."""
        pos = np.asarray(pos, np.float32).reshape(-1, 3)
        vel = np.asarray(vel, np.float32).reshape(-1, 3)
        st = (N.RobotState * len(pos))()
        for k in range(len(pos)):
            st[k].pos[:] = [float(v) for v in pos[k]]
            st[k].vel[:] = [float(v) for v in vel[k]]
        out = (N.TpResult * len(pos))()
        check(self.L.navgpu_tp_find_best_path(self.h, first, len(pos), st, out), "tp_find_best_path")
        return list(out)

    def tp_trajectory(self, instance):
        buf = np.zeros((self.desc.max_sim_steps, 3))
        n = check(self.L.navgpu_tp_trajectory(self.h, instance, _ptr(buf), len(buf)), "tp_trajectory")
        return buf[:n].copy()

    def tp_samples(self, instance):
        n = check(self.L.navgpu_tp_samples(self.h, instance, None, 0), "tp_samples")
        arr = (N.TpSample * max(n, 1))()
        check(self.L.navgpu_tp_samples(self.h, instance, arr, n), "tp_samples")
        return [(a.vx, a.vy, a.vtheta, a.cost, a.n_points) for a in arr[:n]]

    def tp_score_trajectory(self, instance, pose, vel, vel_samples):
        cost = C.c_double()
        a, b, c3 = (np.ascontiguousarray(v, np.float64) for v in (pose, vel, vel_samples))
        check(self.L.navgpu_tp_score_trajectory(self.h, instance, _ptr(a), _ptr(b), _ptr(c3), C.byref(cost)), "tp_score_trajectory")
        return cost.value

    def tp_state(self, first=0, count=None):
        first, count = self._range(first, count)
        arr = (N.TpState * count)()
        check(self.L.navgpu_tp_get_state(self.h, first, count, arr), "tp_get_state")
        return list(arr)

    def tp_set_state(self, states, first=0):
        arr = (N.TpState * len(states))(*states)
        check(self.L.navgpu_tp_set_state(self.h, first, len(states), arr), "tp_set_state")

    # ---------------------------------------------------------------- measurement
    def profile(self, enable=True):
        check(self.L.navgpu_profile_enable(self.h, int(enable)), "profile_enable")

    def profile_reset(self):
        check(self.L.navgpu_profile_reset(self.h), "profile_reset")

    def profile_select(self, names=None):
        """Bracket only the named kernels with events while profiling (None = all)."""
        mask = 0xFFFFFFFF if names is None else sum(1 << N.KERNELS.index(k) for k in names)
        check(self.L.navgpu_profile_select(self.h, mask), "profile_select")

    def profile_read(self):
        out = {}
        for k, name in enumerate(N.KERNELS):
            ms, n = C.c_double(), C.c_uint64()
            check(self.L.navgpu_profile_read(self.h, k, C.byref(ms), C.byref(n)), "profile_read")
            out[name] = (ms.value, n.value)
        return out
