"""ctypes declarations for include/navgpu.h (one-to-one)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

LAYER_STATIC, LAYER_OBSTACLE, LAYER_VOXEL, LAYER_INFLATION = 1, 2, 4, 8
GRID_MASTER, GRID_STATIC, GRID_OBSTACLE, GRID_VOXEL, GRID_PATH, GRID_GOAL, GRID_GOAL_FRONT = range(7)
OBS_MARKING, OBS_CLEARING = 1, 2
K_OBSTACLE, K_MERGE, K_INFLATE, K_BFS, K_SCORE, K_SELECT = range(6)
KERNELS = ("k_obstacle", "k_merge", "k_inflate", "k_bfs", "k_score", "k_select")


class NavgpuError(RuntimeError):
    pass


class FleetDesc(C.Structure):
    _fields_ = [("n_instances", C.c_uint32), ("size_x", C.c_uint32), ("size_y", C.c_uint32),
                ("resolution", C.c_double), ("layers", C.c_int32), ("track_unknown", C.c_int32),
                ("device", C.c_int32), ("max_points", C.c_uint32), ("max_observations", C.c_uint32),
                ("max_plan", C.c_uint32), ("max_footprint", C.c_uint32), ("max_sim_steps", C.c_uint32),
                ("keep_sample_costs", C.c_int32), ("rolling_window", C.c_int32)]


class Observation(C.Structure):
    _fields_ = [("instance", C.c_uint32), ("first_point", C.c_uint32), ("n_points", C.c_uint32),
                ("flags", C.c_uint32), ("origin_x", C.c_double), ("origin_y", C.c_double),
                ("origin_z", C.c_double), ("obstacle_range", C.c_double), ("raytrace_range", C.c_double)]


class ObstacleParams(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("footprint_clearing_enabled", C.c_int32),
                ("combination_method", C.c_int32), ("z_voxels", C.c_int32), ("max_obstacle_height", C.c_double),
                ("origin_z", C.c_double), ("z_resolution", C.c_double), ("unknown_threshold", C.c_int32),
                ("mark_threshold", C.c_int32)]


class InflationParams(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("priority_queue_order", C.c_int32), ("inflation_radius", C.c_double),
                ("cost_scaling_factor", C.c_double), ("inscribed_radius", C.c_double)]


class DwaConfig(C.Structure):
    """navgpu_dwa_config; defaults are the reference's (DWAPlanner.cfg:15-36, local_planner_limits)."""
    _fields_ = [(n, C.c_double) for n in (
        "max_trans_vel", "min_trans_vel", "max_vel_x", "min_vel_x", "max_vel_y", "min_vel_y",
        "max_rot_vel", "min_rot_vel", "acc_lim_x", "acc_lim_y", "acc_lim_theta",
        "sim_time", "sim_granularity", "angular_sim_granularity", "sim_period",
        "path_distance_bias", "goal_distance_bias", "occdist_scale",
        "forward_point_distance", "cheat_factor", "oscillation_reset_dist", "oscillation_reset_angle")] + [
        (n, C.c_int32) for n in (
            "vx_samples", "vy_samples", "vth_samples", "use_dwa", "discretize_by_time", "sum_scores",
            "allow_unknown", "rollout_trig")]

    DEFAULTS = dict(max_trans_vel=0.55, min_trans_vel=0.1, max_vel_x=0.55, min_vel_x=0.0, max_vel_y=0.1,
                    min_vel_y=-0.1, max_rot_vel=1.0, min_rot_vel=0.4, acc_lim_x=2.5, acc_lim_y=2.5,
                    acc_lim_theta=3.2, sim_time=1.7, sim_granularity=0.025, angular_sim_granularity=0.1,
                    sim_period=0.05, path_distance_bias=32.0, goal_distance_bias=24.0, occdist_scale=0.01,
                    forward_point_distance=0.325, cheat_factor=1.0, oscillation_reset_dist=0.05,
                    oscillation_reset_angle=0.2, vx_samples=3, vy_samples=10, vth_samples=20, use_dwa=1,
                    discretize_by_time=0, sum_scores=0, allow_unknown=1, rollout_trig=0)

    def __init__(self, **kw):
        super().__init__()
        d = dict(self.DEFAULTS)
        d.update(kw)
        for k, v in d.items():
            setattr(self, k, v)

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class RobotState(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("vel", C.c_float * 3), ("plan_first", C.c_uint32),
                ("plan_count", C.c_uint32)]


class PlanResult(C.Structure):
    _fields_ = [("best_index", C.c_int32), ("n_samples", C.c_int32), ("n_scored", C.c_int32),
                ("n_valid", C.c_int32), ("n_points", C.c_int32), ("oscillation_flags", C.c_uint32),
                ("xv", C.c_float), ("yv", C.c_float), ("thetav", C.c_float), ("reserved", C.c_float),
                ("cost", C.c_double), ("drive", C.c_double * 3)]


class LocalLimits(C.Structure):
    _fields_ = [("xy_goal_tolerance", C.c_double), ("yaw_goal_tolerance", C.c_double), ("rot_stopped_vel", C.c_double),
                ("trans_stopped_vel", C.c_double), ("max_rot_vel", C.c_double), ("min_rot_vel", C.c_double),
                ("acc_lim_x", C.c_double), ("acc_lim_y", C.c_double), ("acc_lim_theta", C.c_double),
                ("sim_period", C.c_double), ("prune_plan", C.c_int32), ("latch_xy_goal_tolerance", C.c_int32)]


class RobotInput(C.Structure):
    _fields_ = [("pose", C.c_double * 3), ("odom_vel", C.c_double * 3), ("have_pose", C.c_int32), ("reserved", C.c_int32)]


class CmdResult(C.Structure):
    _fields_ = [("cmd_vel", C.c_double * 3), ("ok", C.c_int32), ("branch", C.c_int32), ("local_plan_points", C.c_int32),
                ("trajectory_points", C.c_int32)]


BRANCH_NONE, BRANCH_DWA, BRANCH_STOP, BRANCH_ROTATE, BRANCH_AT_GOAL = range(5)


class TpConfig(C.Structure):
    """navgpu_tp_config; defaults are BaseLocalPlanner.cfg's (base_local_planner/cfg/BaseLocalPlanner.cfg)."""
    _fields_ = [(n, C.c_double) for n in (
        "acc_lim_x", "acc_lim_y", "acc_lim_theta", "sim_time", "sim_granularity", "angular_sim_granularity",
        "pdist_scale", "gdist_scale", "occdist_scale", "heading_lookahead", "oscillation_reset_dist",
        "escape_reset_dist", "escape_reset_theta", "max_vel_x", "min_vel_x", "max_vel_th", "min_vel_th",
        "min_in_place_vel_th", "backup_vel", "sim_period", "heading_scoring_timestep")] + [("y_vels", C.c_double * 8)] + [
        (n, C.c_int32) for n in ("n_y_vels", "vx_samples", "vtheta_samples", "holonomic_robot", "dwa", "allow_unknown",
                                 "heading_scoring", "simple_attractor")]

    DEFAULTS = dict(acc_lim_x=2.5, acc_lim_y=2.5, acc_lim_theta=3.2, sim_time=1.7, sim_granularity=0.025,
                    angular_sim_granularity=0.025, pdist_scale=0.6, gdist_scale=0.8, occdist_scale=0.01,
                    heading_lookahead=0.325, oscillation_reset_dist=0.05, escape_reset_dist=0.10,
                    escape_reset_theta=1.5707963267948966, max_vel_x=0.55, min_vel_x=0.0, max_vel_th=1.0, min_vel_th=-1.0,
                    min_in_place_vel_th=0.4, backup_vel=-0.1, sim_period=0.05, heading_scoring_timestep=0.1,
                    y_vels=(-0.3, -0.1, 0.1, 0.3),
                    vx_samples=20, vtheta_samples=20, holonomic_robot=1, dwa=0, allow_unknown=1, heading_scoring=0,
                    simple_attractor=0)

    def __init__(self, **kw):
        super().__init__()
        d = dict(self.DEFAULTS)
        d.update(kw)
        yv = list(d.pop("y_vels"))
        d.pop("n_y_vels", None)
        for k, v in d.items():
            setattr(self, k, v)
        self.n_y_vels = len(yv)
        for i, v in enumerate(yv):
            self.y_vels[i] = float(v)

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n not in ("y_vels", "n_y_vels")}
        d["y_vels"] = tuple(self.y_vels[i] for i in range(self.n_y_vels))
        return d


class TpState(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("reserved", C.c_uint32), ("prev_x", C.c_double), ("prev_y", C.c_double),
                ("escape_x", C.c_double), ("escape_y", C.c_double), ("escape_theta", C.c_double)]


class TpResult(C.Structure):
    _fields_ = [("xv", C.c_double), ("yv", C.c_double), ("thetav", C.c_double), ("cost", C.c_double), ("drive", C.c_double * 3),
                ("n_points", C.c_int32), ("n_samples", C.c_int32), ("best_sample", C.c_int32), ("reserved", C.c_int32)]


class TpSample(C.Structure):
    _fields_ = [("vx", C.c_double), ("vy", C.c_double), ("vtheta", C.c_double), ("cost", C.c_double), ("n_points", C.c_int32),
                ("reserved", C.c_int32)]


class NavfnResult(C.Structure):
    """Mirror of navgpu_navfn_result (include/navgpu.h)."""
    _fields_ = [("found", C.c_int32), ("path_length", C.c_int32), ("cycles", C.c_int32), ("start_potential", C.c_float)]


class GlobalPlannerParams(C.Structure):
    """Mirror of navgpu_global_planner_params (include/navgpu.h); defaults = GlobalPlanner.cfg / planner_core.cpp:105-152."""
    _fields_ = [("use_dijkstra", C.c_int32), ("use_quadratic", C.c_int32), ("use_grid_path", C.c_int32), ("old_navfn_behavior", C.c_int32),
                ("allow_unknown", C.c_int32), ("lethal_cost", C.c_int32), ("neutral_cost", C.c_int32), ("cost_factor", C.c_float),
                ("outline_map", C.c_int32), ("reserved", C.c_int32)]

    def __init__(self, **kw):
        super().__init__()
        d = dict(use_dijkstra=1, use_quadratic=1, use_grid_path=0, old_navfn_behavior=0, allow_unknown=1, lethal_cost=253, neutral_cost=50,
                 cost_factor=3.0, outline_map=1, reserved=0)
        d.update(kw)
        for k, v in d.items():
            setattr(self, k, v)


def lib_path():
    return os.path.join(_HERE, "libnavgpu.so")


def build():
    """hipcc --offload-arch=gfx950 build of libnavgpu.so (cross-compiles without a GPU)."""
    subprocess.run(["make", "-s", "-C", os.path.join(_HERE, "csrc")], check=True)


# every symbol include/navgpu.h declares: (name, restype, argtypes)
vp, u32, i32, dbl = C.c_void_p, C.c_uint32, C.c_int32, C.c_double
SYMBOLS = [
    ("navgpu_version", C.c_char_p, []),
    ("navgpu_strerror", C.c_char_p, [C.c_int]),
    ("navgpu_last_error", C.c_char_p, []),
    ("navgpu_device_count", C.c_int, []),
    ("navgpu_fleet_create", C.c_int, [C.POINTER(FleetDesc), C.POINTER(vp)]),
    ("navgpu_fleet_destroy", C.c_int, [vp]),
    ("navgpu_sync", C.c_int, [vp]),
    ("navgpu_fleet_set_alloc_limit", C.c_int, [vp, C.c_uint64]),
    ("navgpu_stream", vp, [vp]),
    ("navgpu_fleet_set_origin", C.c_int, [vp, u32, u32, vp]),
    ("navgpu_fleet_get_origin", C.c_int, [vp, u32, u32, vp]),
    ("navgpu_grid_upload", C.c_int, [vp, C.c_int, u32, u32, vp]),
    ("navgpu_grid_download", C.c_int, [vp, C.c_int, u32, u32, vp]),
    ("navgpu_grid_device", C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]),
    ("navgpu_grid_reset", C.c_int, [vp, C.c_int, u32, u32]),
    ("navgpu_grid_reset_window", C.c_int, [vp, C.c_int, u32, u32, u32, u32, u32, u32]),
    ("navgpu_layer_reset_bounding_box", C.c_int, [vp, u32, u32, vp]),
    ("navgpu_static_set_map", C.c_int, [vp, u32, u32, vp, i32, i32, i32, i32, i32]),
    ("navgpu_static_set_rolling_map", C.c_int, [vp, vp, u32, u32, C.c_double, C.c_double, C.c_double, i32, i32, i32, i32, i32]),
    ("navgpu_static_set_transform", C.c_int, [vp, u32, u32, vp]),
    ("navgpu_obstacle_configure", C.c_int, [vp, C.POINTER(ObstacleParams)]),
    ("navgpu_inflation_configure", C.c_int, [vp, C.POINTER(InflationParams)]),
    ("navgpu_set_footprint", C.c_int, [vp, u32, u32, vp, u32]),
    ("navgpu_costmap_stage", C.c_int, [vp, u32, u32, vp, vp, u32, vp, u32]),
    ("navgpu_costmap_update", C.c_int, [vp, u32, u32]),
    ("navgpu_costmap_bounds", C.c_int, [vp, u32, u32, vp]),
    ("navgpu_inflate", C.c_int, [vp, u32, u32, vp]),
    ("navgpu_obstacle_update_bounds", C.c_int, [vp, u32, u32, vp]),
    ("navgpu_obstacle_update_costs", C.c_int, [vp, u32, u32, vp]),
    ("navgpu_planner_configure", C.c_int, [vp, C.POINTER(DwaConfig)]),
    ("navgpu_planner_set_plan", C.c_int, [vp, u32, u32]),
    ("navgpu_planner_stage", C.c_int, [vp, u32, u32, vp, vp, u32]),
    ("navgpu_planner_stage_poses", C.c_int, [vp, u32, u32, vp, vp]),
    ("navgpu_planner_cycle", C.c_int, [vp, u32, u32]),
    ("navgpu_planner_set_bounded_map_grids", C.c_int, [vp, C.c_int32]),
    ("navgpu_planner_set_map_grid_options", C.c_int, [vp, C.c_int32, C.c_int32, C.c_double]),
    ("navgpu_planner_wavefront_levels", C.c_int, [vp, u32, u32, vp]),
    ("navgpu_planner_wavefront_boxes", C.c_int, [vp, u32, u32, vp]),
    ("navgpu_planner_results", C.c_int, [vp, u32, u32, vp]),
    ("navgpu_planner_set_cycles_in_flight", C.c_int, [vp, C.c_int32]),
    ("navgpu_planner_results_previous", C.c_int, [vp, u32, u32, vp]),
    ("navgpu_planner_trajectory", C.c_int, [vp, u32, vp, u32]),
    ("navgpu_planner_samples", C.c_int, [vp, u32, vp, vp, vp, u32]),
    ("navgpu_planner_check_trajectory", C.c_int, [vp, u32, vp, C.POINTER(i32)]),
    ("navgpu_planner_cost_cloud", C.c_int, [vp, u32, vp, u32]),
    ("navgpu_planner_get_oscillation", C.c_int, [vp, u32, u32, vp, vp]),
    ("navgpu_planner_set_oscillation", C.c_int, [vp, u32, u32, vp, vp]),
    ("navgpu_local_plan_window", C.c_int, [vp, u32, vp, vp, dbl, i32, vp, u32, C.POINTER(u32), C.POINTER(u32)]),
    ("navgpu_shortest_angular_distance", dbl, [dbl, dbl]),
    ("navgpu_local_planner_configure", C.c_int, [vp, C.POINTER(LocalLimits)]),
    ("navgpu_local_planner_set_plan", C.c_int, [vp, u32, vp, u32, vp]),
    ("navgpu_local_planner_compute_velocity_commands", C.c_int, [vp, u32, u32, vp, vp]),
    ("navgpu_local_planner_is_goal_reached", C.c_int, [vp, u32, u32, vp, vp]),
    ("navgpu_local_planner_get_plan", C.c_int, [vp, u32, vp, u32]),
    ("navgpu_tp_configure", C.c_int, [vp, C.POINTER(TpConfig)]),
    ("navgpu_tp_update_plan", C.c_int, [vp, u32, vp, u32, i32]),
    ("navgpu_tp_find_best_path", C.c_int, [vp, u32, u32, vp, vp]),
    ("navgpu_tp_trajectory", C.c_int, [vp, u32, vp, u32]),
    ("navgpu_tp_samples", C.c_int, [vp, u32, vp, u32]),
    ("navgpu_tp_score_trajectory", C.c_int, [vp, u32, vp, vp, vp, C.POINTER(dbl)]),
    ("navgpu_tp_get_state", C.c_int, [vp, u32, u32, vp]),
    ("navgpu_tp_set_state", C.c_int, [vp, u32, u32, vp]),
    ("navgpu_navfn_create", C.c_int, [u32, u32, u32, i32, C.POINTER(vp)]),
    ("navgpu_navfn_destroy", C.c_int, [vp]),
    ("navgpu_navfn_set_costmap", C.c_int, [vp, u32, u32, vp, i32, i32, i32]),
    ("navgpu_navfn_set_costmap_from_fleet", C.c_int, [vp, u32, u32, vp, u32, i32]),
    ("navgpu_navfn_plan", C.c_int, [vp, u32, u32, vp, vp, i32, i32, vp]),
    ("navgpu_navfn_plan_wavefront", C.c_int, [vp, u32, u32, vp, vp, i32, vp]),
    ("navgpu_global_planner_plan", C.c_int, [vp, u32, u32, C.POINTER(GlobalPlannerParams), vp, vp, vp, vp]),
    ("navgpu_global_planner_plan_wavefront", C.c_int, [vp, u32, u32, C.POINTER(GlobalPlannerParams), vp, vp, vp, vp]),
    ("navgpu_navfn_path", C.c_int, [vp, u32, vp, u32]),
    ("navgpu_navfn_potential", C.c_int, [vp, u32, vp]),
    ("navgpu_footprint_radii", C.c_int, [vp, u32, C.POINTER(dbl), C.POINTER(dbl)]),
    ("navgpu_footprint_pad", C.c_int, [vp, u32, dbl]),
    ("navgpu_footprint_from_radius", C.c_int, [dbl, vp]),
    ("navgpu_costmap_export", C.c_int, [vp, u32, u32, u32, u32, u32, vp]),
    ("navgpu_profile_enable", C.c_int, [vp, i32]),
    ("navgpu_profile_select", C.c_int, [vp, u32]),
    ("navgpu_profile_reset", C.c_int, [vp]),
    ("navgpu_profile_read", C.c_int, [vp, i32, C.POINTER(dbl), C.POINTER(C.c_uint64)]),
    ("navgpu_kernel_name", C.c_char_p, [i32]),
    ("navgpu_device_sincos", C.c_int, [i32, vp, u32, vp, vp]),
]


def lib():
    """Load libnavgpu.so.  Raises (never falls back) when the HIP extension is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise NavgpuError(f"{path} is missing: build it with navigation_amd.build() / __graft_entry__.build(); "
                          "there is no CPU fallback")
    # (see bench.py / INTEGRATION.md 4: small host-to-device copies through the runtime's copy kernels - its SDMA path stalls ~8 ms once
    # in ~2 500 async copies; a no-op when the HIP runtime of this process has already started or the caller has set the variable)
    os.environ.setdefault("GPU_FORCE_BLIT_COPY_SIZE", "2048")
    L = C.CDLL(path)
    for name, res, args in SYMBOLS:
        fn = getattr(L, name)  # AttributeError if the header and the library ever diverge
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


def check(rc, what=""):
    if rc < 0:
        L = lib()
        raise NavgpuError(f"{what}: {L.navgpu_strerror(rc).decode()} ({rc}) {L.navgpu_last_error().decode()}")
    return rc
