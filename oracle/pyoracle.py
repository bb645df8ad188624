"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/liboracle.so (the CPU restatement of the reference hot path) and, when
present, oracle/_ref/libref_iters.so (the reference's own header-only iterators).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package navigation_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


class DwaConfig(C.Structure):
    """Mirror of navgpu_dwa_config (include/navgpu.h); defaults = the reference's defaults
    (dwa_local_planner/cfg/DWAPlanner.cfg:15-36, local_planner_limits/__init__.py:17-45)."""

    _fields_ = [(n, C.c_double) for n in (
        "max_trans_vel", "min_trans_vel", "max_vel_x", "min_vel_x", "max_vel_y", "min_vel_y",
        "max_rot_vel", "min_rot_vel", "acc_lim_x", "acc_lim_y", "acc_lim_theta",
        "sim_time", "sim_granularity", "angular_sim_granularity", "sim_period",
        "path_distance_bias", "goal_distance_bias", "occdist_scale",
        "forward_point_distance", "cheat_factor", "oscillation_reset_dist", "oscillation_reset_angle")] + [
        (n, C.c_int32) for n in (
            "vx_samples", "vy_samples", "vth_samples", "use_dwa", "discretize_by_time", "sum_scores",
            "allow_unknown", "rollout_trig")]

    def __init__(self, **kw):
        super().__init__()
        d = dict(max_trans_vel=0.55, min_trans_vel=0.1, max_vel_x=0.55, min_vel_x=0.0, max_vel_y=0.1,
                 min_vel_y=-0.1, max_rot_vel=1.0, min_rot_vel=0.4, acc_lim_x=2.5, acc_lim_y=2.5,
                 acc_lim_theta=3.2, sim_time=1.7, sim_granularity=0.025, angular_sim_granularity=0.1,
                 sim_period=0.05, path_distance_bias=32.0, goal_distance_bias=24.0, occdist_scale=0.01,
                 forward_point_distance=0.325, cheat_factor=1.0, oscillation_reset_dist=0.05,
                 oscillation_reset_angle=0.2, vx_samples=3, vy_samples=10, vth_samples=20, use_dwa=1,
                 discretize_by_time=0, sum_scores=0, allow_unknown=1, rollout_trig=0)
        d.update(kw)
        for k, v in d.items():
            setattr(self, k, v)


class PlanResult(C.Structure):
    """Mirror of navgpu_plan_result (include/navgpu.h)."""

    _fields_ = [("best_index", C.c_int32), ("n_samples", C.c_int32), ("n_scored", C.c_int32),
                ("n_valid", C.c_int32), ("n_points", C.c_int32), ("oscillation_flags", C.c_uint32),
                ("xv", C.c_float), ("yv", C.c_float), ("thetav", C.c_float), ("reserved", C.c_float),
                ("cost", C.c_double), ("drive", C.c_double * 3)]


def build():
    """Compile liboracle.so (and oracle/_ref when /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    vp = C.c_void_p
    d = C.c_double
    i = C.c_int
    u = C.c_uint32

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("orc_lc_create", vp, i)
    sig("orc_lc_destroy", None, vp)
    sig("orc_lc_set_rolling", None, vp, i)
    sig("orc_lc_get_origin", None, vp, f64p)
    sig("orc_lc_resize", None, vp, u, u, d, d, d)
    sig("orc_lc_add_static", None, vp, i8p, u, u, d, d, d, i, i)
    sig("orc_lc_add_static_rolling", None, vp, i8p, u, u, d, d, d, i, i, i, i, i)
    sig("orc_lc_set_static_transform", None, vp, f64p)
    sig("orc_lc_add_obstacle", None, vp, i, i, d)
    sig("orc_lc_add_voxel", None, vp, i, i, d, u, d, d, u, u)
    sig("orc_lc_add_inflation", None, vp, d, d, i)
    sig("orc_lc_set_inflation_exact", None, vp, i)
    sig("orc_lc_set_footprint", None, vp, f64p, u)
    sig("orc_lc_inscribed_radius", d, vp)
    sig("orc_lc_circumscribed_radius", d, vp)
    sig("orc_lc_add_observation", None, vp, d, d, d, f32p, u, d, d, i, i)
    sig("orc_lc_clear_observations", None, vp)
    sig("orc_lc_update_map", None, vp, d, d, d)
    sig("orc_lc_reset_bounding_box", None, vp, d, d, d, d)
    sig("orc_lc_get_master", None, vp, u8p)
    sig("orc_lc_set_master", None, vp, u8p)
    sig("orc_lc_get_layer", None, vp, i, u8p)
    sig("orc_lc_set_layer", None, vp, i, u8p)
    sig("orc_lc_get_voxels", None, vp, u32p)
    sig("orc_lc_set_voxels", None, vp, u32p)
    sig("orc_lc_get_bounds", None, vp, i32p)
    sig("orc_lc_size", None, vp, C.POINTER(u), C.POINTER(u))
    sig("orc_inflate", None, u8p, u, u, d, d, d, d, i, i, i, i, i)
    sig("orc_cost_lut", u, d, d, d, d, C.c_void_p, C.c_void_p, u)
    sig("orc_compute_cost", C.c_uint8, d, d, d, d)
    sig("orc_raytrace_cells", i, u, u, u, u, u, u, u, u32p, i)
    sig("orc_polygon_fill", i, u8p, u, u, d, d, d, f64p, u, C.c_uint8)
    sig("orc_merge", None, u8p, u8p, u, u, i, i, i, i, i)
    sig("orc_min_max_distances", None, f64p, u, C.POINTER(d), C.POINTER(d))
    sig("orc_vg_create", vp, u, u, u)
    sig("orc_vg_destroy", None, vp)
    sig("orc_vg_mark_line", None, vp, d, d, d, d, d, d)
    sig("orc_vg_clear_line", None, vp, d, d, d, d, d, d)
    sig("orc_vg_mark_voxel", None, vp, u, u, u)
    sig("orc_vg_get_voxel", i, vp, u, u, u)
    sig("orc_vg_data", None, vp, u32p)
    sig("orc_velocity_samples", i, d, d, i, f64p, i)
    sig("orc_line_cells", i, i, i, i, i, i32p, i)
    sig("orc_footprint_cost", d, u8p, u, u, d, d, d, d, d, d, f64p, u, i)
    sig("orc_obstacle_step_cost", d, u8p, u, u, d, d, d, d, d, d, f64p, u, i)
    sig("orc_adjust_plan", i, f64p, u, d, f64p, i)
    sig("orc_map_grid", None, u8p, u, u, d, d, d, f64p, u, i, i, f64p)
    sig("orc_map_grid_seeded", None, u8p, u, u, u32p, u, i, f64p)
    sig("orc_samples", i, C.POINTER(DwaConfig), f32p, f32p, f32p, f32p, i)
    sig("orc_generate_trajectory", i, C.POINTER(DwaConfig), f32p, f32p, f32p, f64p, i, f64p)
    sig("orc_dwa_create", vp, u, u, d, d, d)
    sig("orc_dwa_destroy", None, vp)
    sig("orc_dwa_set_costmap", None, vp, u8p)
    sig("orc_dwa_configure", None, vp, C.POINTER(DwaConfig))
    sig("orc_dwa_set_plan", None, vp)
    sig("orc_dwa_set_map_grid_options", None, vp, i, i, d)
    sig("orc_dwa_cycle", i, vp, f32p, f32p, f64p, u, f64p, u, C.POINTER(PlanResult), C.c_void_p, i,
        C.c_void_p, C.c_void_p, C.c_void_p, i)
    sig("orc_dwa_check_trajectory", i, vp, f32p, f32p, f32p)
    sig("orc_dwa_update_plan", None, vp, f32p, f64p, u)
    sig("orc_dwa_get_grid", None, vp, i, f64p)
    sig("orc_dwa_get_samples", i, vp, f32p, i)
    sig("orc_dwa_alignment_scale", d, vp)
    sig("orc_dwa_get_oscillation", None, vp, C.POINTER(u), f32p)
    sig("orc_dwa_set_oscillation", None, vp, u, f32p)
    sig("orc_tp_create", vp, u, u, d, d, d, u8p, C.c_void_p, f64p, u)
    sig("orc_tp_destroy", None, vp)
    sig("orc_tp_set_costmap", None, vp, u8p)
    sig("orc_tp_update_plan", None, vp, C.c_void_p, u, i)
    sig("orc_tp_find_best_path", i, vp, f32p, f32p, C.c_void_p, C.c_void_p, i, C.c_void_p, i)
    sig("orc_tp_score_trajectory", d, vp, f64p, f64p, f64p)
    sig("orc_tp_generate", d, vp, f64p, f64p, f64p, f64p, d)
    sig("orc_tp_get_grid", None, vp, i, f64p)
    sig("orc_tp_get_state", None, vp, C.c_void_p)
    sig("orc_tp_set_state", None, vp, C.c_void_p)
    sig("orc_tp_footprint_cells", i, u, u, d, d, d, f32p, f64p, u, i, C.c_void_p, i)
    sig("orc_gp_calculate_potential", None, i, f32p, i, i, u8p, i32p, f32p, i, f32p)
    sig("orc_gp_grid_path", i, f32p, i, i, C.c_double, C.c_double, C.c_double, C.c_double, f32p, i)
    sig("orc_global_planner_plan", i, u8p, i, i, i32p, C.c_float, f64p, f64p, i32p, C.c_void_p, C.c_void_p, i, C.POINTER(i), C.POINTER(i))
    sig("orc_navfn_plan", i, u8p, i, i, i, i, i32p, i32p, i, i, C.c_void_p, C.c_void_p, i, C.POINTER(i))
    sig("orc_navfn_fixed_point", i, u8p, i, i, i, i, i32p, i32p, C.c_void_p, C.c_void_p, i)
    sig("orc_bench_dwa", d, u, u, d, u8p, u, C.POINTER(DwaConfig), f32p, f32p, f64p, u, f64p, f64p, u, u, u,
        C.POINTER(C.c_uint64))
    sig("orc_bench_inflate", d, u8p, u, u, u, d, d, d, d, u, u)
    _LIB = L
    return L


def ref():
    """The reference's own header-only iterators (oracle/_ref), or None when not built."""
    global _REF
    if _REF is not None:
        return _REF
    path = os.path.join(_HERE, "_ref", "libref_iters.so")
    if not os.path.exists(path):
        return None
    R = C.CDLL(path)
    R.ref_line_cells.restype = C.c_int
    R.ref_line_cells.argtypes = [C.c_int] * 4 + [i32p, C.c_int]
    R.ref_velocity_samples.restype = C.c_int
    R.ref_velocity_samples.argtypes = [C.c_double, C.c_double, C.c_int, f64p, C.c_int]
    R.ref_cost_values.restype = None
    R.ref_cost_values.argtypes = [u8p]
    _REF = R
    return R


_REF_GP = None


def ref_gp():
    """The reference's own QuadraticCalculator / PotentialCalculator / GridPath (oracle/_ref/libref_gp.so), or None when not built."""
    global _REF_GP
    if _REF_GP is not None:
        return _REF_GP
    path = os.path.join(_HERE, "_ref", "libref_gp.so")
    if not os.path.exists(path):
        return None
    R = C.CDLL(path)
    R.ref_gp_calculate_potential.restype = None
    R.ref_gp_calculate_potential.argtypes = [C.c_int, f32p, C.c_int, C.c_int, u8p, i32p, f32p, C.c_int, f32p]
    R.ref_gp_grid_path.restype = C.c_int
    R.ref_gp_grid_path.argtypes = [f32p, C.c_int, C.c_int] + [C.c_double] * 4 + [f32p, C.c_int]
    _REF_GP = R
    return R


# ----------------------------------------------------------------------------- convenience wrappers
def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class LayeredCostmap:
    """LayeredCostmap + layers with the reference tests' injection hooks (testing_helper.h)."""

    def __init__(self, track_unknown=False):
        self.L = lib()
        self.h = self.L.orc_lc_create(int(track_unknown))

    def __del__(self):
        try:
            self.L.orc_lc_destroy(self.h)
        except Exception:
            pass

    def resize(self, sx, sy, res=1.0, ox=0.0, oy=0.0):
        self.L.orc_lc_resize(self.h, sx, sy, res, ox, oy)

    def set_rolling(self, rolling=True):
        self.L.orc_lc_set_rolling(self.h, int(rolling))

    def origin(self):
        o = np.zeros(2, np.float64)
        self.L.orc_lc_get_origin(self.h, o)
        return o

    def add_static(self, occ, res=1.0, ox=0.0, oy=0.0, track_unknown_space=True, use_maximum=False):
        occ = np.ascontiguousarray(occ, dtype=np.int8)
        sy, sx = occ.shape
        self.L.orc_lc_add_static(self.h, occ, sx, sy, res, ox, oy, int(track_unknown_space), int(use_maximum))

    def add_static_rolling(self, occ, res, ox, oy, track_unknown_space=True, use_maximum=False, trinary=True, lethal_threshold=100,
                           unknown_cost_value=-1):
        """StaticLayer of a rolling-window costmap: the static map keeps its own geometry (static_layer.cpp:187-193)."""
        occ = np.ascontiguousarray(occ, dtype=np.int8)
        sy, sx = occ.shape
        self.L.orc_lc_add_static_rolling(self.h, occ, sx, sy, res, ox, oy, int(track_unknown_space), int(use_maximum), int(trinary),
                                         int(lethal_threshold), int(unknown_cost_value) & 0xFF)

    def set_static_transform(self, basis, origin):
        """map_frame <- global_frame as tf::Transform: 3x3 basis (row-major) and origin."""
        m = np.concatenate([np.asarray(basis, np.float64).reshape(9), np.asarray(origin, np.float64).reshape(3)])
        self.L.orc_lc_set_static_transform(self.h, np.ascontiguousarray(m))

    def add_obstacle(self, combination_method=1, footprint_clearing=True, max_obstacle_height=2.0):
        self.L.orc_lc_add_obstacle(self.h, combination_method, int(footprint_clearing), max_obstacle_height)

    def add_voxel(self, combination_method=1, footprint_clearing=True, max_obstacle_height=2.0, z_voxels=10,
                  origin_z=0.0, z_resolution=0.2, unknown_threshold=15, mark_threshold=0):
        self.L.orc_lc_add_voxel(self.h, combination_method, int(footprint_clearing), max_obstacle_height, z_voxels,
                                origin_z, z_resolution, unknown_threshold, mark_threshold)

    def add_inflation(self, radius, scaling, exact=False):
        self.L.orc_lc_add_inflation(self.h, radius, scaling, int(exact))

    def set_footprint(self, xy):
        xy = _f64(xy).reshape(-1, 2)
        self.L.orc_lc_set_footprint(self.h, xy, len(xy))

    @property
    def inscribed_radius(self):
        return self.L.orc_lc_inscribed_radius(self.h)

    def add_observation(self, points, origin=(0.0, 0.0, 1.0), obstacle_range=100.0, raytrace_range=100.0,
                        marking=True, clearing=True):
        pts = _f32(points).reshape(-1, 3)
        self.L.orc_lc_add_observation(self.h, origin[0], origin[1], origin[2], pts, len(pts), obstacle_range,
                                      raytrace_range, int(marking), int(clearing))

    def clear_observations(self):
        self.L.orc_lc_clear_observations(self.h)

    def update_map(self, rx=0.0, ry=0.0, ryaw=0.0):
        self.L.orc_lc_update_map(self.h, rx, ry, ryaw)

    def reset_bounding_box(self, min_x, min_y, max_x, max_y):
        """CostmapLayer::resetBoundingBox (costmap_layer.cpp:30-43) on the obstacle / voxel layer."""
        self.L.orc_lc_reset_bounding_box(self.h, min_x, min_y, max_x, max_y)

    def size(self):
        sx, sy = C.c_uint32(), C.c_uint32()
        self.L.orc_lc_size(self.h, C.byref(sx), C.byref(sy))
        return sx.value, sy.value

    def master(self):
        sx, sy = self.size()
        out = np.empty((sy, sx), np.uint8)
        self.L.orc_lc_get_master(self.h, out)
        return out

    def layer(self, which=2):
        sx, sy = self.size()
        out = np.empty((sy, sx), np.uint8)
        self.L.orc_lc_get_layer(self.h, which, out)
        return out

    def set_layer(self, cells, which=2):
        self.L.orc_lc_set_layer(self.h, which, np.ascontiguousarray(cells, np.uint8))

    def set_master(self, cells):
        self.L.orc_lc_set_master(self.h, np.ascontiguousarray(cells, np.uint8))

    def voxels(self):
        sx, sy = self.size()
        out = np.empty((sy, sx), np.uint32)
        self.L.orc_lc_get_voxels(self.h, out)
        return out

    def bounds(self):
        b = np.zeros(4, np.int32)
        self.L.orc_lc_get_bounds(self.h, b)
        return b


def inflate(grid, res, radius, scaling, inscribed, box=None, exact=False):
    g = np.ascontiguousarray(grid, np.uint8).copy()
    sy, sx = g.shape
    if box is None:
        box = (0, 0, sx, sy)
    lib().orc_inflate(g, sx, sy, res, radius, scaling, inscribed, box[0], box[1], box[2], box[3], int(exact))
    return g


def cost_lut(res, radius, scaling, inscribed):
    R = lib().orc_cost_lut(res, radius, scaling, inscribed, None, None, 0)
    n = R + 2
    costs = np.zeros((n, n), np.uint8)
    dists = np.zeros((n, n), np.float64)
    lib().orc_cost_lut(res, radius, scaling, inscribed, costs.ctypes.data, dists.ctypes.data, n * n)
    return R, costs, dists


def velocity_samples(mn, mx, n):
    out = np.zeros(max(4, n + 4), np.float64)
    k = lib().orc_velocity_samples(mn, mx, n, out, len(out))
    return out[:k].copy()


def line_cells(x0, y0, x1, y1):
    cap = max(abs(x1 - x0), abs(y1 - y0)) + 2
    out = np.zeros((cap, 2), np.int32)
    k = lib().orc_line_cells(x0, y0, x1, y1, out, cap)
    return out[:k].copy()


def raytrace_cells(sx, sy, x0, y0, x1, y1, max_length=0xFFFFFFFF):
    cap = max(abs(x1 - x0), abs(y1 - y0)) + 2
    out = np.zeros(cap, np.uint32)
    k = lib().orc_raytrace_cells(sx, sy, x0, y0, x1, y1, max_length, out, cap)
    return out[:k].copy()


def footprint_cost(grid, res, ox, oy, x, y, th, fp, allow_unknown=True):
    g = np.ascontiguousarray(grid, np.uint8)
    fp = _f64(fp).reshape(-1, 2)
    return lib().orc_footprint_cost(g, g.shape[1], g.shape[0], res, ox, oy, x, y, th, fp, len(fp), int(allow_unknown))


def obstacle_step_cost(grid, res, ox, oy, x, y, th, fp, allow_unknown=True):
    g = np.ascontiguousarray(grid, np.uint8)
    fp = _f64(fp).reshape(-1, 2)
    return lib().orc_obstacle_step_cost(g, g.shape[1], g.shape[0], res, ox, oy, x, y, th, fp, len(fp),
                                        int(allow_unknown))


def adjust_plan(plan, res):
    p = _f64(plan).reshape(-1, 2)
    k = lib().orc_adjust_plan(p, len(p), res, np.zeros((1, 2)), 0)
    out = np.zeros((max(k, 1), 2), np.float64)
    lib().orc_adjust_plan(p, len(p), res, out, k)
    return out[:k]


def map_grid(grid, res, ox, oy, plan, mode, allow_unknown=True):
    g = np.ascontiguousarray(grid, np.uint8)
    p = _f64(plan).reshape(-1, 2)
    out = np.zeros(g.shape, np.float64)
    lib().orc_map_grid(g, g.shape[1], g.shape[0], res, ox, oy, p, len(p), mode, int(allow_unknown), out)
    return out


def map_grid_seeded(grid, seeds, allow_unknown=True):
    g = np.ascontiguousarray(grid, np.uint8)
    s = np.ascontiguousarray(seeds, np.uint32)
    out = np.zeros(g.shape, np.float64)
    lib().orc_map_grid_seeded(g, g.shape[1], g.shape[0], s, len(s), int(allow_unknown), out)
    return out


def samples(cfg, pos, vel, goal=(0, 0, 0)):
    cap = (cfg.vx_samples + 2) * (cfg.vy_samples + 2) * (cfg.vth_samples + 2)
    out = np.zeros((cap, 3), np.float32)
    k = lib().orc_samples(C.byref(cfg), _f32(pos), _f32(vel), _f32(goal), out, cap)
    return out[:k].copy()


def generate_trajectory(cfg, pos, vel, sample, cap=4096):
    pts = np.zeros((cap, 3), np.float64)
    meta = np.zeros(4, np.float64)
    k = lib().orc_generate_trajectory(C.byref(cfg), _f32(pos), _f32(vel), _f32(sample), pts, cap, meta)
    return k, pts[:max(k, 0)].copy(), meta


class DwaPlanner:
    def __init__(self, grid, res, ox=0.0, oy=0.0, cfg=None):
        self.L = lib()
        g = np.ascontiguousarray(grid, np.uint8)
        self.shape = g.shape
        self.h = self.L.orc_dwa_create(g.shape[1], g.shape[0], res, ox, oy)
        self.L.orc_dwa_set_costmap(self.h, g)
        self.cfg = cfg or DwaConfig()
        self.L.orc_dwa_configure(self.h, C.byref(self.cfg))

    def __del__(self):
        try:
            self.L.orc_dwa_destroy(self.h)
        except Exception:
            pass

    def set_costmap(self, grid):
        self.L.orc_dwa_set_costmap(self.h, np.ascontiguousarray(grid, np.uint8))

    def configure(self, cfg):
        self.cfg = cfg
        self.L.orc_dwa_configure(self.h, C.byref(cfg))

    def set_plan(self):
        self.L.orc_dwa_set_plan(self.h)

    def set_map_grid_options(self, critic, aggregation="last", yshift=0.0):
        """MapGridCostFunction's aggregationType / yshift (map_grid_cost_function.cpp:42-53) of path | goal | goal_front | alignment."""
        self.L.orc_dwa_set_map_grid_options(self.h, {"path": 0, "goal": 1, "goal_front": 2, "alignment": 3}[critic],
                                            {"last": 0, "sum": 1, "product": 2}[aggregation], float(yshift))

    def cycle(self, pos, vel, plan, footprint, want_samples=True, traj_cap=4096):
        plan = _f64(plan).reshape(-1, 2)
        fp = _f64(footprint).reshape(-1, 2)
        res = PlanResult()
        cap = (self.cfg.vx_samples + 2) * (self.cfg.vy_samples + 2) * (self.cfg.vth_samples + 2)
        traj = np.zeros((traj_cap, 3), np.float64)
        cref = np.zeros(cap, np.float64)
        cfull = np.zeros(cap, np.float64)
        status = np.zeros(cap, np.int32)
        n = self.L.orc_dwa_cycle(self.h, _f32(pos), _f32(vel), plan, len(plan), fp, len(fp), C.byref(res),
                                 traj.ctypes.data, traj_cap,
                                 cref.ctypes.data if want_samples else None,
                                 cfull.ctypes.data if want_samples else None,
                                 status.ctypes.data if want_samples else None, cap if want_samples else 0)
        return res, traj[:max(res.n_points, 0)].copy(), cref[:n].copy(), cfull[:n].copy(), status[:n].copy()

    def update_plan(self, pos, plan):
        plan = _f64(plan).reshape(-1, 2)
        self.L.orc_dwa_update_plan(self.h, _f32(pos), plan, len(plan))

    def check_trajectory(self, pos, vel, sample):
        return bool(self.L.orc_dwa_check_trajectory(self.h, _f32(pos), _f32(vel), _f32(sample)))

    def grid(self, which):
        out = np.zeros(self.shape, np.float64)
        self.L.orc_dwa_get_grid(self.h, which, out)
        return out

    def samples(self):
        """Velocity samples of the last cycle, (n, 3) float32 in slot order."""
        cap = (self.cfg.vx_samples + 2) * (self.cfg.vy_samples + 2) * (self.cfg.vth_samples + 2)
        out = np.zeros((cap, 3), np.float32)
        n = self.L.orc_dwa_get_samples(self.h, out, cap)
        return out[:n].copy()

    def oscillation(self):
        f = C.c_uint32()
        prev = np.zeros(3, np.float32)
        self.L.orc_dwa_get_oscillation(self.h, C.byref(f), prev)
        return f.value, prev

    def set_oscillation(self, flags, prev=(0, 0, 0)):
        self.L.orc_dwa_set_oscillation(self.h, int(flags), _f32(prev))


class TrajectoryPlanner:
    """Legacy base_local_planner::TrajectoryPlanner oracle (trajectory_planner_oracle.hpp).  `cfg`, result, state and
    sample records use the product's POD layouts (ctypes mirrors passed in by the tests) - layouts only."""

    def __init__(self, grid, res, cfg, footprint, ox=0.0, oy=0.0):
        self.L = lib()
        g = np.ascontiguousarray(grid, np.uint8)
        self.shape = g.shape
        fp = _f64(footprint).reshape(-1, 2)
        self.h = self.L.orc_tp_create(g.shape[1], g.shape[0], res, ox, oy, g, C.addressof(cfg), fp, len(fp))

    def __del__(self):
        try:
            self.L.orc_tp_destroy(self.h)
        except Exception:
            pass

    def set_costmap(self, grid):
        self.L.orc_tp_set_costmap(self.h, np.ascontiguousarray(grid, np.uint8))

    def update_plan(self, plan_xy, compute_dists=False):
        plan = _f64(plan_xy).reshape(-1, 2)
        self.L.orc_tp_update_plan(self.h, plan.ctypes.data if len(plan) else None, len(plan), int(compute_dists))

    def find_best_path(self, pos, vel, result_type, sample_type, traj_cap=1024, sample_cap=4096):
        res = result_type()
        traj = np.zeros((traj_cap, 3), np.float64)
        samples = (sample_type * sample_cap)()
        n = self.L.orc_tp_find_best_path(self.h, _f32(pos), _f32(vel), C.addressof(res), traj.ctypes.data, traj_cap,
                                         C.addressof(samples), sample_cap)
        return res, traj[:res.n_points].copy(), [(s.vx, s.vy, s.vtheta, s.cost, s.n_points) for s in samples[:n]]

    def score_trajectory(self, pose, vel, vel_samples):
        return self.L.orc_tp_score_trajectory(self.h, _f64(pose), _f64(vel), _f64(vel_samples))

    def generate(self, pose, vel, vel_samples, acc, impossible_cost):
        return self.L.orc_tp_generate(self.h, _f64(pose), _f64(vel), _f64(vel_samples), _f64(acc), float(impossible_cost))

    def grid(self, which):
        out = np.zeros(self.shape, np.float64)
        self.L.orc_tp_get_grid(self.h, which, out)
        return out

    def state(self, state_type):
        s = state_type()
        self.L.orc_tp_get_state(self.h, C.addressof(s))
        return s

    def set_state(self, s):
        self.L.orc_tp_set_state(self.h, C.addressof(s))


def footprint_cells(size_x, size_y, res, pos, footprint, fill=True, ox=0.0, oy=0.0, cap=65536):
    """FootprintHelper::getFootprintCells -> list of (x, y) cells in the reference's order."""
    L = lib()
    fp = _f64(footprint).reshape(-1, 2)
    out = np.zeros((cap, 2), np.int32)
    n = L.orc_tp_footprint_cells(size_x, size_y, res, ox, oy, _f32(pos), fp, len(fp), int(fill), out.ctypes.data, cap)
    return [tuple(int(v) for v in c) for c in out[:n]]


def min_max_distances(footprint):
    """costmap_2d::calculateMinAndMaxDistances (footprint.cpp:41-67) -> (inscribed, circumscribed)."""
    L = lib()
    fp = _f64(footprint).reshape(-1, 2)
    mn, mx = C.c_double(), C.c_double()
    L.orc_min_max_distances(fp, len(fp), C.byref(mn), C.byref(mx))
    return mn.value, mx.value


def navfn_plan(cmap, goal, start, cost_mode=1, allow_unknown=True, astar=False, at_start=True, want_potential=True):
    """navfn::NavFn (navfn_oracle.hpp): returns (path (n, 2) float32, potarr (ny, nx) float32 or None, cycles).
    cost_mode 0: cmap is costarr itself; 1: setCostmap(isROS=true); 2: setCostmap(isROS=false)."""
    g = np.ascontiguousarray(cmap, np.uint8)
    ny, nx = g.shape
    pot = np.zeros((ny, nx), np.float32) if want_potential else None
    cap = nx * ny // 2 + 4
    path = np.zeros((cap, 2), np.float32)
    cyc = C.c_int()
    n = lib().orc_navfn_plan(g, nx, ny, cost_mode, int(allow_unknown), np.ascontiguousarray(goal, np.int32), np.ascontiguousarray(start, np.int32),
                             int(astar), int(at_start), pot.ctypes.data if want_potential else None, path.ctypes.data, cap, C.byref(cyc))
    return path[:n].copy(), pot, cyc.value


def navfn_fixed_point(cmap, goal, start, cost_mode=1, allow_unknown=True):
    """The fixed point of NavFn::updateCell's rule over the whole map and calcPath on it (navfn_oracle.hpp propagateFixedPoint):
    returns (path (n, 2) float32, potarr (ny, nx) float32)."""
    g = np.ascontiguousarray(cmap, np.uint8)
    ny, nx = g.shape
    pot = np.zeros((ny, nx), np.float32)
    cap = nx * ny // 2 + 4
    path = np.zeros((cap, 2), np.float32)
    n = lib().orc_navfn_fixed_point(g, nx, ny, cost_mode, int(allow_unknown), np.ascontiguousarray(goal, np.int32), np.ascontiguousarray(start, np.int32),
                                    pot.ctypes.data, path.ctypes.data, cap)
    return path[:n].copy(), pot


GP_DEFAULTS = dict(use_dijkstra=1, use_quadratic=1, use_grid_path=0, old_navfn_behavior=0, allow_unknown=1, lethal_cost=253, neutral_cost=50,
                   outline_map=1, cost_factor=3.0)


def global_planner_plan(cmap, start_xy, goal_xy, goal_cell, fixed_point=False, **params):
    """global_planner (global_planner_oracle.hpp): the expansion + traceback of GlobalPlanner::makePlan on map coordinates.
    Returns (path (n, 2) float32 goal first, potential (ny, nx) float32, found_legal, cycles)."""
    pr = dict(GP_DEFAULTS)
    pr.update(params)
    g = np.ascontiguousarray(cmap, np.uint8)
    ny, nx = g.shape
    pot = np.zeros((ny, nx), np.float32)
    cap = 4 * nx * ny + 4
    path = np.zeros((min(cap, 1 << 22), 2), np.float32)
    legal, cyc = C.c_int(), C.c_int()
    ints = np.array([pr[k] for k in ("use_dijkstra", "use_quadratic", "use_grid_path", "old_navfn_behavior", "allow_unknown", "lethal_cost",
                                     "neutral_cost", "outline_map")] + [int(fixed_point)], np.int32)
    n = lib().orc_global_planner_plan(g, nx, ny, ints, float(pr["cost_factor"]), np.ascontiguousarray(start_xy, np.float64),
                                      np.ascontiguousarray(goal_xy, np.float64), np.ascontiguousarray(goal_cell, np.int32), pot.ctypes.data,
                                      path.ctypes.data, len(path), C.byref(legal), C.byref(cyc))
    return path[:n].copy(), pot, bool(legal.value), cyc.value
