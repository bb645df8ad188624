"""NavFn: thin Python handle over navgpu_navfn_* — navfn::NavFn (navfn/src/navfn.cpp) for a batch of plans on one GPU.

  set_costmap   NavFn::setCostmap                        (navfn.cpp:222-283)
  plan          NavFn::setGoal / setStart + calcNavFnDijkstra | calcNavFnAstar   (navfn.cpp:145-171, 293-345)
  path          NavFn::getPathX / getPathY / getPathLen
  potential     NavFn::potarr
All compute happens in libnavgpu.so on the GPU; this file only marshals numpy buffers.
"""
import ctypes as C

import numpy as np

from ._lib import GlobalPlannerParams, NavfnResult, check, lib


class NavFn:
    def __init__(self, nx, ny, n_plans=1, device=0):
        self.L = lib()
        self.nx, self.ny, self.n = nx, ny, n_plans
        h = C.c_void_p()
        check(self.L.navgpu_navfn_create(nx, ny, n_plans, device, C.byref(h)), "navgpu_navfn_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.navgpu_navfn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_costmap(self, cmap, first=0, count=None, cost_mode=1, allow_unknown=True):
        """cmap: (ny, nx) shared by `count` plans, or (count, ny, nx).  cost_mode 1: costmap_2d values (isROS), 2: plain PGM,
        0: the bytes are costarr itself."""
        a = np.ascontiguousarray(cmap, np.uint8)
        shared = a.ndim == 2
        count = (self.n - first) if count is None else count
        if not shared:
            assert a.shape[0] == count
        assert a.shape[-2:] == (self.ny, self.nx)
        check(self.L.navgpu_navfn_set_costmap(self.h, first, count, a.ctypes.data_as(C.c_void_p), int(shared), cost_mode, int(allow_unknown)),
              "navfn_set_costmap")

    def set_costmap_from_fleet(self, fleet, first=0, count=None, fleet_first=0, allow_unknown=True):
        count = (self.n - first) if count is None else count
        check(self.L.navgpu_navfn_set_costmap_from_fleet(self.h, first, count, fleet.h, fleet_first, int(allow_unknown)), "navfn_set_costmap_from_fleet")

    def plan(self, goals, starts, first=0, astar=False, at_start=True):
        g = np.ascontiguousarray(goals, np.int32).reshape(-1, 2)
        s = np.ascontiguousarray(starts, np.int32).reshape(-1, 2)
        assert len(g) == len(s)
        res = (NavfnResult * len(g))()
        check(self.L.navgpu_navfn_plan(self.h, first, len(g), g.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), int(astar), int(at_start),
                                       C.cast(res, C.c_void_p)), "navfn_plan")
        return list(res)

    def plan_wavefront(self, goals, starts, first=0, at_start=True):
        """navgpu_navfn_plan_wavefront: the expansion as a tiled wavefront (the update rule's fixed point; Dijkstra only)."""
        g = np.ascontiguousarray(goals, np.int32).reshape(-1, 2)
        s = np.ascontiguousarray(starts, np.int32).reshape(-1, 2)
        assert len(g) == len(s)
        res = (NavfnResult * len(g))()
        check(self.L.navgpu_navfn_plan_wavefront(self.h, first, len(g), g.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), int(at_start),
                                                 C.cast(res, C.c_void_p)), "navfn_plan_wavefront")
        return list(res)

    def global_planner_plan(self, starts_xy, goals_xy, goal_cells, first=0, wavefront=False, **params):
        """GlobalPlanner::makePlan's expansion + traceback (map coordinates; costs set with cost_mode=0).  wavefront: the
        Dijkstra expansion as the tiled wavefront (navgpu_global_planner_plan_wavefront)."""
        st = np.ascontiguousarray(starts_xy, np.float64).reshape(-1, 2)
        gl = np.ascontiguousarray(goals_xy, np.float64).reshape(-1, 2)
        gc = np.ascontiguousarray(goal_cells, np.int32).reshape(-1, 2)
        gp = GlobalPlannerParams(**params)
        res = (NavfnResult * len(st))()
        fn = self.L.navgpu_global_planner_plan_wavefront if wavefront else self.L.navgpu_global_planner_plan
        check(fn(self.h, first, len(st), C.byref(gp), st.ctypes.data_as(C.c_void_p), gl.ctypes.data_as(C.c_void_p),
                 gc.ctypes.data_as(C.c_void_p), C.cast(res, C.c_void_p)), "global_planner_plan")
        return list(res)

    def path(self, plan=0):
        n = check(self.L.navgpu_navfn_path(self.h, plan, None, 0), "navfn_path")
        out = np.zeros((max(n, 1), 2), np.float32)
        if n:
            check(self.L.navgpu_navfn_path(self.h, plan, out.ctypes.data_as(C.c_void_p), n), "navfn_path")
        return out[:n].copy()

    def potential(self, plan=0):
        out = np.zeros((self.ny, self.nx), np.float32)
        check(self.L.navgpu_navfn_potential(self.h, plan, out.ctypes.data_as(C.c_void_p)), "navfn_potential")
        return out
