"""How many BFS levels do the bench fleet's MapGrid wavefronts take, whole-grid and bounded? (run on the GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NAVGPU_DEBUG_RAW_GRIDS"] = "1"
import navigation_amd as nav
from navigation_amd import _lib as N
import bench
n = int(os.environ.get("PROBE_N", "256"))
fl, insts, cfg = bench.build_fleet(nav, n, 400, 1000)
bench.step(fl); fl.sync()   # costmap update + one planner cycle: the maps now hold the static obstacles, the scan and the inflation
L = fl.L
import ctypes as C
def kbfs_ms():
    t = C.c_double(); k = C.c_uint64()
    L.navgpu_profile_read(fl.h, N.K_BFS if hasattr(N, "K_BFS") else 3, C.byref(t), C.byref(k))
    return t.value, k.value
for bounded in (0, 1):
    fl.set_bounded_map_grids(bounded)
    fl.stage_planner(np.stack([i["pos"] for i in insts]), np.stack([i["vel"] for i in insts]), np.stack([i["plan"] for i in insts]))
    L.navgpu_profile_enable(fl.h, 1); L.navgpu_profile_reset(fl.h)
    for _ in range(5):
        fl.planner_cycle()
    fl.sync()
    print("bounded", bounded, "k_bfs profile", kbfs_ms())
    for gname, gid in (("path", N.GRID_PATH), ("goal", N.GRID_GOAL), ("goal_front", N.GRID_GOAL_FRONT)):
        d = fl.download(gid).reshape(n, -1)
        lv = np.array([int(x[x < 160000].max()) for x in d])
        unre = np.array([int((x == 160001).sum()) for x in d])
        print("   hist", np.histogram(lv, bins=[0, 100, 200, 300, 400, 500, 700])[0])
        print("bounded", bounded, gname, "levels mean %.0f min %d max %d; unreachable-valued cells mean %.0f" % (lv.mean(), lv.min(), lv.max(), unre.mean()))
    if bounded:
        # open (free, unreached) cells inside the box of robot 0 in the path grid
        m = fl.download(N.GRID_MASTER).reshape(n, 400, 400)
        d = fl.download(N.GRID_PATH).reshape(n, 400, 400)
        for r in range(4):
            cx, cy = int(insts[r]["pos"][0] / 0.05), int(insts[r]["pos"][1] / 0.05)
            b = 30
            sub_m = m[r, cy - b:cy + b + 1, cx - b:cx + b + 1]
            sub_d = d[r, cy - b:cy + b + 1, cx - b:cx + b + 1]
            free = (sub_m < 253)
            print("robot", r, "box free cells", int(free.sum()), "of them unreached", int((free & (sub_d == 160001)).sum()), "lethal/inscribed in map", int((m[r] >= 253).sum()))
