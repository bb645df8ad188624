"""What do the raw (bounded) MapGrids look like inside the scoring window?  NAVGPU_DEBUG_RAW_GRIDS=1 python tools/probe_goal_window.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import navigation_amd as nav
from navigation_amd import _lib as N
fl, insts, cfg = bench.build_fleet(nav, 8, 400, 0)
poses = None
if len(sys.argv) > 1:
    _, _, pos_h, vel_h, plans_h = fl._bench_host_inputs
    poses = bench.PoseSchedule(pos_h, vel_h, 64, 1)
for k in range(5):
    bench.step(fl, poses, k)
fl.sync()
m = fl.master()
lv = fl.wavefront_levels()
bx = fl.wavefront_boxes()
for gid, name in ((N.GRID_PATH, "path"), (N.GRID_GOAL, "goal"), (N.GRID_GOAL_FRONT, "goal_front")):
    g = fl.download(gid)
    for i in range(4):
        c, w = 200, 31
        win = g[i][c - w:c + w + 1, c - w:c + w + 1]
        mw = m[i][c - w:c + w + 1, c - w:c + w + 1]
        bad = win >= 160000
        print(name, i, "levels", lv[i], "box", bx[i], "cells >= N_obst in window", int(bad.sum()), "of which costmap >= 253:", int((bad & (mw >= 253)).sum()),
              "free:", int((bad & (mw < 253)).sum()), "N_unreach:", int((win == 160001).sum()))
        if name == "goal" and i == 0:
            ys, xs = np.nonzero(bad & (mw < 253))
            print("  free-but-failing cells (window coords) x range", xs.min() if len(xs) else None, xs.max() if len(xs) else None, "y range", ys.min() if len(ys) else None, ys.max() if len(ys) else None)
import ctypes as C
dbg = C.CDLL(nav.lib_path())
if hasattr(dbg, "navgpu_debug_prep_image"):
    buf = np.zeros(1 << 17, np.uint8)
    win, nb = C.c_uint32(), C.c_uint32()
    for i in range(2):
        dbg.navgpu_debug_prep_image(fl.h, i, buf.ctypes.data_as(C.c_void_p), len(buf), C.byref(win), C.byref(nb))
        w = win.value
        nw = (w + 31) // 32
        wb = (w * w + 15) & ~15
        bits = buf[wb:wb + 16 * w * nw].view(np.uint32).reshape(w, nw, 4)
        cell = np.zeros((4, w, w), bool)
        for k in range(4):
            for j in range(nw):
                for b in range(32):
                    x = 32 * j + b
                    if x < w:
                        cell[k, :, x] = (bits[:, j, k] >> b) & 1
        c, h = 200, w // 2
        for k, (gid, name) in enumerate(((N.GRID_PATH, "path"), (N.GRID_GOAL, "goal"))):
            g = fl.download(gid)[i][c - h:c + h + 1, c - h:c + h + 1] >= 160000
            print("robot", i, name, "win", w, "bits set", int(cell[2 + k].sum()), "grid says", int(g.sum()), "mismatch", int((cell[2 + k] != g).sum()))
        mw = m[i][c - h:c + h + 1, c - h:c + h + 1]
        print("   can-fail bits", int(cell[1].sum()), "not-free bits", int(cell[0].sum()), "lethal cells", int((mw == 254).sum()))
