"""How many BFS levels do the bench fleet's MapGrid wavefronts take? (run on the GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import navigation_amd as nav
from navigation_amd import _lib as N
import bench
fl, insts, cfg = bench.build_fleet(nav, 64, 400, 1000)
bench.step(fl); fl.sync()
for gname, gid in (("path", N.GRID_PATH), ("goal", N.GRID_GOAL), ("goal_front", N.GRID_GOAL_FRONT)):
    d = fl.download(gid).reshape(64, -1)
    lv = np.array([int(x[x < 160000].max()) for x in d])
    print(gname, "levels mean %.0f min %d max %d" % (lv.mean(), lv.min(), lv.max()))
