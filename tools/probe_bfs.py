import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import navigation_amd as nav
from navigation_amd import _lib as N, synth
sys.argv = [sys.argv[0]]
import bench
for n_inst in (1, 85, 256):
    fl, insts, cfg = bench.build_fleet(nav, n_inst, 400, 0)
    for _ in range(2): bench.step(fl)
    fl.sync(); fl.profile(True); fl.profile_reset()
    for _ in range(5): bench.step(fl)
    fl.sync()
    pr = fl.profile_read()
    lv = []
    for gid in (N.GRID_PATH, N.GRID_GOAL, N.GRID_GOAL_FRONT):
        g = fl.download(gid, 0, min(n_inst, 4)).astype(np.int64)
        g[g >= 160000] = -1
        lv.append([int(x.max()) for x in g])
    print(n_inst, {k: round(v[0] / max(v[1], 1), 4) for k, v in pr.items()}, "levels", lv)
    fl.close()
