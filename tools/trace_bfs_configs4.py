"""Per-item phases of the wavefront kernel on BASELINE configs[4] (1000 x 1000, 64 robots): NAVGPU_DEBUG_BFS_TRACE stamps."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NAVGPU_DEBUG_BFS_TRACE"] = os.path.join(ROOT, "gpurun_out", "bfs_trace_c4.txt")
import bench  # noqa: E402
import navigation_amd as nav  # noqa: E402

out = bench.configs4_leg(nav, 0, n_robots=64, steps=2)
print(out["kernel_ms"])
a = np.loadtxt(os.environ["NAVGPU_DEBUG_BFS_TRACE"], dtype=np.float64)
t0 = a[:, 1].min()
us = lambda c: (a[:, c] - t0) / 100.0  # 100 MHz
st, en, lv, p1, p2, z, sd, pp = us(1), us(2), a[:, 3], us(4), us(5), us(6), us(7), us(8)
print("makespan us", en.max(), "items", len(a), "levels mean / max", lv.mean(), lv.max())
print("zero %.1f  seeds %.1f  rows into registers %.1f  levels %.1f (%.3f us/level)  post-pass %.1f" %
      ((z - st).mean(), (sd - z).mean(), (p1 - sd).mean(), (p2 - p1).mean(), ((p2 - p1) / np.maximum(lv, 1)).mean(), (pp - p2).mean()))
k = int(np.argmax(en - st))
print("longest item %d: total %.1f  zero %.1f seeds %.1f init %.1f levels %.1f (%d) post %.1f" %
      (k, (en - st)[k], (z - st)[k], (sd - z)[k], (p1 - sd)[k], (p2 - p1)[k], lv[k], (pp - p2)[k]))
# experiment build only (make EXTRA=-DNAVGPU_BFS_STATS): where a level's time goes
import ctypes as C  # noqa: E402

lib = C.CDLL(nav.lib_path())
if hasattr(lib, "navgpu_debug_bfs_stats"):
    o = (C.c_ulonglong * 16)()
    lib.navgpu_debug_bfs_stats(o, 1)
    bench.configs4_leg(nav, 0, n_robots=64, steps=1)
    lib.navgpu_debug_bfs_stats(o, 0)
    v = list(o)
    wl = max(v[5], 1)
    print("wave-levels %.3e (active %.1f%%), live groups per active wave-level %.2f" % (v[5], 100 * v[7] / wl, v[8] / max(v[7], 1)))
    print("clocks per wave-level: groups %.0f  stores %.0f  publish %.0f  barrier + halo %.0f | total %.0f" %
          (v[1] / wl, v[2] / wl, v[3] / wl, v[4] / wl, sum(v[:5]) / wl))
