"""TEMP: per-workgroup start/end of k_bfs_wave in the bench fleet"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NAVGPU_DEBUG_BFS_TRACE"] = os.path.join(ROOT, "gpurun_out", "bfs_trace.txt")
import navigation_amd as nav
import bench
fl, insts, cfg = bench.build_fleet(nav, 256, 400, 0)
for _ in range(4):
    bench.step(fl)
fl.sync()
a = np.loadtxt(os.environ["NAVGPU_DEBUG_BFS_TRACE"], dtype=np.float64)
t0 = a[:, 1].min()
st = (a[:, 1] - t0) / 100.0  # us (100 MHz)
en = (a[:, 2] - t0) / 100.0
lv = a[:, 3]
print("makespan us", en.max(), "sum of durations / 256 us", (en - st).sum() / 256)
for y in range(3):
    s = slice(256 * y, 256 * y + 256)
    print("grid", y, "start min/mean/max", st[s].min(), st[s].mean(), st[s].max(), "dur mean/max", (en - st)[s].mean(), (en - st)[s].max(), "levels mean", lv[s].mean())
d = en - st
# duration vs levels fit
A = np.vstack([np.ones_like(lv), lv]).T
coef = np.linalg.lstsq(A, d, rcond=None)[0]
print("duration ~ %.1f us + %.3f us/level" % (coef[0], coef[1]))
p1 = (a[:, 4] - t0) / 100.0
p2 = (a[:, 5] - t0) / 100.0
z = (a[:, 6] - t0) / 100.0; sd = (a[:, 7] - t0) / 100.0; pp = (a[:, 8] - t0) / 100.0
print("zero LDS", (z - st).mean(), "seeds", (sd - z).mean(), "free words + init", (p1 - sd).mean(), "post-pass", (pp - p2).mean(), "decode", (en - pp).mean())
print("prologue mean us", (p1 - st).mean(), "levels mean us", (p2 - p1).mean(), "post+decode mean us", (en - p2).mean())
for y in range(3):
    s = slice(256 * y, 256 * y + 256)
    print("grid", y, "prologue %.1f (zero %.1f seeds %.1f init %.1f)  levels %.1f (%.3f us/level)  post+decode %.1f" %
          ((p1 - st)[s].mean(), (z - st)[s].mean(), (sd - z)[s].mean(), (p1 - sd)[s].mean(), (p2 - p1)[s].mean(), ((p2 - p1)[s] / np.maximum(lv[s], 1)).mean(), (en - p2)[s].mean()))
# how many items are in flight over time
ts = np.linspace(0, en.max(), 42)
print("in flight:", [int(((st <= t) & (en > t)).sum()) for t in ts])
