"""Tiled wavefront mode of navfn (navgpu_navfn_plan_wavefront) on the reference's willow map: time per plan, rounds, and how
its potentials / paths sit against the oracle's fixed point and the reference-order array.   python3 tools/probe_navfn_wavefront.py"""
import gzip
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import navigation_amd as nav  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402  (probe = test infrastructure)
import test_navfn as T  # noqa: E402

raw = gzip.open(os.path.join(ROOT, "tests", "golden", "willow_costmap.pgm.gz")).read()
_, w, h, _, data = raw.split(b"\n", 4)
nx, ny = int(w), int(h)
willow = np.frombuffer(data, np.uint8)[:nx * ny].reshape(ny, nx).copy()
cases = T.WILLOW_CASES + [((900, 1000), (350, 450)), ((150, 200), (1000, 1100))]
nf = nav.NavFn(nx, ny, 1)
nf.set_costmap(willow, cost_mode=0)
for start, goal in cases:
    for mode in ("wavefront", "reference_order"):
        for at_start in (True, False):
            fn = (lambda: nf.plan_wavefront([goal], [start], at_start=at_start)) if mode == "wavefront" else (lambda: nf.plan([goal], [start], at_start=at_start))
            fn()
            t = []
            for _ in range(3):
                t0 = time.perf_counter()
                res = fn()
                t.append((time.perf_counter() - t0) * 1e3)
            print(f"{mode:16s} start {start} goal {goal} at_start {int(at_start)}: {min(t):8.2f} ms  found {res[0].found} path {res[0].path_length} "
                  f"cycles/rounds {res[0].cycles} P(start) {res[0].start_potential:.3f}", flush=True)
    nf.plan_wavefront([goal], [start])
    g, gp = nf.potential(0), nf.path(0)
    fpath, fpot = orc.navfn_fixed_point(willow, goal, start, cost_mode=0)
    path, pot, _ = orc.navfn_plan(willow, goal, start, cost_mode=0)
    ps = g[start[1], start[0]]
    settled = fpot < ps
    rel = np.abs(g[settled] - fpot[settled]) / np.maximum(fpot[settled], 1.0)
    print(f"   settled cells {settled.sum()}: differ from the FIFO fixed point {int((rel > 0).sum())}, max rel {rel.max():.3g}; "
          f"above the reference-order array: {int((g[pot < 1e9] > pot[pot < 1e9]).sum())}", flush=True)
    if len(gp) and len(fpath):
        ca = willow.astype(np.int32)
        print(f"   path: {len(gp)} points (fixed point {len(fpath)}, reference order {len(path)}); Hausdorff to fixed point "
              f"{T._hausdorff(gp, fpath):.3f}, to reference order {T._hausdorff(gp, path) if len(path) else float('nan'):.3f} cells; cost "
              f"{T._path_cost(gp, ca):.1f} / {T._path_cost(fpath, ca):.1f} / {T._path_cost(path, ca) if len(path) else float('nan'):.1f}", flush=True)
nf.close()
