"""Which call of which cycle is the once-per-run outlier of the per-cycle host times (bench.py pcie_inclusive.cycle_ms_max)?
   python tools/probe_cycle_outliers.py [cycles in flight = 2]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
import navigation_amd as nav
from navigation_amd.sharding import shard_range as split
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 2
groups = []
for gi in range(4):
    g0, gn = split(256, gi, 4)
    fl, insts, cfg = bench.build_fleet(nav, gn, 400, seed0=g0)
    groups.append(bench.Group(nav, fl, insts, seed=4242 + gi))
for g in groups:
    g.set_depth(depth)
kk = bench.run_cycles(groups, 0, 50)
import gc
gc.collect(); gc.disable()
rows = []
for _ in range(400):
    for gi, g in enumerate(groups):
        t = [time.perf_counter()]
        if g.depth < 2:
            g.collect()
        t.append(time.perf_counter())
        arr, pts = g.scans[kk % bench.SCAN_CYCLES]
        g.fl.stage_observations_raw(g.poses_h, arr, g.n, pts)
        t.append(time.perf_counter())
        g.fl.stage_planner_raw(g.states[kk % len(g.states)], g.n, g.plans_pk)
        t.append(time.perf_counter())
        g.fl.update_map()
        t.append(time.perf_counter())
        g.fl.planner_cycle()
        t.append(time.perf_counter())
        if g.depth >= 2 and g.pending:
            g.fl.results_previous_into(g.rbuf)
        t.append(time.perf_counter())
        g.pending = True
        rows.append((kk, gi) + tuple((t[i + 1] - t[i]) * 1e3 for i in range(6)))
    kk += 1
for g in groups:
    g.collect()
a = np.array([r[2:] for r in rows])
tot = a.sum(axis=1)
names = ["collect", "stage_obs", "stage_plan", "update_map", "planner_cycle", "results_previous"]
print("median per group-cycle (ms):", dict(zip(names, np.round(np.median(a, axis=0), 4))))
for i in np.argsort(-tot)[:4]:
    print("cycle %d group %d: %.3f ms =" % (rows[i][0], rows[i][1], tot[i]), dict(zip(names, np.round(a[i], 3))))
