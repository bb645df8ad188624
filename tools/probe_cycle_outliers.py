"""Which call of which cycle is the once-per-run outlier of the per-cycle host times (bench.py pcie_inclusive.cycle_ms_max)?
   python tools/probe_cycle_outliers.py [cycles in flight = 2]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
import navigation_amd as nav
from navigation_amd.sharding import shard_range as split
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 2
preburn = sys.argv[2] if len(sys.argv) > 2 else ""  # "tiny" / "big": that many async copies through torch first (is the stall count- or byte-bound?)
if preburn:
    import torch
    n = 4 if preburn == "tiny" else 553000
    src = torch.zeros(n, dtype=torch.uint8).pin_memory()
    dst = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    st = torch.cuda.Stream()
    t0 = time.perf_counter()
    worst = 0.0
    with torch.cuda.stream(st):
        for i in range(3000):
            t1 = time.perf_counter()
            dst.copy_(src, non_blocking=True)
            worst = max(worst, time.perf_counter() - t1)
    st.synchronize()
    print("preburn %s: 3000 copies in %.1f ms, slowest call %.3f ms" % (preburn, (time.perf_counter() - t0) * 1e3, worst * 1e3))
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
