"""Host time of the stream-group loop, per call:  python3 tools/probe_host_time.py [groups] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import navigation_amd as nav  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = 256 // G
groups = []
for q in range(G):
    fl, insts, _ = bench.build_fleet(nav, n, 400, seed0=q * n)
    groups.append(bench.Group(nav, fl, insts, seed=4242 + q))
k = bench.run_cycles(groups, 0, 5)
acc = {"collect": 0.0, "stage_obs": 0.0, "stage_plan": 0.0, "update_map": 0.0, "planner_cycle": 0.0}
t_all = time.perf_counter()
for kk in range(k, k + steps):
    for g in groups:
        t0 = time.perf_counter()
        g.collect()
        t1 = time.perf_counter()
        arr, pts = g.scans[kk % bench.SCAN_CYCLES]
        g.fl.stage_observations_raw(g.poses_h, arr, g.n, pts)
        t2 = time.perf_counter()
        g.fl.stage_planner_raw(g.states[kk % len(g.states)], g.n, g.plans_pk)
        t3 = time.perf_counter()
        g.fl.update_map()
        t4 = time.perf_counter()
        g.fl.planner_cycle()
        t5 = time.perf_counter()
        g.pending = True
        for name, d in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[name] += d
for g in groups:
    g.collect()
dt = time.perf_counter() - t_all
print("groups", G, "ms_per_step", dt / steps * 1e3)
print("host us per group-cycle:", {a: round(b / steps / G * 1e6, 1) for a, b in acc.items()}, "sum without collect",
      round(sum(b for a, b in acc.items() if a != "collect") / steps / G * 1e6, 1))
