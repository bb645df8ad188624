"""Does running the fleet as G groups on G streams overlap the latency-bound wavefronts with the issue-bound scoring?
   python tools/probe_overlap.py [groups ...]        (env NAVGPU_BFS_WGS_PER_CU=1|2)
Free-running steps (no sync inside the timed region), 256 robots in all, the contract workload of bench.py."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import navigation_amd as nav  # noqa: E402
import bench  # noqa: E402

groups = [int(a) for a in sys.argv[1:]] or [1, 2, 4]
for G in groups:
    n = 256 // G
    fleets = []
    for g in range(G):
        fl, insts, cfg = bench.build_fleet(nav, n, 400, seed0=g * n)
        _, _, pos_h, vel_h, _ = fl._bench_host_inputs
        fleets.append((fl, bench.PoseSchedule(pos_h, vel_h, 64, seed=4242 + g)))
    kk = 0
    for _ in range(5):
        for fl, ps in fleets:
            bench.step(fl, ps, kk)
        kk += 1
    for fl, _ in fleets:
        fl.sync()
    for rep in range(3):
        K = 50
        t0 = time.perf_counter()
        for _ in range(K):
            for fl, ps in fleets:
                bench.step(fl, ps, kk)
            kk += 1
        for fl, _ in fleets:
            fl.sync()
        dt = (time.perf_counter() - t0) / K
        print(f"groups {G} x {n} robots, BFS WGs/CU {os.environ.get('NAVGPU_BFS_WGS_PER_CU', '2')}: {dt * 1e3:.3f} ms per step", flush=True)
    lat = []
    for _ in range(30):
        t0 = time.perf_counter()
        for fl, ps in fleets:
            bench.step(fl, ps, kk)
        kk += 1
        for fl, _ in fleets:
            fl.sync()
        lat.append(time.perf_counter() - t0)
    lat.sort()
    print(f"   synchronised every cycle: median {lat[len(lat) // 2] * 1e3:.3f} ms", flush=True)
    for fl, _ in fleets:
        fl.close()
