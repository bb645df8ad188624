"""The stream-group loop with one HOST THREAD per group (each hands over, queues and collects its own group's cycles; ctypes drops the GIL
inside the library):  python3 tools/probe_threads.py [groups] [steps] [threads: 0 = one thread round-robin, as bench.py's default]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import navigation_amd as nav  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
threaded = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n = 256 // G
groups = []
for q in range(G):
    fl, insts, _ = bench.build_fleet(nav, n, 400, seed0=q * n)
    groups.append(bench.Group(nav, fl, insts, seed=4242 + q))
k0 = bench.run_cycles(groups, 0, 5)


def worker(g):
    for k in range(k0, k0 + steps):
        g.cycle(k)
    g.collect()


for g in groups:
    g.fl.sync()
t0 = time.perf_counter()
if threaded:
    th = [threading.Thread(target=worker, args=(g,)) for g in groups]
    for t in th:
        t.start()
    for t in th:
        t.join()
else:
    bench.run_cycles(groups, k0, steps)
for g in groups:
    g.fl.sync()
dt = time.perf_counter() - t0
print("groups", G, "threads" if threaded else "one thread", "HW queues", os.environ.get("GPU_MAX_HW_QUEUES", "default"), "ms_per_step %.4f" % (dt / steps * 1e3),
      "trajectories/s %.3e" % (sum(g.scored for g in groups) / dt), flush=True)
