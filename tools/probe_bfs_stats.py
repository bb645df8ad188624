"""Where does a level of k_bfs_rows spend its time?  Needs an experiment build:
   make -C navigation_amd/csrc clean all EXTRA=-DNAVGPU_BFS_STATS   (rebuild without EXTRA afterwards)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import navigation_amd as nav
fn = C.CDLL(nav.lib_path()).navgpu_debug_bfs_stats
fl, insts, cfg = bench.build_fleet(nav, 256, 400, 0)
_, _, pos_h, vel_h, plans_h = fl._bench_host_inputs
poses = bench.PoseSchedule(pos_h, vel_h, 64, 1)
for k in range(3):
    bench.step(fl, poses, k)
fl.sync()
out = (C.c_ulonglong * 16)()
fn(out, 1)
K = 10
for k in range(K):
    bench.step(fl, poses, 3 + k)
fl.sync()
fn(out, 0)
v = [x / K for x in out]
wl = max(v[5], 1)
print("per launch: wave-levels %.3e (active %.1f%%), spins per wave-level %.2f, active words per active wave-level %.2f" % (v[5], 100 * v[7] / wl, v[6] / wl, v[8] / max(v[7], 1)))
print("shader clocks per wave-level: poll %.0f  halo+words %.0f  stores %.0f  publish %.0f  group end %.0f  | total %.0f" %
      (v[0] / wl, v[1] / wl, v[2] / wl, v[3] / wl, v[4] / wl, sum(v[:5]) / wl))
