"""Where does a k_score_sweep wave spend its time?  Needs an experiment build:
   make -C navigation_amd/csrc clean all EXTRA=-DNAVGPU_SWEEP_TIMING   (rebuild without EXTRA afterwards)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import navigation_amd as nav
L = nav.lib()
fn = C.CDLL(nav.lib_path()).navgpu_debug_sweep_stats
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
n = max(out[10], 1)
names = ["prologue up to the barrier", "prologue barrier", "lane setup", "sweep (incl. looking closer)", "  of which: looking closer", "wait: queue complete",
         "walk phase", "wait: walks done", "whole kernel"]
tick_ns = 10.0  # s_memtime: 100 MHz
for i, nm in enumerate(names):
    print("%-32s %8.2f us per wave" % (nm, out[i] / n * tick_ns / 1e3))
print("unscreened wave-steps per wave: %.2f   (waves sampled: %d)" % (out[9] / n, n))
