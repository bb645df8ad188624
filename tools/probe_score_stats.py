"""How often does k_score leave its screened path?  Needs an experiment build:
   make -C navigation_amd/csrc clean all EXTRA=-DNAVGPU_SCORE_STATS   (rebuild without EXTRA afterwards)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import navigation_amd as nav
L = nav.lib()
fn = C.CDLL(nav.lib_path()).navgpu_debug_score_stats
fl, insts, cfg = bench.build_fleet(nav, 256, 400, 0)
_, _, pos_h, vel_h, plans_h = fl._bench_host_inputs
poses = bench.PoseSchedule(pos_h, vel_h, 64, 1)
for k in range(3):
    bench.step(fl, poses, k)
fl.sync()
out = (C.c_ulonglong * 24)()
fn(out, 1)
K = 10
for k in range(K):
    bench.step(fl, poses, 3 + k)
fl.sync()
fn(out, 0)
v = [x / K for x in out]
if v[0] == 0:
    v[0] = v[2] = 1e-9  # timing-only build (EXTRA=-DNAVGPU_SCORE_TIMING): no counters
print("per launch: lane-steps %.3e  unscreened lanes %.3e (%.1f%%)  wave-steps %.3e  waves w/ unscreened lane %.3e (%.1f%%)  of which last-step waves %.3e" %
      (v[0], v[1], 100 * v[1] / v[0], v[2], v[3], 100 * v[3] / v[2], v[6]))
print("walk lanes %.3e (%.1f%% of lane-steps)  waves with a walk %.3e (%.1f%% of wave-steps)" % (v[4], 100 * v[4] / v[0], v[5], 100 * v[5] / v[2]))
print("non-last lanes with: can-fail bit %.3e  path bit %.3e  goal bit %.3e  fwd margin %.3e | not screened at all: off map %.3e  outside window %.3e  screen off %.3e" % tuple(v[8:15]))
w, g = max(v[19], 1), max(v[21], 1)
print("per wave (10 ns ticks -> us): image load %.2f  setup+rollout %.2f  reduce+wait %.2f | per workgroup residence %.2f us, %d workgroups, %d waves" %
      (v[16] / w / 100, v[17] / w / 100, v[18] / w / 100, v[20] / g / 100, g, w))
