"""How often does k_score leave its screened path?  Needs an experiment build:
   make -C navigation_amd/csrc clean all EXTRA=-DNAVGPU_SCORE_STATS   (rebuild without EXTRA afterwards)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import navigation_amd as nav
L = nav.lib()
fn = C.CDLL(nav.lib_path()).navgpu_debug_score_stats
if len(sys.argv) > 1 and sys.argv[1] == "c5":  # configs[4]: 1000 x 1000, 5-vertex footprint, 64 x 64 x 32 samples
    from navigation_amd import _lib as N, synth
    R, n = 32, 1000
    fl = nav.Fleet(R, n, n, synth.RES, layers=N.LAYER_OBSTACLE | N.LAYER_INFLATION, max_points=1440, max_observations=1, max_sim_steps=24,
                   max_plan=256, max_footprint=8)
    fl.configure_obstacle()
    fl.set_footprint(synth.FOOTPRINT5)
    fl.configure_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, synth.inscribed_radius(synth.FOOTPRINT5))
    fl.configure_planner(nav.DwaConfig(vx_samples=64, vy_samples=64, vth_samples=32, sim_time=2.0, sim_granularity=0.1, discretize_by_time=1))
    insts = [synth.make_instance(n, 700 + i) for i in range(R)]
    fl.upload(N.GRID_MASTER, np.stack([i["cells"] for i in insts]))
    p5 = np.array([[float(v) for v in i["pos"]] for i in insts])
    fl.stage_observations(p5, [dict(instance=k, points=synth.laser_scan(i, 0), origin=(p5[k][0], p5[k][1], 0.3), obstacle_range=2.5,
                                    raytrace_range=3.0) for k, i in enumerate(insts)])
    fl.stage_planner(np.stack([i["pos"] for i in insts]), np.stack([i["vel"] for i in insts]), np.stack([i["plan"] for i in insts]))
    fl.set_plan()
    poses = None
else:
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
print("image load split: staging + barrier 1 %.2f us, lane mapping + barrier 2 %.2f us, copy + barrier 3 %.2f us" % (v[22] / w / 100, v[23] / w / 100, (v[16] - v[22] - v[23]) / w / 100))
