"""How many trajectory points of k_score_sweep look closer, are queued for a footprint walk (for legality only / for their cost),
fail?  Needs an experiment build:  make -C navigation_amd/csrc clean all EXTRA=-DNAVGPU_SWEEP_COUNTS   (rebuild without EXTRA afterwards)
   python tools/probe_sweep_counts.py [configs4]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import navigation_amd as nav
L = nav.lib()
fn = C.CDLL(nav.lib_path()).navgpu_debug_sweep_counts
out = (C.c_ulonglong * 16)()
if len(sys.argv) > 1 and sys.argv[1] == "configs4":  # (all launches of that leg, warm-up included: read the ratios)
    fn(out, 1)
    bench.configs4_leg(nav, 0, n_robots=64)
    fn(out, 0)
    K = 1
else:
    fl, insts, cfg = bench.build_fleet(nav, 256, 400, 0)
    _, _, pos_h, vel_h, plans_h = fl._bench_host_inputs
    poses = bench.PoseSchedule(pos_h, vel_h, 64, 1)
    for k in range(3):
        bench.step(fl, poses, k)
    fl.sync()
    fn(out, 1)
    K = 10
    for k in range(K):
        bench.step(fl, poses, 3 + k)
    fl.sync()
    fn(out, 0)
v = [x / K for x in out]
print("per launch: swept wave-steps %.0f (live lane-steps %.3g), free-run wave-steps %.0f (live lane-steps %.3g)" % (v[4], v[8], v[9], v[10]))
print("  wave-steps that look closer %.0f (%.1f %% of swept), lanes that do %.3g" % (v[3], 100 * v[3] / max(v[4], 1), v[0]))
print("  queued: legality only %.3g, for the cost %.3g (of those the last point's %.3g); stalls %.3g" % (v[1], v[2], v[11], v[12]))
print("  walked %.3g in %.0f wave-rounds (%.1f lanes each); failed %.3g; legality-only walks that were legal %.3g" % (v[5], v[7], v[5] / max(v[7], 1), v[6], v[13]))
