"""Where does a cycle's host time go?  Per-call host durations of the async timed loop (stage_poses / update_map / planner_cycle).
   python tools/probe_stage.py [torch]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
if len(sys.argv) > 1:
    import torch
    torch.cuda.set_device(0)
import navigation_amd as nav

fl, insts, cfg = bench.build_fleet(nav, 256, 400, 0)
_, _, pos_h, vel_h, plans_h = fl._bench_host_inputs
poses = bench.PoseSchedule(pos_h, vel_h, 64, 1)
for k in range(5):
    bench.step(fl, poses, k)
fl.sync()
for rep in range(2):
  for mode in ("fixed", "poses", "fixed+prof", "poses+prof", "poses64+prof"):
    fl.profile(False)
    if "prof" in mode:
        fl.profile_select(["k_score"])
        fl.profile(True)
        fl.profile_reset()
    t = np.zeros(4)
    t0 = time.perf_counter()
    K = 50
    for k in range(K):
        a = time.perf_counter()
        if mode.startswith("poses64"):
            fl.stage_poses(poses.pos[k % 64][:64], poses.vel[:64])
        elif mode.startswith("poses"):
            fl.stage_poses(poses.pos[k % 64], poses.vel)
        b = time.perf_counter()
        fl.update_map()
        c = time.perf_counter()
        fl.planner_cycle()
        d = time.perf_counter()
        t[:3] += (b - a, c - b, d - c)
    e = time.perf_counter()
    fl.sync()
    f = time.perf_counter()
    print(mode, "per step ms: stage %.3f update_map %.3f planner_cycle %.3f | loop %.3f total %.3f" % (*(t[:3] / K * 1e3), (e - t0) / K * 1e3, (f - t0) / K * 1e3), flush=True)
