import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get("WITH_TORCH"):
    import torch
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
import navigation_amd as nav
sys.argv = [sys.argv[0]]
import bench
fl, insts, cfg = bench.build_fleet(nav, 256, 400, 0)
poses_h, obs_arr, n_obs, pts_h, states_h, n_st, plans_h = bench.raw_inputs(fl)
for _ in range(3):
    bench.step(fl)
fl.sync()
def t(f, n=10):
    fl.sync(); t0 = time.perf_counter()
    for _ in range(n): f()
    fl.sync(); return (time.perf_counter() - t0) / n * 1e3
print("step only            %.3f ms" % t(lambda: bench.step(fl)))
print("stage_obs            %.3f ms" % t(lambda: fl.stage_observations_raw(poses_h, obs_arr, n_obs, pts_h)))
print("stage_planner        %.3f ms" % t(lambda: fl.stage_planner_raw(states_h, n_st, plans_h)))
print("results              %.3f ms" % t(lambda: fl.results()))
def full():
    fl.stage_observations_raw(poses_h, obs_arr, n_obs, pts_h); fl.stage_planner_raw(states_h, n_st, plans_h); bench.step(fl); fl.results()
print("full                 %.3f ms" % t(full))
from navigation_amd import _lib as N
raw = np.stack([i["cells"] for i in insts])
fl.upload(N.GRID_MASTER, raw)
fl.inflate(boxes=[[0, 0, 400, 400]] * 256)
fl.sync()
fl.profile(True); fl.profile_reset()
print("full after raw upload %.3f ms" % t(full))
print({k: round(v[0] / max(v[1], 1), 4) for k, v in fl.profile_read().items()})
fl.profile(False)
print("step only again       %.3f ms" % t(lambda: bench.step(fl)))
