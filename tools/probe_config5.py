"""configs[4] shape on one GPU: R robots, 1000x1000 costmaps, voxel layer (10 z-voxels) + inflation, 5-vertex footprint,
64x64x32 samples, 20 steps.  Prints per-kernel milliseconds (HIP events) for bounded and whole-grid wavefronts."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import navigation_amd as nav
    from navigation_amd import _lib as N, synth
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    steps = 10
    fl = nav.Fleet(R, n, n, synth.RES, layers=N.LAYER_VOXEL | N.LAYER_INFLATION, track_unknown=False, max_points=1440, max_observations=1,
                   max_sim_steps=24, max_plan=256, max_footprint=8)
    fl.configure_obstacle(z_voxels=10, origin_z=0.0, z_resolution=0.2, unknown_threshold=15, mark_threshold=0, max_obstacle_height=2.0)
    fl.set_footprint(synth.FOOTPRINT5)
    fl.configure_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, synth.inscribed_radius(synth.FOOTPRINT5))
    fl.configure_planner(nav.DwaConfig(vx_samples=64, vy_samples=64, vth_samples=32, sim_time=2.0, sim_granularity=0.1, discretize_by_time=1))
    insts = [synth.make_instance(n, 700 + i) for i in range(R)]
    fl.upload(N.GRID_MASTER, np.stack([i["cells"] for i in insts]))
    poses = np.array([[float(v) for v in i["pos"]] for i in insts])
    obs = [dict(instance=k, points=synth.laser_scan(i, 0, z=0.3, z_jitter=1.5), origin=(poses[k][0], poses[k][1], 0.3), obstacle_range=2.5,
                raytrace_range=3.0) for k, i in enumerate(insts)]
    fl.stage_observations(poses, obs)
    fl.stage_planner(np.stack([i["pos"] for i in insts]), np.stack([i["vel"] for i in insts]), np.stack([i["plan"] for i in insts]))
    fl.set_plan()
    out = {}
    for mode in ("bounded", "whole"):
        if mode == "whole":
            fl.set_bounded_map_grids(False)
        for _ in range(2):
            fl.update_map()
            fl.planner_cycle()
        fl.sync()
        fl.profile(True)
        fl.profile_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            fl.update_map()
            fl.planner_cycle()
        fl.sync()
        dt = (time.perf_counter() - t0) / steps
        pr = fl.profile_read()
        fl.profile(False)
        t1 = time.perf_counter()
        for _ in range(steps):
            fl.update_map()
            fl.planner_cycle()
        fl.sync()
        dt1 = (time.perf_counter() - t1) / steps
        res = fl.results()
        out[mode] = {"ms_per_step_profiled": dt * 1e3, "ms_per_step": dt1 * 1e3, "trajectories_per_s": sum(r.n_scored for r in res) / dt1,
                     "kernel_ms": {k: round(v[0] / v[1], 4) for k, v in pr.items() if v[1]}, "robots": R, "map": n}
    print(json.dumps(out))
    fl.close()


if __name__ == "__main__":
    main()
