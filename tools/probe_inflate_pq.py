"""Reference-order inflation (priority_queue_order = 1) on the contract workload's per-cycle windows: ms per costmap update of 256 robots."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import navigation_amd as nav
from navigation_amd import _lib as N, synth
n_inst, n = 256, 400
fl = nav.Fleet(n_inst, n, n, synth.RES, layers=N.LAYER_STATIC | N.LAYER_OBSTACLE | N.LAYER_INFLATION, max_points=720, max_observations=1)
fl.configure_obstacle()
fl.set_footprint(synth.FOOTPRINT)
fl.configure_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, synth.inscribed_radius(synth.FOOTPRINT), priority_queue_order=True)
insts = [synth.make_instance(n, i) for i in range(n_inst)]
for i, ins in enumerate(insts):
    fl.add_static_map(np.where(ins["cells"] == 254, 100, 0).astype(np.int8), first=i, count=1)
poses = np.array([[float(v) for v in ins["pos"]] for ins in insts])
def stage(k):
    obs = [dict(instance=i, points=synth.laser_scan(ins, k), origin=(float(ins["pos"][0]), float(ins["pos"][1]), 0.3), obstacle_range=2.5, raytrace_range=3.0)
           for i, ins in enumerate(insts)]
    fl.stage_observations(poses, obs)
stage(0)
t0 = time.perf_counter(); fl.update_map(); fl.sync(); print("first update (whole map re-inflated): %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
for k in range(1, 4):
    stage(k)
    fl.sync()
    t0 = time.perf_counter(); fl.update_map(); fl.sync(); dt = time.perf_counter() - t0
    b = fl.bounds()
    print("cycle %d: %.2f ms per update of %d robots, window cells per robot %.0f" % (k, dt * 1e3, n_inst, ((b[:, 1] - b[:, 0]) * (b[:, 3] - b[:, 2])).mean()), flush=True)
fl.close()
