#!/usr/bin/env python3
"""Generate the golden vectors of SURVEY 8(c) (G1-G6; G7 navfn and G8 global_planner for row f-4) under tests/golden/ from the CPU oracle.

The reference itself cannot be built in this image (its hot-path sources need ROS, Boost, Eigen and PCL headers), so these
vectors come from oracle/ - the restatement that IS pinned by every fixture the reference's own tests hold
(tests/test_oracle_reference_fixtures.py, tests/test_navfn.py).  They freeze its outputs on seeded inputs: a later edit
of the oracle that changes any byte fails tests/test_goldens.py on the CPU, and the HIP path is checked against the
same files on the GPU without the oracle in the loop.  Every file stores the seed and parameters it was made with.
    python tools/make_goldens.py        (rewrites tests/golden/g*.npz)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from navigation_amd import synth  # noqa: E402  (pure numpy data generation)
from oracle import pyoracle as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def random_map(rs, n, density, unknown_frac=0.0):
    g = np.zeros((n, n), np.uint8)
    g[rs.random_sample((n, n)) < density] = 254
    if unknown_frac:
        g[(rs.random_sample((n, n)) < unknown_frac) & (g == 0)] = 255
    m = (rs.random_sample((n, n)) < 0.02) & (g == 0)
    g[m] = rs.randint(1, 253, m.sum())
    return g


def g1_inflation():
    rs = np.random.RandomState(101)
    maps64 = np.stack([random_map(rs, 64, d, u) for d, u in ((0.01, 0.0), (0.05, 0.1), (0.02, 0.03))])
    map400 = random_map(rs, 400, 0.01)
    par = dict(res=0.05, radius=0.55, scaling=10.0, inscribed=0.2)
    out = dict(seed=101, maps64=maps64, map400=map400, **par)
    for name, maps in (("64", maps64), ("400", map400[None])):
        out["ref" + name] = np.stack([orc.inflate(m, 0.05, 0.55, 10.0, 0.2, exact=False) for m in maps])
        out["exact" + name] = np.stack([orc.inflate(m, 0.05, 0.55, 10.0, 0.2, exact=True) for m in maps])
    out["n_cells_ref_differs_from_exact_400"] = int((out["ref400"] != out["exact400"]).sum())
    np.savez_compressed(os.path.join(OUT, "g1_inflation.npz"), **out)


def inflated_instance(n, idx):
    ins = synth.make_instance(n, idx)
    ins["master"] = orc.inflate(ins["cells"], synth.RES, synth.INFLATION_RADIUS, synth.COST_SCALING, synth.inscribed_radius(synth.FOOTPRINT),
                                exact=True)
    return ins


def g2_mapgrid():
    out = dict(seed_instances="synth.make_instance(n, 300 + k)")
    for n, key in ((64, "64"), (400, "400")):
        ins = inflated_instance(n, 300 + n)
        plan = ins["plan"] if n == 400 else np.stack([np.linspace(0.4, 2.8, 30), 1.6 + 0.5 * np.sin(np.linspace(0, 3, 30))], 1)
        out["master" + key] = ins["master"]
        out["plan" + key] = plan
        for mode, name in ((0, "path"), (1, "goal")):
            g = orc.map_grid(ins["master"], synth.RES, 0.0, 0.0, plan, mode, allow_unknown=True)
            out[name + key] = g.astype(np.uint32)
    np.savez_compressed(os.path.join(OUT, "g2_mapgrid.npz"), **out)


def g3_rollout():
    out = {}
    for tag, n, kw, seed in (("cfg1", 200, dict(vx_samples=10, vy_samples=10, vth_samples=5, sim_time=1.0, sim_granularity=0.1, discretize_by_time=1), 310),
                             ("cfg2", 400, dict(vx_samples=32, vy_samples=32, vth_samples=16, sim_time=2.0, sim_granularity=0.1, discretize_by_time=1), 311)):
        ins = inflated_instance(n, seed)
        cfg = orc.DwaConfig(**kw)
        p = orc.DwaPlanner(ins["master"], synth.RES, 0.0, 0.0, cfg)
        p.set_plan()
        res, traj, cref, cfull, status = p.cycle(ins["pos"], ins["vel"], ins["plan"], synth.FOOTPRINT)
        out[tag + "_master"] = ins["master"]
        out[tag + "_pos"] = ins["pos"]
        out[tag + "_vel"] = ins["vel"]
        out[tag + "_plan"] = ins["plan"]
        out[tag + "_cfg"] = np.array([kw[k] for k in ("vx_samples", "vy_samples", "vth_samples", "sim_time", "sim_granularity", "discretize_by_time")], np.float64)
        out[tag + "_result"] = np.array([res.best_index, res.n_samples, res.n_scored, res.n_valid, res.n_points, res.oscillation_flags], np.int64)
        out[tag + "_winner"] = np.array([res.xv, res.yv, res.thetav, res.cost, *res.drive], np.float64)
        out[tag + "_traj"] = traj
        if tag == "cfg1":  # every sample of the small configuration
            out[tag + "_samples"] = p.samples()
            out[tag + "_status"] = status
            out[tag + "_cost_full"] = cfull
            out[tag + "_cost_ref"] = cref
    out["seed_instances"] = "synth.make_instance(n, 310 / 311)"
    np.savez_compressed(os.path.join(OUT, "g3_rollout.npz"), **out)


def g4_footprint():
    rs = np.random.RandomState(104)
    n = 80
    g = random_map(rs, n, 0.03, 0.02)
    size = n * 0.05
    poses = np.stack([rs.uniform(-0.2, size + 0.2, 64), rs.uniform(-0.2, size + 0.2, 64), rs.uniform(-np.pi, np.pi, 64)], 1)
    out = dict(seed=104, grid=g, poses=poses, res=0.05)
    for name, fp in (("fp4", synth.FOOTPRINT), ("fp5", synth.FOOTPRINT5)):
        for au in (0, 1):
            out[f"{name}_allow{au}_footprint_cost"] = np.array([orc.footprint_cost(g, 0.05, 0.0, 0.0, x, y, th, fp, allow_unknown=bool(au)) for x, y, th in poses])
            out[f"{name}_allow{au}_step_cost"] = np.array([orc.obstacle_step_cost(g, 0.05, 0.0, 0.0, x, y, th, fp, allow_unknown=bool(au)) for x, y, th in poses])
    np.savez_compressed(os.path.join(OUT, "g4_footprint.npz"), **out)


def g5_velocity_iterator():
    cases = [(-30.0, 30.0, 4), (0.0, 0.55, 32), (-0.1, 0.1, 32), (0.2, 0.2, 5), (-1.0, 1.0, 16), (-0.3, 0.0, 7), (0.0, 0.4, 1), (-0.05, 0.25, 10)]
    out = dict(cases=np.array(cases, np.float64))
    for k, (mn, mx, ns) in enumerate(cases):
        out[f"samples{k}"] = orc.velocity_samples(mn, mx, int(ns))
    # SimpleTrajectoryGenerator::initialise sample lists (x-outer, y, theta-inner), DWA window and goal-limited window
    for k, (use_dwa, vel) in enumerate(((1, (0.3, 0.0, 0.2)), (0, (0.1, 0.05, -0.4)), (1, (0.0, 0.0, 0.0)))):
        cfg = orc.DwaConfig(vx_samples=6, vy_samples=5, vth_samples=7, use_dwa=use_dwa)
        out[f"init{k}"] = orc.samples(cfg, (1.0, 1.0, 0.3), vel, (1.4, 1.2, 0.0))
        out[f"init{k}_args"] = np.array([use_dwa, *vel], np.float64)
    np.savez_compressed(os.path.join(OUT, "g5_velocity_iterator.npz"), **out)


def g6_voxel():
    n = 120
    ins = synth.make_instance(n, 320)
    o = orc.LayeredCostmap(True)
    o.resize(n, n, synth.RES, 0, 0)
    o.set_footprint(synth.FOOTPRINT5)
    o.add_voxel(z_voxels=10, origin_z=0.0, z_resolution=0.2, unknown_threshold=15, mark_threshold=0, max_obstacle_height=2.0)
    o.add_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, exact=True)
    o.set_footprint(synth.FOOTPRINT5)
    out = dict(seed_instance="synth.make_instance(120, 320)", n_cycles=3)
    pose = [float(v) for v in ins["pos"]]
    out["pose"] = np.array(pose)
    for cyc in range(3):
        pts = synth.laser_scan(ins, cyc, z=0.3, z_jitter=1.5)
        org = (pose[0], pose[1], 0.3 + 0.25 * cyc)
        o.clear_observations()
        o.add_observation(pts, origin=org, obstacle_range=2.5, raytrace_range=3.0)
        o.update_map(*pose)
        out[f"points{cyc}"] = pts
        out[f"origin{cyc}"] = np.array(org)
        out[f"voxels{cyc}"] = o.voxels()
        out[f"layer{cyc}"] = o.layer(2)
        out[f"master{cyc}"] = o.master()
        out[f"box{cyc}"] = o.bounds()
    np.savez_compressed(os.path.join(OUT, "g6_voxel.npz"), **out)


def planner_map(rs, n):
    """A costmap as the global planners see one: lethal blobs with an inscribed ring and a cost gradient around them."""
    g = np.zeros((n, n), np.uint8)
    for _ in range(n // 6):
        cx, cy, r = rs.randint(4, n - 4), rs.randint(4, n - 4), rs.randint(1, 4)
        g[cy - r:cy + r + 1, cx - r:cx + r + 1] = 254
    return orc.inflate(g, 0.05, 0.4, 6.0, 0.1, exact=True)


def g7_navfn():
    """navfn::NavFn: Dijkstra and A* potentials, cycle counts and gradient paths on 96 x 96 costmaps."""
    rs = np.random.RandomState(107)
    out = dict(seed=107)
    for k in range(3):
        cm = planner_map(rs, 96)
        start, goal = (rs.randint(6, 40), rs.randint(6, 90)), (rs.randint(56, 90), rs.randint(6, 90))
        for cell in (start, goal):
            cm[cell[1] - 1:cell[1] + 2, cell[0] - 1:cell[0] + 2] = 0
        out[f"costmap{k}"] = cm
        out[f"start{k}"] = np.array(start, np.int32)
        out[f"goal{k}"] = np.array(goal, np.int32)
        for astar in (0, 1):
            path, pot, cyc = orc.navfn_plan(cm, goal, start, astar=bool(astar), allow_unknown=True)
            out[f"path{k}_{astar}"] = path
            out[f"potential{k}_{astar}"] = pot
            out[f"cycles{k}_{astar}"] = np.int32(cyc)
    np.savez_compressed(os.path.join(OUT, "g7_navfn.npz"), **out)


G8_VARIANTS = [dict(), dict(use_quadratic=0, use_grid_path=1), dict(use_dijkstra=0, use_grid_path=1), dict(old_navfn_behavior=1),
               dict(use_dijkstra=0), dict(allow_unknown=0, cost_factor=0.55, neutral_cost=66)]


def g8_global_planner():
    """global_planner's expanders and tracebacks on a 96 x 96 costmap, fractional start / goal."""
    rs = np.random.RandomState(108)
    cm = planner_map(rs, 96)
    start, goal = np.array([14.3, 20.7]), np.array([80.6, 71.2])
    for x, y in (start, goal):
        cm[int(y) - 1:int(y) + 3, int(x) - 1:int(x) + 3] = 0
    out = dict(seed=108, costmap=cm, start=start, goal=goal, n_variants=len(G8_VARIANTS))
    for k, kw in enumerate(G8_VARIANTS):
        s, g = (np.floor(start), np.floor(goal)) if kw.get("old_navfn_behavior") else (start, goal)
        path, pot, legal, cyc = orc.global_planner_plan(cm, s, g, g.astype(np.int32), **kw)
        out[f"path{k}"] = path
        out[f"potential{k}"] = pot
        out[f"legal{k}"] = np.int32(legal)
        out[f"cycles{k}"] = np.int32(cyc)
    np.savez_compressed(os.path.join(OUT, "g8_global_planner.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for fn in (g1_inflation, g2_mapgrid, g3_rollout, g4_footprint, g5_velocity_iterator, g6_voxel, g7_navfn, g8_global_planner):
        fn()
        print("wrote", fn.__name__)
