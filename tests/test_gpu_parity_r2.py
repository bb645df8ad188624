"""Parity tests added in round 2 (same bar as test_gpu_parity.py: the HIP path through the C-ABI against the
CPU oracle, bit-exact on bytes / indices / codes, 1e-5 on cost floats).  They close the holes the round-1 review
named: footprints leaving a clean map border, the continued-acceleration generator, updateWithOverwrite, the
layer-granular ObstacleLayer::updateCosts, the inflation cost table as the device applies it, BASELINE configs[2]
at its full 256 robots with LaserScan cycles, and configs[4]'s voxel layer at 1000x1000."""
import numpy as np
import pytest

from test_gpu_parity import (FREE, INSCRIBED, LETHAL, NOINFO, L, _check_planner, _compare_cycle, _planner_pair, nav)  # noqa: F401

pytestmark = pytest.mark.gpu


# ----------------------------------------------------------------------------------------------
# CostmapModel::footprintCost at the map edge (costmap_model.cpp:74-101): a footprint vertex that
# fails worldToMap makes the point -1 -> ObstacleCostFunction -6, whatever the cells in reach hold.
# All-FREE maps: the only way to fail is through the border, and every shortcut would say "free".
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("poly5", [False, True])
@pytest.mark.parametrize("allow_unknown", [0, 1])
@pytest.mark.parametrize("sum_scores", [0, 1])
def test_planner_clean_border_offmap_vertices(nav, orc, poly5, allow_unknown, sum_scores):
    from navigation_amd import synth
    n = 120
    size = n * synth.RES
    fp = synth.FOOTPRINT5 if poly5 else synth.FOOTPRINT
    cfgk = dict(vx_samples=6, vy_samples=4, vth_samples=7, sim_time=1.2, sim_granularity=0.1, discretize_by_time=1,
                min_vel_x=-0.2, allow_unknown=allow_unknown, sum_scores=sum_scores)
    master = np.zeros((n, n), np.uint8)
    fl, p = _planner_pair(nav, orc, n, master, cfgk, fp)
    n_edge_fail = n_valid = 0
    mid = size / 2
    spots = []
    for d in (0.10, 0.15, 0.20, 0.35, 0.5):  # start illegal ... start legal, the border within the samples' reach
        spots += [(d, mid), (size - d, mid), (mid, d), (mid, size - d)]          # the four borders
    spots += [(0.12, 0.17), (size - 0.12, size - 0.17), (0.14, size - 0.11), (size - 0.19, 0.13)]  # the corners
    for k, (x, y) in enumerate(spots):
        # plan from the robot into the map's middle so that the MapGrids are reachable
        plan = np.stack([np.linspace(x, mid, 40), np.linspace(y, mid, 40)], 1)
        for yaw in (0.0, 0.6 + 0.37 * k, -2.2 + 0.11 * k):
            r, cost, st = _compare_cycle(fl, p, [x, y, yaw], [0.1, 0.0, 0.2], plan, fp)
            sc = st == 1
            n_edge_fail += int((cost[sc] == -6.0).sum())
            n_valid += int((cost[sc] >= 0).sum())
    assert n_edge_fail > 0 and n_valid > 0, "the border case was not reached"
    fl.close()


# ----------------------------------------------------------------------------------------------
# use_dwa = false: the goal/sim_time-limited sample window (simple_trajectory_generator.cpp:91-105) and
# continued acceleration inside the rollout (computeNewVelocities each step, :221-248)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("discretize_by_time", [0, 1])
@pytest.mark.parametrize("plan_len", [200, 12])  # goal 8 m away / 0.5 m away (the window is goal-limited then)
def test_planner_continued_acceleration(nav, orc, discretize_by_time, plan_len):
    kw = dict(use_dwa=0, vx_samples=7, vy_samples=5, vth_samples=9, sim_time=1.5, discretize_by_time=discretize_by_time)
    if discretize_by_time:
        kw["sim_granularity"] = 0.1
    _check_planner(nav, orc, 200, kw, n_inst=2, cycles=2, seed0=70, plan_len=plan_len)


def test_planner_continued_acceleration_sum_scores_polygon(nav, orc):
    from navigation_amd import synth
    _check_planner(nav, orc, 160, dict(use_dwa=0, vx_samples=5, vy_samples=4, vth_samples=6, sim_time=1.0, sim_granularity=0.1,
                                       discretize_by_time=1, sum_scores=1, min_vel_x=-0.2), n_inst=2, seed0=74,
                   footprint=synth.FOOTPRINT5, allow_unknown=0, unknown_frac=0.01)


# ----------------------------------------------------------------------------------------------
# combination_method = 0: ObstacleLayer::updateCosts -> updateWithOverwrite (costmap_layer.cpp:107-124)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("track_unknown", [False, True])
def test_costmap_cycles_overwrite(nav, orc, track_unknown):
    from navigation_amd import synth
    N = L(nav)
    n, nI = 200, 3
    insts = [synth.make_instance(n, 80 + i) for i in range(nI)]
    insc = synth.inscribed_radius(synth.FOOTPRINT)
    fl = nav.Fleet(nI, n, n, synth.RES, layers=N.LAYER_STATIC | N.LAYER_OBSTACLE | N.LAYER_INFLATION, max_points=720,
                   max_observations=2, track_unknown=track_unknown)
    fl.configure_obstacle(combination_method=0)
    fl.set_footprint(synth.FOOTPRINT)
    fl.configure_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, insc)
    oracles = []
    for i, ins in enumerate(insts):
        occ = np.where(ins["cells"] == 254, 100, 0).astype(np.int8)
        occ[5:30, 5:30] = -1  # unknown static area
        fl.add_static_map(occ, first=i, count=1)
        o = orc.LayeredCostmap(track_unknown)
        o.set_footprint(synth.FOOTPRINT)
        o.add_static(occ, res=synth.RES)
        o.add_obstacle(combination_method=0)
        o.add_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, exact=True)
        o.set_footprint(synth.FOOTPRINT)
        oracles.append(o)
    n_overwritten = 0
    for cyc in range(3):
        obs, poses = [], []
        for i, ins in enumerate(insts):
            pts = synth.laser_scan(ins, cyc)
            org = (float(ins["pos"][0]), float(ins["pos"][1]), 0.3)
            obs.append(dict(instance=i, points=pts, origin=org, obstacle_range=2.5, raytrace_range=3.0))
            poses.append([float(v) for v in ins["pos"]])
            oracles[i].clear_observations()
            oracles[i].add_observation(pts, origin=org, obstacle_range=2.5, raytrace_range=3.0)
            oracles[i].update_map(*poses[-1])
        fl.stage_observations(poses, obs)
        fl.update_map()
        m, ol, b = fl.master(), fl.download(N.GRID_OBSTACLE), fl.bounds()
        for i in range(nI):
            assert np.array_equal(b[i], oracles[i].bounds()), (cyc, i)
            assert np.array_equal(ol[i], oracles[i].layer(2)), (cyc, i)
            assert np.array_equal(m[i], oracles[i].master()), (cyc, i)
            # overwrite really differs from max somewhere: a raytraced FREE cell on top of a static lethal one
            st = np.where(insts[i]["cells"] == 254, 254, 0)
            n_overwritten += int(((ol[i] == FREE) & (st == 254)).sum())
    assert n_overwritten > 0 or not track_unknown
    fl.close()


# ----------------------------------------------------------------------------------------------
# navgpu_obstacle_update_costs = ObstacleLayer::updateCosts alone (obstacle_layer.cpp:427-448): the master it is
# handed already holds the earlier layers' output, which must survive outside what the layer grid overrides.
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("combination_method", [0, 1])
def test_obstacle_update_costs_keeps_earlier_layers(nav, orc, combination_method):
    N = L(nav)
    n = 96
    rs = np.random.RandomState(7 + combination_method)
    fl = nav.Fleet(2, n, n, 0.05, layers=N.LAYER_OBSTACLE, max_points=64, track_unknown=True)
    fl.configure_obstacle(combination_method=combination_method)
    masters, layers = [], []
    for k in range(2):
        m = rs.randint(0, 253, (n, n)).astype(np.uint8)  # "static layer" content already in the master
        m[rs.random_sample((n, n)) < 0.05] = NOINFO
        m[10:20, 10:60] = LETHAL
        lay = np.full((n, n), NOINFO, np.uint8)
        lay[rs.random_sample((n, n)) < 0.3] = FREE
        lay[rs.random_sample((n, n)) < 0.05] = LETHAL
        masters.append(m)
        layers.append(lay)
    fl.upload(N.GRID_MASTER, np.stack(masters))
    fl.upload(N.GRID_OBSTACLE, np.stack(layers))
    boxes = [[5, 7, 90, 80], [0, 0, n, n]]
    fl.obstacle_update_costs(boxes)
    got = fl.master()
    for k in range(2):
        x0, y0, xn, yn = boxes[k]
        exp = masters[k].copy()
        sub_m, sub_l = exp[y0:yn, x0:xn], layers[k][y0:yn, x0:xn]
        if combination_method == 0:   # updateWithOverwrite (costmap_layer.cpp:107-124)
            take = sub_l != NOINFO
        else:                          # updateWithMax (costmap_layer.cpp:62-85)
            take = (sub_l != NOINFO) & ((sub_m == NOINFO) | (sub_m < sub_l))
        sub_m[take] = sub_l[take]
        assert np.array_equal(got[k], exp), (combination_method, k)
    fl.close()


# ----------------------------------------------------------------------------------------------
# the inflation cost table AS THE DEVICE APPLIES IT: one seed, every (dx, dy) within and just beyond the radius
# against InflationLayer::computeCaches' cached_costs_ (inflation_layer.cpp:295-328)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("res,radius,scaling,insc", [(0.05, 0.55, 10.0, 0.2), (0.05, 0.7, 3.0, 0.3), (0.1, 1.0, 1.0, 0.0),
                                                    (0.025, 0.3, 25.0, 0.11), (0.05, 1.2, 5.0, 0.46)])
def test_inflate_cost_table_through_device(nav, orc, res, radius, scaling, insc):
    N = L(nav)
    R, costs, dists = orc.cost_lut(res, radius, scaling, insc)
    n = 2 * (R + 3) + 1
    c = n // 2
    g = np.zeros((n, n), np.uint8)
    g[c, c] = LETHAL
    fl = nav.Fleet(1, n, n, res, layers=N.LAYER_INFLATION)
    fl.configure_inflation(radius, scaling, insc)
    fl.upload(N.GRID_MASTER, g)
    fl.inflate(boxes=[[0, 0, n, n]])
    got = fl.master()[0]
    exp = np.zeros((n, n), np.uint8)
    for dy in range(-c, c + 1):
        for dx in range(-c, c + 1):
            ax, ay = abs(dx), abs(dy)
            if ax <= R + 1 and ay <= R + 1 and dists[ax, ay] <= R:  # enqueue's `distance > cell_inflation_radius_` test (:286)
                exp[c + dy, c + dx] = costs[ax, ay]
    assert np.array_equal(got, exp)
    assert np.array_equal(got, orc.inflate(g, res, radius, scaling, insc, exact=False))
    assert got[c, c] == LETHAL and (got[c, c + 1] == INSCRIBED) == (res <= insc)
    fl.close()


# ----------------------------------------------------------------------------------------------
# BASELINE configs[2] at full size: 256 robots on one GPU, 400x400 costmaps, a LaserScan update every cycle,
# 32x32x16 samples, 20 steps.  Oracle on 8 robots (all grids, winners), size-independent properties on all 256.
# ----------------------------------------------------------------------------------------------
def test_config2_full_fleet_256_with_scans(nav, orc):
    from navigation_amd import synth
    N = L(nav)
    n, nI, n_cyc = 400, 256, 2
    checked = (0, 37, 64, 101, 128, 190, 222, 255)
    insc = synth.inscribed_radius(synth.FOOTPRINT)
    fl = nav.Fleet(nI, n, n, synth.RES, layers=N.LAYER_STATIC | N.LAYER_OBSTACLE | N.LAYER_INFLATION, max_points=720,
                   max_observations=1, keep_sample_costs=True)
    fl.configure_obstacle()
    fl.set_footprint(synth.FOOTPRINT)
    fl.configure_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, insc)
    cfg = synth.fleet_config()
    fl.configure_planner(cfg)
    ocfg = orc.DwaConfig(**cfg.as_dict())
    insts = [synth.make_instance(n, i) for i in range(nI)]
    oracles = {}
    for i, ins in enumerate(insts):
        occ = np.where(ins["cells"] == 254, 100, 0).astype(np.int8)
        fl.add_static_map(occ, first=i, count=1)
        if i in checked:
            o = orc.LayeredCostmap(False)
            o.set_footprint(synth.FOOTPRINT)
            o.add_static(occ, res=synth.RES)
            o.add_obstacle()
            o.add_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, exact=True)
            o.set_footprint(synth.FOOTPRINT)
            oracles[i] = o
    planners = {}
    fl.set_plan()
    pos = np.stack([i["pos"] for i in insts]).copy()
    vel = np.stack([i["vel"] for i in insts]).copy()
    plans = np.stack([i["plan"] for i in insts])
    static_lethal = np.stack([i["cells"] for i in insts]) == 254
    for cyc in range(n_cyc):
        obs, poses = [], []
        for i, ins in enumerate(insts):
            pts = synth.laser_scan(ins, cyc)
            org = (float(ins["pos"][0]), float(ins["pos"][1]), 0.3)
            obs.append(dict(instance=i, points=pts, origin=org, obstacle_range=2.5, raytrace_range=3.0))
            poses.append([float(v) for v in ins["pos"]])
            if i in oracles:
                oracles[i].clear_observations()
                oracles[i].add_observation(pts, origin=org, obstacle_range=2.5, raytrace_range=3.0)
                oracles[i].update_map(*poses[-1])
        fl.stage_observations(poses, obs)
        fl.update_map()
        m = fl.master()
        b = fl.bounds()
        # properties, all 256: static lethal cells stay lethal (max-merge), no unknowns are created, every marked
        # point of the scan is lethal in the master, the update box is inside the map and not empty
        assert (m[static_lethal] == LETHAL).all() and (m != NOINFO).all()
        assert (b[:, 0] >= 0).all() and (b[:, 2] >= 0).all() and (b[:, 1] <= n).all() and (b[:, 3] <= n).all()
        assert (b[:, 1] > b[:, 0]).all() and (b[:, 3] > b[:, 2]).all()
        for i in (3, 77, 200):
            pts = obs[i]["points"]
            d2 = ((pts - np.array(obs[i]["origin"], np.float32)) ** 2).sum(1)
            near = pts[(d2 < 2.5 ** 2 - 1e-3) & (d2 > 0.6 ** 2)]  # beyond the footprint polygon that updateCosts clears
            cx, cy = (near[:, 0] / synth.RES).astype(int), (near[:, 1] / synth.RES).astype(int)
            assert (m[i][cy, cx] == LETHAL).all()
        for i, o in oracles.items():
            assert np.array_equal(b[i], o.bounds()), (cyc, i)
            assert np.array_equal(m[i], o.master()), (cyc, i)
        # planner on the updated costmaps
        p2 = pos.copy()
        p2[:, 2] += 0.3 * cyc
        res = fl.find_best_path(p2, vel, plans)
        for i in range(nI):
            cost, status, _ = fl.samples(i) if (i in oracles or i % 16 == 5) else (None, None, None)
            if cost is None:
                assert res[i].n_samples > 16384 and 0 <= res[i].n_valid <= res[i].n_scored <= res[i].n_samples
                continue
            ok = (status == 1) & (cost >= 0)
            assert res[i].n_valid == ok.sum() and res[i].n_scored == (status == 1).sum()
            if ok.any():
                assert res[i].best_index == int(np.flatnonzero(ok)[np.argmin(cost[ok])]) and res[i].cost == cost[ok].min()
            else:
                assert res[i].best_index == -1
        for i in oracles:
            if i not in planners:
                planners[i] = orc.DwaPlanner(m[i], synth.RES, 0.0, 0.0, ocfg)
                planners[i].set_plan()
            else:
                planners[i].set_costmap(m[i])
            ores, otraj, cref, cfull, ostatus = planners[i].cycle(p2[i], vel[i], plans[i], synth.FOOTPRINT)
            cost, status, vels = fl.samples(i)
            assert np.array_equal(status, ostatus)
            assert np.array_equal(vels.view(np.uint32), planners[i].samples().view(np.uint32))
            sc = status == 1
            assert np.array_equal(cost[sc] < 0, cfull[sc] < 0)
            assert np.array_equal(cost[sc][cost[sc] < 0], cfull[sc][cfull[sc] < 0])
            assert np.allclose(cost[sc][cost[sc] >= 0], cfull[sc][cfull[sc] >= 0], rtol=0, atol=1e-5)
            r = res[i]
            assert (r.best_index, r.n_valid, r.n_scored) == (ores.best_index, ores.n_valid, ores.n_scored)
            assert abs(r.cost - ores.cost) <= 1e-5 and r.oscillation_flags == ores.oscillation_flags
            for gid, which in ((N.GRID_PATH, 0), (N.GRID_GOAL, 1), (N.GRID_GOAL_FRONT, 2)):
                assert np.array_equal(fl.download(gid, i, 1)[0].astype(np.float64), planners[i].grid(which)), (cyc, i, which)
    fl.close()


# ----------------------------------------------------------------------------------------------
# BASELINE configs[4] on one robot: 1000x1000 costmap with the voxel_grid 3-D obstacle layer + inflation, then the
# 64x64x32-sample planner with the 5-vertex polygon footprint on the costmap the voxel layer produced.
# ----------------------------------------------------------------------------------------------
def test_config4_voxel_layer_1000x1000_and_planner(nav, orc):
    from navigation_amd import synth
    N = L(nav)
    n = 1000
    ins = synth.make_instance(n, 90)
    insc = synth.inscribed_radius(synth.FOOTPRINT5)
    fl = nav.Fleet(1, n, n, synth.RES, layers=N.LAYER_VOXEL | N.LAYER_INFLATION, track_unknown=False, max_points=1440,
                   max_observations=2, keep_sample_costs=True, max_sim_steps=64, max_plan=256)
    fl.configure_obstacle(z_voxels=10, origin_z=0.0, z_resolution=0.2, unknown_threshold=15, mark_threshold=0, max_obstacle_height=2.0)
    fl.set_footprint(synth.FOOTPRINT5)
    fl.configure_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, insc)
    o = orc.LayeredCostmap(False)
    o.resize(n, n, synth.RES, 0, 0)
    o.set_footprint(synth.FOOTPRINT5)
    o.add_voxel(z_voxels=10, origin_z=0.0, z_resolution=0.2, unknown_threshold=15, mark_threshold=0, max_obstacle_height=2.0)
    o.add_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, exact=True)
    o.set_footprint(synth.FOOTPRINT5)
    pose = [float(v) for v in ins["pos"]]
    for cyc in range(3):
        pts = synth.laser_scan(ins, cyc, z=0.3, z_jitter=1.5)
        org = (pose[0], pose[1], 0.3 + 0.2 * cyc)
        # long ranges so that marks and 3-D rays cover a large part of the 50 m map
        obs = [dict(instance=0, points=pts, origin=org, obstacle_range=9.0, raytrace_range=9.5)]
        o.clear_observations()
        o.add_observation(pts, origin=org, obstacle_range=9.0, raytrace_range=9.5)
        o.update_map(*pose)
        fl.stage_observations([pose], obs)
        fl.update_map()
        assert np.array_equal(fl.download(N.GRID_VOXEL)[0], o.voxels()), ("voxel columns", cyc)
        assert np.array_equal(fl.download(N.GRID_OBSTACLE)[0], o.layer(2)), ("voxel layer 2-D grid", cyc)
        assert np.array_equal(fl.bounds()[0], o.bounds()), ("box", cyc)
        assert np.array_equal(fl.master()[0], o.master()), ("master", cyc)
    m = fl.master()[0]
    assert (m == LETHAL).sum() > 50
    cfg = nav.DwaConfig(vx_samples=64, vy_samples=64, vth_samples=32, sim_time=2.0, sim_granularity=0.1, discretize_by_time=1)
    fl.configure_planner(cfg)
    p = orc.DwaPlanner(m, synth.RES, 0.0, 0.0, orc.DwaConfig(**cfg.as_dict()))
    r, cost, st = _compare_cycle(fl, p, ins["pos"], ins["vel"], ins["plan"], synth.FOOTPRINT5)
    assert r.n_samples > 131072
    fl.close()


# ----------------------------------------------------------------------------------------------
# navgpu_planner_stage_poses: a cycle with an unchanged plan stages pose + velocity only and must give what a full
# stage of the same inputs gives (= the oracle's cycle).  More cycles than the staging ring has slots, a full stage in
# between, robots near the end of their plan (whole-grid searches) and far from it (bounded ones).
# ----------------------------------------------------------------------------------------------
def test_stage_poses_equals_full_stage(nav, orc):
    from navigation_amd import synth
    from test_gpu_parity import _inflated_instance
    N = L(nav)
    n, nI = 200, 3
    cfg = nav.DwaConfig(vx_samples=8, vy_samples=6, vth_samples=9, sim_time=1.5, sim_granularity=0.1, discretize_by_time=1)
    ocfg = orc.DwaConfig(**cfg.as_dict())
    insts = [_inflated_instance(orc, n, 120 + i, synth) for i in range(nI)]
    fl = nav.Fleet(nI, n, n, synth.RES, layers=N.LAYER_OBSTACLE, keep_sample_costs=True, max_sim_steps=64)
    fl.configure_planner(cfg)
    fl.set_footprint(synth.FOOTPRINT)
    fl.upload(N.GRID_MASTER, np.stack([i["master"] for i in insts]))
    planners = [orc.DwaPlanner(i["master"], synth.RES, 0.0, 0.0, ocfg) for i in insts]
    fl.set_plan()
    for p in planners:
        p.set_plan()
    plans = [insts[0]["plan"], insts[1]["plan"][:25], insts[2]["plan"][:120]]  # goal 8 m, 1 m, 4.8 m away
    pos = np.stack([i["pos"] for i in insts]).copy()
    vel = np.stack([i["vel"] for i in insts]).copy()
    with pytest.raises(nav.NavgpuError):
        fl.stage_poses(pos, vel)  # no plan staged yet
    rs = np.random.RandomState(3)
    for cyc in range(11):
        if cyc in (0, 6):
            fl.stage_planner(pos, vel, plans)       # the plan "arrives"
        else:
            pos = (pos + rs.normal(size=pos.shape) * [0.03, 0.03, 0.2]).astype(np.float32)
            vel = (vel + rs.normal(size=vel.shape) * [0.05, 0.0, 0.1]).astype(np.float32)
            fl.stage_poses(pos, vel)
        fl.planner_cycle()
        res = fl.results()
        for k in range(nI):
            o, otraj, _, cfull, ost = planners[k].cycle(pos[k], vel[k], plans[k], synth.FOOTPRINT)
            cost, status, vels = fl.samples(k)
            assert np.array_equal(status, ost) and np.array_equal(vels.view(np.uint32), planners[k].samples().view(np.uint32))
            sc = status == 1
            assert np.array_equal(cost[sc] < 0, cfull[sc] < 0)
            assert np.array_equal(cost[sc][cost[sc] < 0], cfull[sc][cfull[sc] < 0])
            assert np.allclose(cost[sc][cost[sc] >= 0], cfull[sc][cfull[sc] >= 0], rtol=0, atol=1e-5)
            r = res[k]
            assert (r.best_index, r.n_valid, r.n_scored) == (o.best_index, o.n_valid, o.n_scored), (cyc, k)
            assert abs(r.cost - o.cost) <= 1e-5 and r.oscillation_flags == o.oscillation_flags
            for gid, which in ((N.GRID_PATH, 0), (N.GRID_GOAL, 1), (N.GRID_GOAL_FRONT, 2)):
                assert np.array_equal(fl.download(gid, k, 1)[0].astype(np.float64), planners[k].grid(which)), (cyc, k, which)
    b = fl.wavefront_boxes()
    assert b.shape == (nI, 4) and (b[:, 0] <= b[:, 1]).all() and (b[:, 2] <= b[:, 3]).all()
    fl.close()


# ----------------------------------------------------------------------------------------------
# The call sequence of the costmap_2d::Layer adapters (navigation_amd/plugin/navgpu_layers.cpp, LayerBridge) on a
# ROLLING local costmap: stage -> navgpu_obstacle_update_bounds (shifts the resident layer grid, clears, marks) ->
# upload the host's master -> navgpu_obstacle_update_costs -> navgpu_inflate -> download.  The host side of
# LayeredCostmap::updateMap (updateOrigin of the master, resetMap of the box, layered_costmap.cpp:86-137) is done in numpy.
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("voxel", [False, True])
def test_layer_adapter_sequence_rolling(nav, orc, voxel):
    from navigation_amd import synth
    N = L(nav)
    n = 120
    insc = synth.inscribed_radius(synth.FOOTPRINT)
    layers = (N.LAYER_VOXEL if voxel else N.LAYER_OBSTACLE) | N.LAYER_INFLATION
    fl = nav.Fleet(1, n, n, synth.RES, layers=layers, track_unknown=True, max_points=720, max_observations=1, rolling_window=True)
    fl.configure_obstacle()
    fl.set_footprint(synth.FOOTPRINT)
    fl.configure_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, insc)
    o = orc.LayeredCostmap(True)
    o.resize(n, n, synth.RES, 0, 0)
    o.set_rolling(True)
    o.set_footprint(synth.FOOTPRINT)
    o.add_voxel() if voxel else o.add_obstacle()
    o.add_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, exact=True)
    o.set_footprint(synth.FOOTPRINT)
    world = synth.make_instance(400, 77)
    host_master = np.full((n, n), NOINFO, np.uint8)  # LayeredCostmap(track_unknown): default NO_INFORMATION
    origin = np.zeros(2)
    for cyc in range(6):
        x, y, yaw = 3.0 + 0.41 * cyc + (7.0 if cyc == 4 else 0.0), 2.5 + 0.23 * cyc, 0.3 * cyc
        inst = dict(world)
        inst["pos"] = np.array([x, y, yaw], np.float32)
        pts = synth.laser_scan(inst, cyc, max_range=4.0, z=0.3, z_jitter=1.0 if voxel else None)
        org = (float(x), float(y), 0.3)
        o.clear_observations()
        o.add_observation(pts, origin=org, obstacle_range=2.5, raytrace_range=3.0)
        o.update_map(float(x), float(y), float(yaw))
        # --- the adapter's updateBounds
        fl.stage_observations([[float(x), float(y), float(yaw)]],
                              [dict(instance=0, points=pts, origin=org, obstacle_range=2.5, raytrace_range=3.0)])
        fl.obstacle_update_bounds([[1e30, 1e30, -1e30, -1e30]])
        new_origin = fl.origins()[0]
        assert np.array_equal(new_origin, o.origin()), ("origin", cyc)
        # --- host side of LayeredCostmap::updateMap: master.updateOrigin + resetMap(box)
        cox, coy = int(round((new_origin[0] - origin[0]) / synth.RES)), int(round((new_origin[1] - origin[1]) / synth.RES))
        shifted = np.full((n, n), NOINFO, np.uint8)
        ys, xs = np.mgrid[0:n, 0:n]
        sy, sx = ys + coy, xs + cox
        ok = (sy >= 0) & (sy < n) & (sx >= 0) & (sx < n)
        shifted[ys[ok], xs[ok]] = host_master[sy[ok], sx[ok]]
        host_master, origin = shifted, new_origin.copy()
        x0, xn, y0, yn = [int(v) for v in o.bounds()]
        host_master[y0:yn, x0:xn] = NOINFO
        # --- the adapter's updateCosts
        fl.upload(N.GRID_MASTER, host_master)
        fl.obstacle_update_costs([[x0, y0, xn, yn]])
        fl.inflate(boxes=[[x0, y0, xn, yn]])
        host_master = fl.master()[0]
        assert np.array_equal(fl.download(N.GRID_OBSTACLE)[0], o.layer(2)), ("layer grid", cyc)
        if voxel:
            assert np.array_equal(fl.download(N.GRID_VOXEL)[0], o.voxels()), ("voxel columns", cyc)
        assert np.array_equal(host_master, o.master()), ("master", cyc)
    fl.close()


# ----------------------------------------------------------------------------------------------
# priority_queue_order = 1: InflationLayer::updateCosts as written (inflation_layer.cpp:226-293), byte for byte -
# including the cells where the reference's heap order makes it differ from the exact transform
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,density,unk", [(64, 0.01, 0.0), (64, 0.05, 0.1), (97, 0.02, 0.05), (400, 0.01, 0.0), (400, 0.001, 0.02),
                                            (250, 0.2, 0.0)])
def test_inflate_reference_priority_queue_order(nav, orc, n, density, unk):
    from test_gpu_parity import _random_map
    N = L(nav)
    rs = np.random.RandomState(n * 7 + int(density * 1000))  # the maps of test_inflate_full_window
    maps = np.stack([_random_map(rs, n, density, unk) for _ in range(3)])
    fl = nav.Fleet(3, n, n, 0.05, layers=N.LAYER_INFLATION)
    fl.configure_inflation(0.55, 10.0, 0.2, priority_queue_order=True)
    fl.upload(N.GRID_MASTER, maps)
    fl.inflate(boxes=[[0, 0, n, n]] * 3)
    got = fl.master()
    n_vs_exact = 0
    for k in range(3):
        ref = orc.inflate(maps[k], 0.05, 0.55, 10.0, 0.2, exact=False)
        assert np.array_equal(got[k], ref), f"map {k}: {int((got[k] != ref).sum())} cells differ from the reference's priority-queue walk"
        n_vs_exact += int((ref != orc.inflate(maps[k], 0.05, 0.55, 10.0, 0.2, exact=True)).sum())
    print(f"n={n} density={density}: byte-identical to the reference walk; {n_vs_exact} cells of it differ from the exact transform")
    if (n, density) == (400, 0.01):
        assert n_vs_exact > 0  # the case SURVEY 7 hard part 1 is about is really exercised
    # partial boxes and another radius / resolution on the same fleet geometry
    boxes = [[10, 20, min(60, n), min(90, n)], [0, 0, n, n], [n // 2, 0, n // 2 + 1, n]]
    fl.configure_inflation(1.0, 3.0, 0.35, priority_queue_order=True)
    fl.upload(N.GRID_MASTER, maps)
    fl.inflate(boxes=boxes)
    got = fl.master()
    for k in range(3):
        assert np.array_equal(got[k], orc.inflate(maps[k], 0.05, 1.0, 3.0, 0.35, box=boxes[k], exact=False)), ("box", k)
    # and back to the default mode on the same fleet
    fl.configure_inflation(0.55, 10.0, 0.2)
    fl.upload(N.GRID_MASTER, maps)
    fl.inflate(boxes=[[0, 0, n, n]] * 3)
    assert np.array_equal(fl.master()[0], orc.inflate(maps[0], 0.05, 0.55, 10.0, 0.2, exact=True))
    fl.close()


def test_layered_cycles_reference_priority_queue_order(nav, orc):
    """LayeredCostmap::updateMap cycles (static + obstacle + inflation) with the inflation layer in reference order:
    master grids byte-identical to the oracle running the reference's own walk."""
    from navigation_amd import synth
    N = L(nav)
    n, nI = 200, 2
    insts = [synth.make_instance(n, 140 + i, density=0.01) for i in range(nI)]
    insc = synth.inscribed_radius(synth.FOOTPRINT)
    fl = nav.Fleet(nI, n, n, synth.RES, layers=N.LAYER_STATIC | N.LAYER_OBSTACLE | N.LAYER_INFLATION, max_points=720, max_observations=1)
    fl.configure_obstacle()
    fl.set_footprint(synth.FOOTPRINT)
    fl.configure_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, insc, priority_queue_order=True)
    oracles = []
    for i, ins in enumerate(insts):
        occ = np.where(ins["cells"] == 254, 100, 0).astype(np.int8)
        fl.add_static_map(occ, first=i, count=1)
        o = orc.LayeredCostmap(False)
        o.set_footprint(synth.FOOTPRINT)
        o.add_static(occ, res=synth.RES)
        o.add_obstacle()
        o.add_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, exact=False)
        o.set_footprint(synth.FOOTPRINT)
        oracles.append(o)
    for cyc in range(3):
        obs, poses = [], []
        for i, ins in enumerate(insts):
            pts = synth.laser_scan(ins, cyc)
            org = (float(ins["pos"][0]), float(ins["pos"][1]), 0.3)
            obs.append(dict(instance=i, points=pts, origin=org, obstacle_range=2.5, raytrace_range=3.0))
            poses.append([float(v) for v in ins["pos"]])
            oracles[i].clear_observations()
            oracles[i].add_observation(pts, origin=org, obstacle_range=2.5, raytrace_range=3.0)
            oracles[i].update_map(*poses[-1])
        fl.stage_observations(poses, obs)
        fl.update_map()
        m = fl.master()
        for i in range(nI):
            assert np.array_equal(m[i], oracles[i].master()), (cyc, i)
    fl.close()


# ----------------------------------------------------------------------------------------------
# k_bfs_rows2: the register-resident row sweep, two rows per lane, for maps beyond k_bfs_rows' reach (up to 1024 x 1344;
# configs[4]'s 1000 x 1000).  Sizes pick its corner cases: 32 words per row with a ragged last word and with a full one,
# fewer words (a scalar row load instead of the 16-byte one), an odd row count (a lane whose second row does not exist),
# a partial last wave, 12 waves; (1000, 1400) is beyond it (k_bfs_global).
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nx,ny", [(1000, 1000), (1023, 1024), (992, 800), (700, 650), (993, 641), (1024, 1121), (801, 1344), (1000, 1400), (840, 1009)])
def test_big_map_wavefronts_match_oracle(nav, orc, nx, ny):
    from navigation_amd import synth
    from test_gpu_parity import _mapgrid_case
    res = synth.RES
    rs = np.random.RandomState(nx + ny)
    m = np.zeros((ny, nx), np.uint8)
    for _ in range(nx * ny // 2500):
        cx, cy, r = rs.randint(0, nx), rs.randint(0, ny), rs.randint(1, 6)
        m[max(0, cy - r):cy + r + 1, max(0, cx - r):cx + r + 1] = LETHAL
    # walls with gaps, also across the seams between the 32-cell words and at the last column / row
    for x in (31, 32, 511, 512, nx - 1):
        m[5:ny - 5, x] = LETHAL
        m[rs.randint(5, ny - 5, 6), x] = 0
    m[ny - 1, 3:nx // 2] = LETHAL
    m[rs.random_sample(m.shape) < 0.005] = NOINFO
    sx, sy = nx * res, ny * res
    plan = np.stack([np.linspace(0.1 * sx, 0.9 * sx, 60), np.linspace(0.2 * sy, 0.85 * sy, 60)], 1)
    for px, py in plan:
        m[int(py / res), int(px / res)] = 0
    levels = _mapgrid_case(nav, orc, m, plan, [0.3 * sx, 0.4 * sy, 0.3])
    assert levels > 300


def test_big_map_long_search(nav, orc):
    """A serpentine over a 1000 x 600 map: > 60 000 levels through every strip and both halves of every wave."""
    from test_gpu_parity import _mapgrid_case
    nx, ny = 1000, 600
    m = np.zeros((ny, nx), np.uint8)
    for k, row in enumerate(range(8, ny - 2, 8)):
        m[row, :] = LETHAL
        if k % 2:
            m[row, 1:3] = 0
        else:
            m[row, nx - 3:nx - 1] = 0
    plan = np.stack([np.linspace(0.1, 0.6, 12), np.full(12, 0.12)], 1)
    levels = _mapgrid_case(nav, orc, m, plan, [0.3, 0.12, 0.0])
    assert levels > 60000, levels


def test_big_map_bounded_equals_complete(nav, orc):
    """1000 x 1000, three robots: bounded searches (stores inside the robot's region only) give every sample the cost
    and status the whole-grid searches give, the completed grids are equal, robot 0 equals the oracle."""
    from navigation_amd import synth
    from test_gpu_parity import _inflated_instance
    N = L(nav)
    n, n_inst = 1000, 3
    cfg = nav.DwaConfig(vx_samples=8, vy_samples=5, vth_samples=9, sim_time=1.7, sim_granularity=0.085, discretize_by_time=1)
    insts = [_inflated_instance(orc, n, 60 + i, synth) for i in range(n_inst)]
    masters = np.stack([i["master"] for i in insts])
    pos = np.stack([i["pos"] for i in insts])
    vel = np.stack([i["vel"] for i in insts])
    plans = np.stack([i["plan"] for i in insts])
    out = {}
    for bounded in (1, 0):
        fl = nav.Fleet(n_inst, n, n, synth.RES, layers=N.LAYER_OBSTACLE, keep_sample_costs=True, max_sim_steps=32, max_plan=256)
        fl.configure_planner(cfg)
        fl.set_footprint(synth.FOOTPRINT5)
        fl.set_bounded_map_grids(bounded)
        fl.upload(N.GRID_MASTER, masters)
        fl.set_plan()
        for _ in range(2):  # the second cycle runs in last cycle's longest-first order, on reused seed scratch
            res = fl.find_best_path(pos, vel, plans)
        lv = fl.wavefront_levels()
        samples = [fl.samples(k) for k in range(n_inst)]
        grids = [fl.download(g) for g in (N.GRID_PATH, N.GRID_GOAL, N.GRID_GOAL_FRONT)]
        out[bounded] = (res, lv, samples, grids)
        fl.close()
    (rb, lb, sb, gb), (rc, lc, sc, gc) = out[1], out[0]
    assert (lb < lc).sum() >= 2 * n_inst, (lb, lc)
    for k in range(n_inst):
        assert (rb[k].best_index, rb[k].n_valid, rb[k].n_scored, rb[k].cost) == (rc[k].best_index, rc[k].n_valid, rc[k].n_scored, rc[k].cost)
        assert np.array_equal(sb[k][0], sc[k][0]) and np.array_equal(sb[k][1], sc[k][1])
    for a, b in zip(gb, gc):
        assert np.array_equal(a, b)
    p = orc.DwaPlanner(masters[0], synth.RES, 0.0, 0.0, orc.DwaConfig(**cfg.as_dict()))
    p.set_plan()
    o, _, _, cfull, ost = p.cycle(pos[0], vel[0], plans[0], synth.FOOTPRINT5)
    scored = ost == 1
    assert np.array_equal(sb[0][1], ost)
    assert np.allclose(sb[0][0][scored], cfull[scored], rtol=0, atol=1e-5)
    for which in range(3):
        assert np.array_equal(gb[which][0].astype(np.float64).reshape(-1), p.grid(which).reshape(-1))


# ----------------------------------------------------------------------------------------------
# StaticLayer under a rolling window (static_layer.cpp:262-333): a static map with its own geometry is looked up per
# master cell through a map_frame <- global_frame transform; the layer's extent joins the bounds every cycle.
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("use_maximum,track_unknown", [(False, True), (True, True), (False, False), (True, False)])
def test_rolling_static_layer_cycles(nav, orc, use_maximum, track_unknown):
    from navigation_amd import synth
    N = L(nav)
    n, nI = 120, 3
    insc = synth.inscribed_radius(synth.FOOTPRINT)
    fl = nav.Fleet(nI, n, n, synth.RES, layers=N.LAYER_STATIC | N.LAYER_OBSTACLE | N.LAYER_INFLATION, track_unknown=track_unknown,
                   max_points=720, max_observations=1, rolling_window=True)
    fl.configure_obstacle()
    fl.set_footprint(synth.FOOTPRINT)
    fl.configure_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, insc)
    # the static map: 260 x 200 cells of 0.1 m (coarser than the 0.05 m costmap), origin (-3, -2), walls, free space, unknown
    rs = np.random.RandomState(11)
    occ = np.zeros((200, 260), np.int8)
    occ[rs.random_sample(occ.shape) < 0.01] = 100
    occ[40:44, 20:200] = 100
    occ[:, 120:122] = 100
    occ[rs.random_sample(occ.shape) < 0.05] = -1
    occ[rs.random_sample(occ.shape) < 0.02] = 60  # below the lethal threshold: free in a trinary map
    kw = dict(track_unknown_space=track_unknown, use_maximum=use_maximum)
    fl.set_rolling_static_map(occ, 0.1, -3.0, -2.0, **kw)
    world = synth.make_instance(400, 78)
    oracles = []
    for i in range(nI):
        o = orc.LayeredCostmap(track_unknown)
        o.resize(n, n, synth.RES, 0, 0)
        o.set_rolling(True)
        o.set_footprint(synth.FOOTPRINT)
        o.add_static_rolling(occ, 0.1, -3.0, -2.0, **kw)
        o.add_obstacle()
        o.add_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, exact=True)
        o.set_footprint(synth.FOOTPRINT)
        oracles.append(o)
    starts = rs.uniform(2.0, 4.0, (nI, 2))
    for cyc in range(5):
        poses, obs = [], []
        for i in range(nI):
            # odom drifts against the map: a rotation + translation that changes every cycle (robot 0 keeps the identity)
            a = 0.0 if i == 0 else 0.07 * cyc * i - 0.2
            basis = [np.cos(a), -np.sin(a), 0, np.sin(a), np.cos(a), 0, 0, 0, 1]
            origin = [0.0, 0.0, 0.0] if i == 0 else [0.31 * i - 0.05 * cyc, -0.4 + 0.03 * cyc, 0.0]
            fl.set_static_transform(basis, origin, first=i, count=1)
            oracles[i].set_static_transform(basis, origin)
            x = starts[i, 0] + 0.41 * cyc * (1 if i != 1 else -1) + (14.0 if (cyc == 3 and i == 2) else 0.0)  # robot 2 leaves the static map
            y = starts[i, 1] + 0.23 * cyc
            yaw = 0.3 * cyc - 0.5 * i
            inst = dict(world)
            inst["pos"] = np.array([x, y, yaw], np.float32)
            pts = synth.laser_scan(inst, cyc, max_range=4.0)
            org = (float(x), float(y), 0.3)
            poses.append([float(x), float(y), float(yaw)])
            obs.append(dict(instance=i, points=pts, origin=org, obstacle_range=2.5, raytrace_range=3.0))
            oracles[i].clear_observations()
            oracles[i].add_observation(pts, origin=org, obstacle_range=2.5, raytrace_range=3.0)
            oracles[i].update_map(*poses[-1])
        fl.stage_observations(poses, obs)
        fl.update_map()
        m, b, org_g = fl.master(), fl.bounds(), fl.origins()
        for i in range(nI):
            assert np.array_equal(org_g[i], oracles[i].origin()), ("origin", cyc, i)
            assert np.array_equal(b[i], oracles[i].bounds()), ("box", cyc, i)
            assert np.array_equal(m[i], oracles[i].master()), ("master", cyc, i)
        # (use_maximum over a NO_INFORMATION default: the rolling branch's plain std::max keeps 255 everywhere, static_layer.cpp:328)
        assert use_maximum or ((m[0] == LETHAL).sum() > 50 and (not track_unknown or (m[0] == NOINFO).sum() > 50))
    fl.close()


# ----------------------------------------------------------------------------------------------
# Differential fuzz: random planner configurations (limits, accelerations, sample counts, granularities, scales incl.
# zero, use_dwa / discretize_by_time / sum_scores / allow_unknown, footprints, map sizes incl. non-multiples of 32,
# plan lengths) through _check_planner: grids, sample velocities, accept masks, failure codes, costs, winner, trajectory.
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", list(range(48)))
def test_planner_random_configurations_vs_oracle(nav, orc, seed):
    from navigation_amd import synth
    rs = np.random.RandomState(7000 + seed)
    pick = lambda *v: v[rs.randint(len(v))]
    max_vel_x = float(rs.uniform(0.2, 0.9))
    cfg = dict(
        max_vel_x=max_vel_x, min_vel_x=float(pick(0.0, -0.1, 0.05)), max_vel_y=float(pick(0.0, 0.1, 0.25)), min_vel_y=float(pick(0.0, -0.1, -0.25)),
        max_trans_vel=float(max_vel_x * pick(1.0, 0.8, 1.5)), min_trans_vel=float(pick(0.0, 0.1, -1.0)),
        max_rot_vel=float(rs.uniform(0.5, 1.5)), min_rot_vel=float(pick(0.0, 0.3, 0.4)),
        acc_lim_x=float(rs.uniform(0.5, 3.0)), acc_lim_y=float(rs.uniform(0.5, 3.0)), acc_lim_theta=float(rs.uniform(1.0, 4.0)),
        sim_time=float(pick(0.8, 1.3, 1.7, 2.0)), sim_granularity=float(pick(0.025, 0.05, 0.1)), angular_sim_granularity=float(pick(0.05, 0.1, 0.2)),
        sim_period=float(pick(0.05, 0.1, 0.2)), path_distance_bias=float(pick(32.0, 0.0, 5.0)), goal_distance_bias=float(pick(24.0, 0.0, 40.0)),
        occdist_scale=float(pick(0.01, 0.0, 0.2)), forward_point_distance=float(pick(0.325, 0.0, 0.6, -0.2)),
        vx_samples=int(pick(1, 3, 6, 9)), vy_samples=int(pick(1, 4, 7)), vth_samples=int(pick(1, 5, 12, 20)),
        use_dwa=int(pick(1, 1, 0)), discretize_by_time=int(pick(0, 1)), sum_scores=int(pick(0, 0, 1)))
    n = int(pick(96, 120, 157, 200, 233))
    fp = pick(synth.FOOTPRINT, synth.FOOTPRINT5, np.array([[0.15, 0.0], [-0.1, 0.12], [-0.1, -0.12]]))
    _check_planner(nav, orc, n, cfg, n_inst=2, footprint=fp, allow_unknown=int(pick(0, 1)), unknown_frac=float(pick(0.0, 0.01)), cycles=2,
                   seed0=500 + 3 * seed, plan_len=int(pick(200, 60, 25)), near_obstacles=int(pick(0, 3, 8)))


# ----------------------------------------------------------------------------------------------
# Differential fuzz of the layered costmap: random layer parameters (2-D / voxel obstacle layer, combination method,
# footprint clearing, unknown tracking, ranges, height limits, voxel geometry and thresholds, static layer, rolling
# window, inflation geometry, map sizes), 4 cycles of moving robots with one or two sensors (marking-only and
# clearing-only observations included): master, layer grid, voxel columns, box and origin against the oracle.
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", list(range(40)))
def test_layered_costmap_random_configurations_vs_oracle(nav, orc, seed):
    from navigation_amd import synth
    N = L(nav)
    rs = np.random.RandomState(9100 + seed)
    pick = lambda *v: v[rs.randint(len(v))]
    n = int(pick(96, 130, 160, 211))
    nI = 2
    voxel, rolling, track_unknown = bool(pick(0, 1)), bool(pick(0, 1)), bool(pick(0, 1))
    with_static = bool(pick(0, 1)) and not rolling
    fp = pick(synth.FOOTPRINT, synth.FOOTPRINT5)
    insc = synth.inscribed_radius(fp)
    comb, fpc = int(pick(1, 1, 0)), bool(pick(1, 1, 0))
    max_h = float(pick(2.0, 1.0, 0.6))
    vox = dict(z_voxels=int(pick(10, 16, 5)), origin_z=float(pick(0.0, -0.1)), z_resolution=float(pick(0.2, 0.1, 0.25)),
               unknown_threshold=int(pick(15, 4, 0)), mark_threshold=int(pick(0, 0, 1)))
    infl_r, infl_k = float(pick(0.55, 0.3, 0.7)), float(pick(10.0, 3.0))
    obs_range, ray_range = float(pick(2.5, 1.5, 4.0)), float(pick(3.0, 2.0, 5.0))
    layers = (N.LAYER_VOXEL if voxel else N.LAYER_OBSTACLE) | N.LAYER_INFLATION | (N.LAYER_STATIC if with_static else 0)
    fl = nav.Fleet(nI, n, n, synth.RES, layers=layers, track_unknown=track_unknown, max_points=1440, max_observations=2, rolling_window=rolling)
    fl.configure_obstacle(combination_method=comb, footprint_clearing_enabled=fpc, max_obstacle_height=max_h, **vox)
    fl.set_footprint(fp)
    fl.configure_inflation(infl_r, infl_k, insc)
    world = synth.make_instance(400, 900 + seed)
    occ = np.where(world["cells"][:n, :n] == 254, 100, 0).astype(np.int8)
    occ[rs.random_sample(occ.shape) < 0.02] = -1
    oracles = []
    for i in range(nI):
        o = orc.LayeredCostmap(track_unknown)
        o.resize(n, n, synth.RES, 0, 0)
        o.set_rolling(rolling)
        o.set_footprint(fp)
        if with_static:
            fl.add_static_map(occ, first=i, count=1, track_unknown_space=track_unknown)
            o.add_static(occ, res=synth.RES, track_unknown_space=track_unknown)
        if voxel:
            o.add_voxel(combination_method=comb, footprint_clearing=fpc, max_obstacle_height=max_h, **vox)
        else:
            o.add_obstacle(combination_method=comb, footprint_clearing=fpc, max_obstacle_height=max_h)
        o.add_inflation(infl_r, infl_k, exact=True)
        o.set_footprint(fp)
        oracles.append(o)
    half = n * synth.RES / 2
    starts = rs.uniform(half - 0.5, half + 0.5, (nI, 2))
    two_sensors = bool(pick(0, 1))
    for cyc in range(4):
        poses, obs = [], []
        for i in range(nI):
            x = starts[i, 0] + (0.33 * cyc * (1 if i == 0 else -0.6) if rolling else 0.05 * cyc)
            y = starts[i, 1] + (0.19 * cyc if rolling else -0.04 * cyc)
            yaw = 0.4 * cyc - 0.7 * i
            inst = dict(world)
            inst["pos"] = np.array([x, y, yaw], np.float32)
            pts = synth.laser_scan(inst, cyc, max_range=4.0, z=0.3, z_jitter=1.2 if voxel else None)
            if cyc == 1:
                pts[::9, 2] = max_h + 0.3  # above max_obstacle_height
            org = (float(x), float(y), 0.3 + 0.1 * cyc)
            poses.append([float(x), float(y), float(yaw)])
            kinds = [dict(marking=True, clearing=True)] if not two_sensors else [dict(marking=True, clearing=False), dict(marking=False, clearing=True)]
            oracles[i].clear_observations()
            for kk, kind in enumerate(kinds):
                p_k = pts if kk == 0 else pts[::2].copy()
                obs.append(dict(instance=i, points=p_k, origin=org, obstacle_range=obs_range, raytrace_range=ray_range, **kind))
                oracles[i].add_observation(p_k, origin=org, obstacle_range=obs_range, raytrace_range=ray_range, **kind)
            oracles[i].update_map(*poses[-1])
        fl.stage_observations(poses, obs)
        fl.update_map()
        m, ol, b, org_g = fl.master(), fl.download(N.GRID_OBSTACLE), fl.bounds(), fl.origins()
        vx = fl.download(N.GRID_VOXEL) if voxel else None
        for i in range(nI):
            tag = (seed, cyc, i, dict(voxel=voxel, rolling=rolling, static=with_static, unknown=track_unknown, comb=comb))
            assert np.array_equal(org_g[i], oracles[i].origin()), ("origin",) + tag
            assert np.array_equal(b[i], oracles[i].bounds()), ("box",) + tag
            assert np.array_equal(ol[i], oracles[i].layer(2)), ("layer",) + tag
            if voxel:
                assert np.array_equal(vx[i], oracles[i].voxels()), ("voxels",) + tag
            assert np.array_equal(m[i], oracles[i].master()), ("master",) + tag
    fl.close()


# ----------------------------------------------------------------------------------------------
# Differential fuzz of the legacy TrajectoryPlanner (f-3): random BaseLocalPlanner configurations, 6 closed-loop cycles
# of two robots each, against the oracle: every createTrajectories call's sample, cost, length; the result, the
# trajectory and the persistent state (escape / oscillation flags).
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", list(range(40)))
def test_trajectory_planner_random_configurations_vs_oracle(nav, orc, seed):
    from navigation_amd import synth
    from test_gpu_parity import _inflated_instance, _tp_compare_cycle
    N = L(nav)
    rs = np.random.RandomState(8200 + seed)
    pick = lambda *v: v[rs.randint(len(v))]
    n, n_inst = int(pick(120, 160, 201)), 2
    size = n * synth.RES
    cfg = N.TpConfig(
        acc_lim_x=float(rs.uniform(1.0, 3.0)), acc_lim_y=float(rs.uniform(1.0, 3.0)), acc_lim_theta=float(rs.uniform(1.5, 4.0)),
        sim_time=float(pick(0.8, 1.2, 1.7)), sim_granularity=float(pick(0.025, 0.05, 0.1)), angular_sim_granularity=float(pick(0.025, 0.05, 0.1)),
        pdist_scale=float(pick(0.6, 0.0, 2.0)), gdist_scale=float(pick(0.8, 0.0, 1.5)), occdist_scale=float(pick(0.01, 0.0, 0.1)),
        heading_lookahead=float(pick(0.325, 0.6)), max_vel_x=float(rs.uniform(0.3, 0.8)), min_vel_x=float(pick(0.0, 0.1)),
        max_vel_th=float(rs.uniform(0.6, 1.4)), min_vel_th=float(-rs.uniform(0.6, 1.4)), min_in_place_vel_th=float(pick(0.4, 0.2)),
        backup_vel=float(pick(-0.1, -0.2)), sim_period=float(pick(0.05, 0.1)),
        y_vels=pick((-0.3, -0.1, 0.1, 0.3), (-0.2, 0.2), (0.15,)),
        vx_samples=int(pick(3, 6, 8)), vtheta_samples=int(pick(5, 9, 12)), holonomic_robot=int(pick(0, 1)), dwa=int(pick(0, 1)),
        allow_unknown=int(pick(0, 1)), heading_scoring=int(pick(0, 0, 1)), simple_attractor=int(pick(0, 0, 0, 1)),
        heading_scoring_timestep=float(pick(0.1, 0.4, 0.8)))
    fp = pick(synth.FOOTPRINT, synth.FOOTPRINT5)
    insts = [_inflated_instance(orc, n, 300 + 2 * seed + i, synth) for i in range(n_inst)]
    c = n // 2
    for ins in insts:  # a few lethal cells near the robot: colliding samples
        for _ in range(int(pick(0, 4, 9))):
            a, d = rs.uniform(0, 2 * np.pi), rs.uniform(0.4, 1.0) / synth.RES
            ins["cells"][int(c + d * np.sin(a)), int(c + d * np.cos(a))] = LETHAL
        ins["master"] = orc.inflate(ins["cells"], synth.RES, synth.INFLATION_RADIUS, synth.COST_SCALING, synth.inscribed_radius(fp), exact=True)
    fl = nav.Fleet(n_inst, n, n, synth.RES, layers=N.LAYER_OBSTACLE, max_sim_steps=96, max_plan=256)
    fl.set_footprint(fp)
    fl.upload(N.GRID_MASTER, np.stack([i["master"] for i in insts]))
    fl.configure_trajectory_planner(cfg)
    oracles = [orc.TrajectoryPlanner(i["master"], synth.RES, cfg, fp) for i in insts]
    plan_len = int(pick(200, 80, 30))
    for k, ins in enumerate(insts):
        fl.tp_update_plan(k, ins["plan"][:plan_len])
        oracles[k].update_plan(ins["plan"][:plan_len])
    pos = np.array([[size / 2, size / 2, rs.uniform(-3.0, 3.0)] for _ in range(n_inst)], np.float32)
    vel = np.zeros((n_inst, 3), np.float32)
    vel[0] = (0.2, 0.0, 0.1)
    for cyc in range(6):
        got = _tp_compare_cycle(fl, N, oracles, pos, vel, cyc)
        dt = 0.1
        for k in range(n_inst):
            r = got[k]
            th = float(pos[k, 2])
            pos[k, 0] += (r.drive[0] * np.cos(th) - r.drive[1] * np.sin(th)) * dt
            pos[k, 1] += (r.drive[0] * np.sin(th) + r.drive[1] * np.cos(th)) * dt
            pos[k, 2] += r.drive[2] * dt
            vel[k] = r.drive
    fl.close()


# ----------------------------------------------------------------------------------------------
# MapGridCostFunction's aggregation types and yshift (map_grid_cost_function.cpp:42-53, 75-129): what DWAPlanner never
# sets itself, configured per critic through navgpu_planner_set_map_grid_options; every sample's cost, failure code,
# the winner and checkTrajectory against the oracle.
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("options", [
    [("path", "sum", 0.0)],
    [("goal", "product", 0.0), ("path", "sum", 0.0)],
    [("goal_front", "sum", 0.15), ("alignment", "last", -0.2)],
    [("path", "last", 0.1), ("goal", "last", -0.1)],
    [("path", "product", 0.05), ("goal", "sum", 0.0), ("goal_front", "product", 0.0), ("alignment", "sum", 0.3)],
])
def test_map_grid_aggregation_and_yshift(nav, orc, options):
    for kw in (dict(discretize_by_time=1, sim_granularity=0.1, sim_time=1.5), dict()):
        _check_planner(nav, orc, 160, dict(vx_samples=5, vy_samples=4, vth_samples=9, **kw), n_inst=2, cycles=2, seed0=40, near_obstacles=4,
                       map_grid_options=options)
    # setting everything back to Last / 0 returns to the usual kernels (and to bounded wavefronts)
    from navigation_amd import synth
    N = L(nav)
    fl = nav.Fleet(1, 120, 120, synth.RES, layers=N.LAYER_OBSTACLE, keep_sample_costs=True)
    fl.configure_planner(nav.DwaConfig(vx_samples=4, vy_samples=1, vth_samples=5))
    fl.set_footprint(synth.FOOTPRINT)
    ins = synth.make_instance(120, 77)
    fl.upload(N.GRID_MASTER, ins["cells"][None])
    fl.set_plan()
    r0 = fl.find_best_path([ins["pos"]], [ins["vel"]], [ins["plan"]])[0]
    c0 = fl.samples(0)[0].copy()
    fl.set_map_grid_options("path", "sum", 0.1)
    fl.find_best_path([ins["pos"]], [ins["vel"]], [ins["plan"]])
    assert not np.array_equal(fl.samples(0)[0], c0)
    fl.set_map_grid_options("path", "last", 0.0)
    r1 = fl.find_best_path([ins["pos"]], [ins["vel"]], [ins["plan"]])[0]
    assert (r1.best_index, r1.cost) == (r0.best_index, r0.cost) and np.array_equal(fl.samples(0)[0], c0)
    # DWAPlanner::checkTrajectory with the options set: the general step in the explicit-sample kernel
    from test_gpu_parity import _inflated_instance
    ins2 = _inflated_instance(orc, 160, 4, synth)
    m2 = ins2["master"]
    m2[:, int(ins2["pos"][0] / synth.RES) + 12] = LETHAL
    cfg2 = nav.DwaConfig(vx_samples=6, vy_samples=6, vth_samples=8, sim_time=1.0, sim_granularity=0.1, discretize_by_time=1)
    f2 = nav.Fleet(1, 160, 160, synth.RES, layers=N.LAYER_OBSTACLE)
    f2.configure_planner(cfg2)
    f2.set_footprint(synth.FOOTPRINT)
    f2.upload(N.GRID_MASTER, m2)
    p2 = orc.DwaPlanner(m2, synth.RES, 0.0, 0.0, orc.DwaConfig(**cfg2.as_dict()))
    for critic, agg, ysh in options:
        f2.set_map_grid_options(critic, agg, ysh)
        p2.set_map_grid_options(critic, agg, ysh)
    pos2 = ins2["pos"].copy()
    pos2[2] = 0.0
    f2.find_best_path([pos2], [ins2["vel"]], [ins2["plan"]])
    p2.cycle(pos2, ins2["vel"], ins2["plan"], synth.FOOTPRINT)
    for vs in ([0.5, 0.0, 0.0], [0.1, 0.0, 0.5], [0.0, 0.0, 0.0], [0.3, 0.1, -0.4], [0.55, 0.0, 0.0]):
        assert f2.check_trajectory(0, vs) == p2.check_trajectory(pos2, ins2["vel"], vs), vs
    f2.close()
    from navigation_amd._lib import NavgpuError, check
    for bad in ((4, 0, 0.0), (0, 3, 0.0), (1, 0, float("nan"))):
        with pytest.raises(NavgpuError):
            check(fl.L.navgpu_planner_set_map_grid_options(fl.h, *bad), "set_map_grid_options")
    fl.close()


# ----------------------------------------------------------------------------------------------
# Determinism of the whole step at configs[2]'s size: the same 24 cycles (perturbed poses, scans, plans) run twice on two
# fleets give the same winners, costs, counters, wavefront levels and master grids, cycle by cycle.  (Work queues,
# atomics and the hand-shakes between waves must not leak scheduling order into results.)
# ----------------------------------------------------------------------------------------------
def test_fleet_256_cycles_are_deterministic(nav):
    import bench
    runs = []
    for _ in range(2):
        fl, insts, cfg = bench.build_fleet(nav, 256, 400, 0)
        _, _, pos_h, vel_h, plans_h = fl._bench_host_inputs
        poses = bench.PoseSchedule(pos_h, vel_h, 24, seed=5)
        out = []
        for k in range(24):
            bench.step(fl, poses, k)
            res = fl.results()
            out.append((np.array([(r.best_index, r.n_scored, r.n_valid, r.oscillation_flags) for r in res]), np.array([r.cost for r in res]),
                        fl.wavefront_levels().copy()))
        runs.append((out, fl.master().copy()))
        fl.close()
    for k in range(24):
        a, b = runs[0][0][k], runs[1][0][k]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), k
    assert np.array_equal(runs[0][1], runs[1][1])
    assert (runs[0][0][-1][0][:, 0] >= 0).sum() > 200


# ----------------------------------------------------------------------------------------------
# CostmapLayer::resetBoundingBox (costmap_layer.cpp:30-60; Costmap2DROS::resetBoundingBox's per-layer call): the obstacle
# / voxel layer's 2-D grid is reset inside a world box and the box joins the bounds of the next update (extra bounds) -
# so the master is rewritten there.  Marks accumulated over two cycles, a reset between cycles 2 and 3 (robot 0 a box in
# the scanned area, robot 1 a box partly off the map), then two more cycles; non-rolling and rolling, 2-D and voxel.
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("voxel,rolling", [(False, False), (True, False), (False, True), (True, True)])
def test_layer_reset_bounding_box(nav, orc, voxel, rolling):
    from navigation_amd import synth
    N = L(nav)
    n, nI = 160, 2
    insc = synth.inscribed_radius(synth.FOOTPRINT)
    layers = (N.LAYER_VOXEL if voxel else N.LAYER_OBSTACLE) | N.LAYER_INFLATION
    fl = nav.Fleet(nI, n, n, synth.RES, layers=layers, track_unknown=True, max_points=720, max_observations=1, rolling_window=rolling)
    fl.configure_obstacle()
    fl.set_footprint(synth.FOOTPRINT)
    fl.configure_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, insc)
    oracles = []
    for i in range(nI):
        o = orc.LayeredCostmap(True)
        o.resize(n, n, synth.RES, 0, 0)
        o.set_rolling(rolling)
        o.set_footprint(synth.FOOTPRINT)
        o.add_voxel() if voxel else o.add_obstacle()
        o.add_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, exact=True)
        o.set_footprint(synth.FOOTPRINT)
        oracles.append(o)
    world = synth.make_instance(400, 91)
    for cyc in range(5):
        if cyc == 2:
            boxes = np.array([[3.0, 2.5, 5.2, 4.4], [-1.0, 6.0, 2.5, 9.5]])
            fl.reset_bounding_box(boxes)
            for i in range(nI):
                oracles[i].reset_bounding_box(*boxes[i])
            ol = fl.download(N.GRID_OBSTACLE)
            for i in range(nI):
                assert np.array_equal(ol[i], oracles[i].layer(2)), ("layer after reset", i)
        poses, obs = [], []
        for i in range(nI):
            x, y, yaw = 3.6 + 0.3 * i + (0.25 * cyc if rolling else 0.0), 3.9 - 0.2 * i, 0.5 * cyc - 0.3 * i
            inst = dict(world)
            inst["pos"] = np.array([x, y, yaw], np.float32)
            pts = synth.laser_scan(inst, cyc, max_range=4.0, z=0.3, z_jitter=1.0 if voxel else None)
            org = (float(x), float(y), 0.3)
            poses.append([float(x), float(y), float(yaw)])
            obs.append(dict(instance=i, points=pts, origin=org, obstacle_range=2.5, raytrace_range=3.0))
            oracles[i].clear_observations()
            oracles[i].add_observation(pts, origin=org, obstacle_range=2.5, raytrace_range=3.0)
            oracles[i].update_map(*poses[-1])
        fl.stage_observations(poses, obs)
        fl.update_map()
        m, ol, b = fl.master(), fl.download(N.GRID_OBSTACLE), fl.bounds()
        for i in range(nI):
            assert np.array_equal(b[i], oracles[i].bounds()), ("box", cyc, i, b[i], oracles[i].bounds())
            assert np.array_equal(ol[i], oracles[i].layer(2)), ("layer", cyc, i)
            assert np.array_equal(m[i], oracles[i].master()), ("master", cyc, i)
        if cyc == 2 and not rolling:  # the reset box is part of this update's window
            assert b[0][0] <= 60 and b[0][1] >= 104 and b[0][2] <= 50 and b[0][3] >= 88
    # Costmap2D::resetMap on cell coordinates (what the Layer adapter forwards)
    fl.reset_window(N.GRID_OBSTACLE, 10, 20, 70, 90)
    want = [o.layer(2).copy() for o in oracles]
    got = fl.download(N.GRID_OBSTACLE)
    for i in range(nI):
        want[i][20:90, 10:70] = 255
        assert np.array_equal(got[i], want[i])
    fl.close()
