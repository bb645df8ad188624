"""Parity tests proper: the HIP path (through the C-ABI, libnavgpu.so) against the CPU oracle on
the same seeded inputs, and against the reference's own test expectations.  Need a real MI355X.

Bar: bit-exact for bytes / cell indices / selected sample index / status codes; 1e-5 on trajectory
cost floats (north_star); inflation bit-exact vs the order-independent exact-EDT specification and
">= reference, differing fraction reported and <= 1e-4" vs the reference's priority-queue walk."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LETHAL, INSCRIBED, NOINFO, FREE = 254, 253, 255, 0
MAX_Z = 1.0


@pytest.fixture(scope="module")
def nav():
    import navigation_amd as nav
    nav.lib()  # raises if libnavgpu.so is missing: no fallback
    assert nav.lib().navgpu_device_count() > 0, "no HIP device visible"
    return nav


def L(nav):
    from navigation_amd import _lib
    return _lib


# ----------------------------------------------------------------------------------------------
# inflation
# ----------------------------------------------------------------------------------------------
def _random_map(rs, n, density, unknown_frac=0.0):
    g = np.zeros((n, n), np.uint8)
    g[rs.random_sample((n, n)) < density] = LETHAL
    if unknown_frac:
        g[(rs.random_sample((n, n)) < unknown_frac) & (g == 0)] = NOINFO
    # some pre-existing non-zero costs so the max() rule matters
    m = (rs.random_sample((n, n)) < 0.02) & (g == 0)
    g[m] = rs.randint(1, 253, m.sum())
    return g


@pytest.mark.parametrize("n,density,unk", [(64, 0.01, 0.0), (64, 0.05, 0.1), (97, 0.02, 0.05), (400, 0.01, 0.0),
                                            (400, 0.001, 0.02), (400, 0.2, 0.0)])
def test_inflate_full_window(nav, orc, n, density, unk):
    N = L(nav)
    rs = np.random.RandomState(n * 7 + int(density * 1000))
    maps = np.stack([_random_map(rs, n, density, unk) for _ in range(3)])
    fl = nav.Fleet(3, n, n, 0.05, layers=N.LAYER_INFLATION)
    insc = 0.2
    fl.configure_inflation(0.55, 10.0, insc)
    fl.upload(N.GRID_MASTER, maps)
    fl.inflate(boxes=[[0, 0, n, n]] * 3)
    got = fl.master()
    n_diff = 0
    for k in range(3):
        exact = orc.inflate(maps[k], 0.05, 0.55, 10.0, insc, exact=True)
        assert np.array_equal(got[k], exact), "GPU inflation != exact-EDT oracle"
        ref = orc.inflate(maps[k], 0.05, 0.55, 10.0, insc, exact=False)
        assert (got[k] >= ref).all() or unk > 0  # with unknowns max() vs the 255-rule can order differently
        n_diff += int((got[k] != ref).sum())
    frac = n_diff / (3.0 * n * n)
    print(f"inflation n={n} density={density}: {n_diff} cells differ from the reference PQ walk ({frac:.2e})")
    # SURVEY §7 hard part 1: <= 1e-4 at the BASELINE shape (400x400, 1 %); small dense maps sit higher
    assert frac <= (1e-4 if (n == 400 and density <= 0.01) else 1e-3)
    fl.close()


def test_inflate_partial_boxes_and_radii(nav, orc):
    N = L(nav)
    rs = np.random.RandomState(5)
    n = 150
    for radius, scaling, insc in [(0.55, 10.0, 0.2), (1.0, 3.0, 0.35), (0.05, 10.0, 0.0), (2.0, 1.0, 0.5)]:
        maps = np.stack([_random_map(rs, n, 0.01, 0.03) for _ in range(4)])
        boxes = [[0, 0, n, n], [10, 20, 60, 90], [100, 100, 150, 150], [40, 0, 41, 150]]
        fl = nav.Fleet(4, n, n, 0.05, layers=N.LAYER_INFLATION)
        fl.configure_inflation(radius, scaling, insc)
        fl.upload(N.GRID_MASTER, maps)
        fl.inflate(boxes=boxes)
        got = fl.master()
        for k in range(4):
            exact = orc.inflate(maps[k], 0.05, radius, scaling, insc, box=boxes[k], exact=True)
            assert np.array_equal(got[k], exact), (radius, k)
        fl.close()


def test_inflate_lut_matches_reference_tables(nav, orc):
    # the cost table the GPU uploads is computed by the same fp64 libm expressions as the oracle's
    R, costs, dists = orc.cost_lut(0.05, 0.55, 10.0, 0.2)
    assert R == 11 and costs[0, 0] == LETHAL and costs[1, 0] == INSCRIBED


# ----------------------------------------------------------------------------------------------
# layered costmap cycles: the reference's own scenarios through the GPU path
# ----------------------------------------------------------------------------------------------
def _radii(length, width):
    return [[width, length], [width, -length], [-width, -length], [-width, length]]


class GpuLayered:
    """Drives a 1-instance fleet the way the reference tests drive LayeredCostmap (testing_helper.h):
    static observations accumulate and are re-applied on every updateMap."""

    def __init__(self, nav, orc, n, res=1.0, static=None, inflation=None, polygon=None, track_unknown=False):
        N = L(nav)
        layers = N.LAYER_OBSTACLE | (N.LAYER_STATIC if static is not None else 0) | (N.LAYER_INFLATION if inflation else 0)
        self.fl = nav.Fleet(1, n, n, res, layers=layers, track_unknown=track_unknown, max_points=64, max_observations=16)
        self.N = N
        self.obs = []
        self.o = orc.LayeredCostmap(track_unknown)
        if static is None:
            self.o.resize(n, n, res, 0, 0)
        if polygon is not None:
            self.o.set_footprint(polygon)
        if static is not None:
            self.o.add_static(static, res=res)
            self.fl.add_static_map(static)
        self.o.add_obstacle()
        self.fl.configure_obstacle()
        if inflation:
            self.o.add_inflation(inflation[0], inflation[1], exact=True)
        if polygon is not None:
            self.o.set_footprint(polygon)
            self.fl.set_footprint(polygon)
        if inflation:
            self.fl.configure_inflation(inflation[0], inflation[1], self.o.inscribed_radius)

    def add_observation(self, pts, origin=(0.0, 0.0, MAX_Z)):
        self.o.add_observation(pts, origin=origin)
        self.obs.append(dict(instance=0, points=np.asarray(pts, np.float32), origin=origin, obstacle_range=100.0,
                             raytrace_range=100.0))

    def update(self, rx=0.0, ry=0.0, ryaw=0.0):
        self.o.update_map(rx, ry, ryaw)
        self.fl.stage_observations([[rx, ry, ryaw]], self.obs)
        self.fl.update_map()
        m = self.fl.master()[0]
        assert np.array_equal(m, self.o.master()), "GPU master grid != oracle"
        assert np.array_equal(self.fl.download(self.N.GRID_OBSTACLE)[0], self.o.layer(2)), "GPU obstacle layer != oracle"
        assert np.array_equal(self.fl.bounds()[0], self.o.bounds()), "update box differs"
        return m


def count(m, v, equal=True):
    return int((m == v).sum()) if equal else int((m != v).sum())


def test_reference_obstacle_scenarios(nav, orc, ten_by_ten):  # costmap_2d/test/obstacle_tests.cpp
    g = GpuLayered(nav, orc, 10, static=ten_by_ten)
    g.add_observation([[0.0, 0.0, MAX_Z / 2]], origin=(0, 0, MAX_Z / 2))
    assert count(g.update(), LETHAL) == 21
    g = GpuLayered(nav, orc, 10, static=ten_by_ten)
    assert count(g.update(), LETHAL) == 20
    g.add_observation([[9.5, 9.5, MAX_Z / 2]], origin=(0.5, 0.5, MAX_Z / 2))
    assert count(g.update(), LETHAL) == 21
    layer = g.o.layer(2)
    for i in range(10):
        layer[i, i] = LETHAL
    g.o.set_layer(layer, 2)
    g.fl.upload(g.N.GRID_OBSTACLE, layer)
    m = g.update()
    assert count(m, LETHAL) == 21 and count(m, FREE) == 79
    g = GpuLayered(nav, orc, 10, track_unknown=True)
    for p in (3.0, 5.0, 7.0):
        g.add_observation([[p, p, MAX_Z]])
    m = g.update()
    assert count(m, LETHAL) == 3 and count(m, NOINFO) == 92 and count(m, FREE) == 5
    g = GpuLayered(nav, orc, 10, track_unknown=True)
    g.add_observation([[0.0, 5.0, 0.4]])
    g.add_observation([[1.0, 5.0, 2.2]])
    assert count(g.update(), LETHAL) == 1


def test_reference_inflation_scenarios(nav, orc, ten_by_ten):  # costmap_2d/test/inflation_tests.cpp
    g = GpuLayered(nav, orc, 10, static=ten_by_ten, inflation=(1.0, 1.0), polygon=_radii(1, 1))
    m = g.update()
    assert count(m, LETHAL) == 20 and count(m, INSCRIBED) == 28
    g.add_observation([[0, 0, 0.4]])
    m = g.update()
    assert count(m, LETHAL) + count(m, INSCRIBED) == 51
    g.add_observation([[2, 0, 0.0]])
    m = g.update()
    assert count(m, LETHAL) + count(m, INSCRIBED) == 54
    g.add_observation([[1, 9, 0.0]])
    m = g.update()
    assert m[9, 1] == LETHAL and m[9, 0] == INSCRIBED and m[9, 2] == INSCRIBED
    g.add_observation([[0, 9, 0.0]])
    assert g.update()[9, 0] == LETHAL
    g = GpuLayered(nav, orc, 10, inflation=(3.0, 1.0), polygon=_radii(1, 1.75))
    g.add_observation([[5, 5, MAX_Z]])
    for _ in range(2):
        m = g.update()
        assert count(m, FREE, False) == 29 and count(m, LETHAL) == 1 and count(m, INSCRIBED) == 4
    g = GpuLayered(nav, orc, 10, inflation=(4.1, 1.0), polygon=_radii(2.1, 2.3))
    g.add_observation([[0, 0, MAX_Z]])
    m = g.update()
    assert m[0, 0] == LETHAL and m[0, 1] == INSCRIBED and m[0, 2] == INSCRIBED and m[0, 3] < INSCRIBED and m[1, 1] == INSCRIBED
    assert count(m, NOINFO) == 0  # testInflationShouldNotCreateUnknowns (inflation_tests.cpp:156-175)


def test_costmap_cycles_synthetic_fleet(nav, orc):
    """Several update cycles of a small fleet on 400x400 maps with LaserScan clouds: master grid,
    obstacle layer and update box bit-exact vs the oracle every cycle."""
    from navigation_amd import synth
    N = L(nav)
    n, nI = 400, 3
    insts = [synth.make_instance(n, i) for i in range(nI)]
    insc = synth.inscribed_radius(synth.FOOTPRINT)
    fl = nav.Fleet(nI, n, n, synth.RES, layers=N.LAYER_STATIC | N.LAYER_OBSTACLE | N.LAYER_INFLATION, max_points=720,
                   max_observations=2)
    fl.configure_obstacle()
    fl.set_footprint(synth.FOOTPRINT)
    fl.configure_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, insc)
    oracles = []
    for i, ins in enumerate(insts):
        occ = np.where(ins["cells"] == 254, 100, 0).astype(np.int8)
        fl.add_static_map(occ, first=i, count=1)
        o = orc.LayeredCostmap(False)
        o.set_footprint(synth.FOOTPRINT)
        o.add_static(occ, res=synth.RES)
        o.add_obstacle()
        o.add_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, exact=True)
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
        ol = fl.download(N.GRID_OBSTACLE)
        b = fl.bounds()
        for i in range(nI):
            assert np.array_equal(b[i], oracles[i].bounds()), (cyc, i, b[i], oracles[i].bounds())
            assert np.array_equal(ol[i], oracles[i].layer(2)), (cyc, i)
            assert np.array_equal(m[i], oracles[i].master()), (cyc, i)
    fl.close()


# ----------------------------------------------------------------------------------------------
# planner: MapGrid wavefronts, per-sample costs, selection, oscillation
# ----------------------------------------------------------------------------------------------
def _inflated_instance(orc, n, idx, synth):
    ins = synth.make_instance(n, idx)
    insc = synth.inscribed_radius(synth.FOOTPRINT)
    ins["master"] = orc.inflate(ins["cells"], synth.RES, synth.INFLATION_RADIUS, synth.COST_SCALING, insc, exact=True)
    return ins


def _check_planner(nav, orc, n, cfg_kw, n_inst=2, footprint=None, allow_unknown=1, unknown_frac=0.0, cycles=1, seed0=0,
                   plan_len=None, near_obstacles=0, map_grid_options=()):
    from navigation_amd import synth
    N = L(nav)
    fp = synth.FOOTPRINT if footprint is None else footprint
    cfg = nav.DwaConfig(allow_unknown=allow_unknown, **cfg_kw)
    ocfg = orc.DwaConfig(**cfg.as_dict())
    fl = nav.Fleet(n_inst, n, n, synth.RES, layers=N.LAYER_OBSTACLE, keep_sample_costs=True, max_sim_steps=128, max_plan=256)
    fl.configure_planner(cfg)
    fl.set_footprint(fp)
    insts = [_inflated_instance(orc, n, seed0 + i, synth) for i in range(n_inst)]
    rs = np.random.RandomState(99)
    for ins in insts:
        if near_obstacles:  # lethal cells 0.35 - 1.2 m from the robot (inside the disc make_instance keeps clear): some samples collide
            cx, cy = ins["pos"][0] / synth.RES, ins["pos"][1] / synth.RES
            for _ in range(near_obstacles):
                a, d = rs.uniform(0, 2 * np.pi), rs.uniform(0.35, 1.2) / synth.RES
                x, y = int(cx + d * np.cos(a)), int(cy + d * np.sin(a))
                if 0 <= x < n and 0 <= y < n:
                    ins["cells"][y, x] = LETHAL
            ins["master"] = orc.inflate(ins["cells"], synth.RES, synth.INFLATION_RADIUS, synth.COST_SCALING, synth.inscribed_radius(synth.FOOTPRINT), exact=True)
        if unknown_frac:
            m = ins["master"]
            m[(rs.random_sample(m.shape) < unknown_frac) & (m == 0)] = NOINFO
    fl.upload(N.GRID_MASTER, np.stack([i["master"] for i in insts]))
    planners = [orc.DwaPlanner(i["master"], synth.RES, 0.0, 0.0, ocfg) for i in insts]
    for critic, agg, ysh in map_grid_options:  # MapGridCostFunction's aggregationType / yshift, beyond DWAPlanner's own wiring
        fl.set_map_grid_options(critic, agg, ysh)
        for p in planners:
            p.set_map_grid_options(critic, agg, ysh)
    fl.set_plan()
    for p in planners:
        p.set_plan()
    for cyc in range(cycles):
        pos = np.stack([i["pos"] for i in insts]).copy()
        vel = np.stack([i["vel"] for i in insts]).copy()
        if cyc:
            pos[:, 0] += 0.03 * cyc
            pos[:, 2] += 0.4 * cyc
            vel[:, 2] = -vel[:, 2]
        plans = np.stack([i["plan"][:plan_len] for i in insts])
        res = fl.find_best_path(pos, vel, plans)
        for k, ins in enumerate(insts):
            ores, otraj, cref, cfull, ostatus = planners[k].cycle(pos[k], vel[k], plans[k], fp)
            # MapGrid grids, bit-exact
            for gid, which in ((N.GRID_PATH, 0), (N.GRID_GOAL, 1), (N.GRID_GOAL_FRONT, 2)):
                g = fl.download(gid, k, 1)[0]
                og = planners[k].grid(which)
                assert np.array_equal(g.astype(np.float64), og), f"MapGrid {which} differs (inst {k}, cycle {cyc})"
            cost, status, vels = fl.samples(k)
            assert len(cost) == ores.n_samples == res[k].n_samples
            # VelocityIterator / SimpleTrajectoryGenerator::initialise: every slot's (vx, vy, vtheta), bit for bit
            assert np.array_equal(vels.view(np.uint32), planners[k].samples().view(np.uint32)), "sample velocities differ"
            assert np.array_equal(status, ostatus), "generator accept/reject mask differs"
            scored = status == 1
            # footprint-collision / failure-code mask bit-exact, costs within 1e-5
            neg_o, neg_g = cfull[scored] < 0, cost[scored] < 0
            assert np.array_equal(neg_o, neg_g), "valid/invalid mask differs"
            assert np.array_equal(cfull[scored][neg_o], cost[scored][neg_g]), "failure codes differ"
            assert np.allclose(cost[scored][~neg_g], cfull[scored][~neg_o], rtol=0, atol=1e-5)
            r = res[k]
            assert r.best_index == ores.best_index, (r.best_index, ores.best_index)
            assert r.n_scored == ores.n_scored and r.n_valid == ores.n_valid
            assert abs(r.cost - ores.cost) <= 1e-5
            assert r.oscillation_flags == ores.oscillation_flags
            if r.best_index >= 0:
                assert (r.xv, r.yv, r.thetav) == (ores.xv, ores.yv, ores.thetav)
                assert list(r.drive) == list(ores.drive)
                t = fl.trajectory(k)
                assert t.shape == otraj.shape and np.allclose(t, otraj, rtol=0, atol=1e-6)
    fl.close()
    return res


def test_planner_config1_shape(nav, orc):
    # BASELINE config 1 shape: 200x200, 10x10x5 samples, 10 steps
    _check_planner(nav, orc, 200, dict(vx_samples=10, vy_samples=10, vth_samples=5, sim_time=1.0, sim_granularity=0.1,
                                       discretize_by_time=1), n_inst=3, cycles=3)


def test_planner_config2_shape(nav, orc):
    # BASELINE config 2 shape: 400x400, 32x32x16 samples, 20 steps
    res = _check_planner(nav, orc, 400, dict(vx_samples=32, vy_samples=32, vth_samples=16, sim_time=2.0, sim_granularity=0.1,
                                             discretize_by_time=1), n_inst=2)
    assert res[0].n_samples > 16384


def test_planner_reference_defaults_variable_steps(nav, orc):
    # the reference's own defaults: discretize_by_time = false -> per-sample step counts (up to 38)
    _check_planner(nav, orc, 200, dict(), n_inst=2, cycles=2)


def test_planner_unknown_cells_and_polygon5(nav, orc):
    from navigation_amd import synth
    for au in (0, 1):
        _check_planner(nav, orc, 200, dict(vx_samples=8, vy_samples=6, vth_samples=9, sim_time=1.5, sim_granularity=0.1,
                                           discretize_by_time=1), n_inst=2, footprint=synth.FOOTPRINT5, allow_unknown=au,
                       unknown_frac=0.01, seed0=10)


def test_planner_sum_scores_and_zero_scales(nav, orc):
    _check_planner(nav, orc, 160, dict(vx_samples=6, vy_samples=5, vth_samples=7, sim_time=1.2, sim_granularity=0.1,
                                       discretize_by_time=1, sum_scores=1, occdist_scale=0.02), n_inst=2, seed0=20)
    _check_planner(nav, orc, 160, dict(vx_samples=6, vy_samples=5, vth_samples=7, sim_time=1.2, sim_granularity=0.1,
                                       discretize_by_time=1, occdist_scale=0.0, forward_point_distance=0.0), n_inst=2, seed0=22)


def test_planner_oscillation_flags_persist(nav, orc):
    from navigation_amd import synth
    N = L(nav)
    n = 160
    cfg = nav.DwaConfig(vx_samples=6, vy_samples=6, vth_samples=8, sim_time=1.0, sim_granularity=0.1, discretize_by_time=1,
                        min_vel_x=-0.3)
    ocfg = orc.DwaConfig(**cfg.as_dict())
    ins = _inflated_instance(orc, n, 3, synth)
    fl = nav.Fleet(1, n, n, synth.RES, layers=N.LAYER_OBSTACLE, keep_sample_costs=True)
    fl.configure_planner(cfg)
    fl.set_footprint(synth.FOOTPRINT)
    fl.upload(N.GRID_MASTER, ins["master"])
    p = orc.DwaPlanner(ins["master"], synth.RES, 0.0, 0.0, ocfg)
    # force every combination of sticky flags through both implementations
    for flags in (0, 1 << 8, 1 << 9, (1 << 4) | (1 << 10), (1 << 5) | (1 << 0), (1 << 1) | (1 << 11), 0xFFF):
        prev = np.array([ins["pos"][0] - 0.01, ins["pos"][1], ins["pos"][2]], np.float32)
        fl.set_oscillation([flags], [prev])
        p.set_oscillation(flags, prev)
        r = fl.find_best_path([ins["pos"]], [ins["vel"]], [ins["plan"]])[0]
        o, _, _, cfull, st = p.cycle(ins["pos"], ins["vel"], ins["plan"], synth.FOOTPRINT)
        cost, status, _ = fl.samples(0)
        assert np.array_equal(cost[status == 1] < 0, cfull[st == 1] < 0)
        assert r.best_index == o.best_index and r.oscillation_flags == o.oscillation_flags
        gf, gp = fl.oscillation()
        of, op = p.oscillation()
        assert gf[0] == of and np.array_equal(gp[0], op)
    fl.close()


def test_planner_check_trajectory(nav, orc):
    from navigation_amd import synth
    N = L(nav)
    n = 160
    cfg = nav.DwaConfig(vx_samples=6, vy_samples=6, vth_samples=8, sim_time=1.0, sim_granularity=0.1, discretize_by_time=1)
    ocfg = orc.DwaConfig(**cfg.as_dict())
    ins = _inflated_instance(orc, n, 4, synth)
    # a wall right in front of the robot so some samples collide
    m = ins["master"]
    cx = int(ins["pos"][0] / synth.RES)
    m[:, cx + 12] = LETHAL
    fl = nav.Fleet(1, n, n, synth.RES, layers=N.LAYER_OBSTACLE)
    fl.configure_planner(cfg)
    fl.set_footprint(synth.FOOTPRINT)
    fl.upload(N.GRID_MASTER, m)
    p = orc.DwaPlanner(m, synth.RES, 0.0, 0.0, ocfg)
    pos = ins["pos"].copy()
    pos[2] = 0.0
    fl.find_best_path([pos], [ins["vel"]], [ins["plan"]])
    p.cycle(pos, ins["vel"], ins["plan"], synth.FOOTPRINT)
    for vs in ([0.5, 0.0, 0.0], [0.1, 0.0, 0.5], [0.0, 0.0, 0.0], [0.3, 0.1, -0.4], [0.55, 0.0, 0.0]):
        assert fl.check_trajectory(0, vs) == p.check_trajectory(pos, ins["vel"], vs), vs
    fl.close()


def test_fleet_batch_equals_single(nav, orc):
    """Instances are independent: a batched launch gives each robot what a fleet of one gives it."""
    from navigation_amd import synth
    N = L(nav)
    n, nI = 200, 5
    cfg = nav.DwaConfig(vx_samples=10, vy_samples=10, vth_samples=5, sim_time=1.0, sim_granularity=0.1, discretize_by_time=1)
    insts = [_inflated_instance(orc, n, 30 + i, synth) for i in range(nI)]
    big = nav.Fleet(nI, n, n, synth.RES, layers=N.LAYER_OBSTACLE)
    big.configure_planner(cfg)
    big.set_footprint(synth.FOOTPRINT)
    big.upload(N.GRID_MASTER, np.stack([i["master"] for i in insts]))
    rb = big.find_best_path(np.stack([i["pos"] for i in insts]), np.stack([i["vel"] for i in insts]),
                            np.stack([i["plan"] for i in insts]))
    for k, ins in enumerate(insts):
        one = nav.Fleet(1, n, n, synth.RES, layers=N.LAYER_OBSTACLE)
        one.configure_planner(cfg)
        one.set_footprint(synth.FOOTPRINT)
        one.upload(N.GRID_MASTER, ins["master"])
        r1 = one.find_best_path([ins["pos"]], [ins["vel"]], [ins["plan"]])[0]
        assert (r1.best_index, r1.cost, r1.n_valid) == (rb[k].best_index, rb[k].cost, rb[k].n_valid)
        one.close()
    big.close()


def test_full_size_properties(nav):
    """BASELINE config-3 scale (64 instances here to bound memory/time): size-independent
    properties — inflation is idempotent, never lowers a cell, never creates unknowns; the planner
    returns a valid winner whose cost is the minimum of all sample costs."""
    from navigation_amd import synth
    N = L(nav)
    n, nI = 400, 64
    fl = nav.Fleet(nI, n, n, synth.RES, layers=N.LAYER_OBSTACLE | N.LAYER_INFLATION, keep_sample_costs=True)
    fl.configure_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, synth.inscribed_radius(synth.FOOTPRINT))
    fl.set_footprint(synth.FOOTPRINT)
    cfg = synth.fleet_config()
    fl.configure_planner(cfg)
    insts = [synth.make_instance(n, i) for i in range(nI)]
    maps = np.stack([i["cells"] for i in insts])
    fl.upload(N.GRID_MASTER, maps)
    fl.inflate(boxes=[[0, 0, n, n]] * nI)
    a = fl.master()
    assert (a >= maps).all() and (a != NOINFO).all() and ((a == LETHAL) == (maps == LETHAL)).all()
    fl.inflate(boxes=[[0, 0, n, n]] * nI)
    assert np.array_equal(fl.master(), a), "inflation is not idempotent"
    res = fl.find_best_path(np.stack([i["pos"] for i in insts]), np.stack([i["vel"] for i in insts]),
                            np.stack([i["plan"] for i in insts]))
    for k in (0, 17, 63):
        cost, status, _ = fl.samples(k)
        ok = (status == 1) & (cost >= 0)
        assert res[k].n_valid == ok.sum() and res[k].n_scored == (status == 1).sum()
        assert res[k].best_index == int(np.flatnonzero(ok)[np.argmin(cost[ok])])
        assert res[k].cost == cost[ok].min()
    fl.close()


# ----------------------------------------------------------------------------------------------
# voxel layer: 3-D marking + 3-D raytrace clearing
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("track_unknown,z_voxels,unknown_thr,mark_thr", [(False, 10, 15, 0), (True, 10, 15, 0), (True, 16, 0, 0), (True, 8, 5, 0),
                                                                         (True, 10, 15, 1), (False, 16, 4, 2)])
def test_voxel_layer_cycles(nav, orc, track_unknown, z_voxels, unknown_thr, mark_thr):
    from navigation_amd import synth
    N = L(nav)
    n, nI = 200, 2
    insts = [synth.make_instance(n, 50 + i) for i in range(nI)]
    insc = synth.inscribed_radius(synth.FOOTPRINT5)
    fl = nav.Fleet(nI, n, n, synth.RES, layers=N.LAYER_VOXEL | N.LAYER_INFLATION, track_unknown=track_unknown,
                   max_points=1440, max_observations=2)
    fl.configure_obstacle(z_voxels=z_voxels, origin_z=0.0, z_resolution=0.2, unknown_threshold=unknown_thr, mark_threshold=mark_thr,
                          max_obstacle_height=2.0)
    fl.set_footprint(synth.FOOTPRINT5)
    fl.configure_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, insc)
    oracles = []
    for ins in insts:
        o = orc.LayeredCostmap(track_unknown)
        o.resize(n, n, synth.RES, 0, 0)
        o.set_footprint(synth.FOOTPRINT5)
        o.add_voxel(z_voxels=z_voxels, origin_z=0.0, z_resolution=0.2, unknown_threshold=unknown_thr, mark_threshold=mark_thr,
                    max_obstacle_height=2.0)
        o.add_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, exact=True)
        o.set_footprint(synth.FOOTPRINT5)
        oracles.append(o)
    for cyc in range(4):
        obs, poses = [], []
        for i, ins in enumerate(insts):
            pts = synth.laser_scan(ins, cyc, z=0.3, z_jitter=1.5)
            if cyc == 2:  # some points above max_obstacle_height / below the floor
                pts[::7, 2] = 2.4
                pts[3::11, 2] = -0.2
            org = (float(ins["pos"][0]), float(ins["pos"][1]), 0.3 + 0.25 * cyc)
            obs.append(dict(instance=i, points=pts, origin=org, obstacle_range=2.5, raytrace_range=3.0))
            poses.append([float(v) for v in ins["pos"]])
            oracles[i].clear_observations()
            oracles[i].add_observation(pts, origin=org, obstacle_range=2.5, raytrace_range=3.0)
            if mark_thr:  # a second sensor hitting the same columns at other heights: the order across observations counts
                pts2 = pts[::2].copy()
                pts2[:, 2] = np.clip(pts2[:, 2] + 0.45, 0.0, 1.9)
                org2 = (org[0] + 0.05, org[1] - 0.05, 0.9)
                obs.append(dict(instance=i, points=pts2, origin=org2, obstacle_range=2.5, raytrace_range=3.0))
                oracles[i].add_observation(pts2, origin=org2, obstacle_range=2.5, raytrace_range=3.0)
            oracles[i].update_map(*poses[-1])
        fl.stage_observations(poses, obs)
        fl.update_map()
        vox = fl.download(N.GRID_VOXEL)
        ol = fl.download(N.GRID_OBSTACLE)
        m = fl.master()
        b = fl.bounds()
        for i in range(nI):
            assert np.array_equal(vox[i], oracles[i].voxels()), ("voxel columns", cyc, i)
            assert np.array_equal(ol[i], oracles[i].layer(2)), ("voxel layer 2-D grid", cyc, i)
            assert np.array_equal(b[i], oracles[i].bounds()), ("box", cyc, i)
            assert np.array_equal(m[i], oracles[i].master()), ("master", cyc, i)
    fl.close()


# ----------------------------------------------------------------------------------------------
# larger grids: the RPT=12 LDS wavefront (600x600) and the global-memory fallback (1000x1000)
# ----------------------------------------------------------------------------------------------
def test_planner_600x600(nav, orc):
    _check_planner(nav, orc, 600, dict(vx_samples=8, vy_samples=6, vth_samples=9, sim_time=1.5, sim_granularity=0.1,
                                       discretize_by_time=1), n_inst=1, seed0=40)


def test_planner_config5_shape_1000x1000(nav, orc):
    """BASELINE config 5 shape on one instance: 1000x1000 map, 5-vertex polygon footprint,
    64x64x32 velocity samples, 20 steps (the voxel layer of config 5 is covered by
    test_voxel_layer_cycles)."""
    from navigation_amd import synth
    res = _check_planner(nav, orc, 1000, dict(vx_samples=64, vy_samples=64, vth_samples=32, sim_time=2.0, sim_granularity=0.1,
                                              discretize_by_time=1), n_inst=1, footprint=synth.FOOTPRINT5, seed0=60)
    assert res[0].n_samples > 131072


# ----------------------------------------------------------------------------------------------
# rolling window (SURVEY f-2): Costmap2D::updateOrigin / VoxelLayer::updateOrigin on the device
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("voxel", [False, True])
def test_rolling_window_cycles(nav, orc, voxel):
    from navigation_amd import synth
    N = L(nav)
    n, nI = 120, 3
    insc = synth.inscribed_radius(synth.FOOTPRINT)
    layers = (N.LAYER_VOXEL if voxel else N.LAYER_OBSTACLE) | N.LAYER_INFLATION
    fl = nav.Fleet(nI, n, n, synth.RES, layers=layers, track_unknown=True, max_points=720, max_observations=1, rolling_window=True)
    fl.configure_obstacle()
    fl.set_footprint(synth.FOOTPRINT)
    fl.configure_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, insc)
    rs = np.random.RandomState(3)
    starts = rs.uniform(2.0, 4.0, (nI, 2))
    oracles = []
    for i in range(nI):
        o = orc.LayeredCostmap(True)
        o.resize(n, n, synth.RES, 0, 0)
        o.set_rolling(True)
        o.set_footprint(synth.FOOTPRINT)
        if voxel:
            o.add_voxel()
        else:
            o.add_obstacle()
        o.add_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, exact=True)
        o.set_footprint(synth.FOOTPRINT)
        oracles.append(o)
    world = synth.make_instance(400, 77)  # a fixed 20 m world the robots drive through
    for cyc in range(6):
        poses, obs = [], []
        for i in range(nI):
            # motion incl. backwards and a jump larger than the window (everything scrolls out)
            x = starts[i, 0] + 0.37 * cyc * (1 if i != 1 else -0.5) + (9.0 if (cyc == 4 and i == 2) else 0.0)
            y = starts[i, 1] + 0.21 * cyc
            yaw = 0.3 * cyc - 0.5 * i
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
        m = fl.master()
        ol = fl.download(N.GRID_OBSTACLE)
        org_g = fl.origins()
        b = fl.bounds()
        vx = fl.download(N.GRID_VOXEL) if voxel else None
        for i in range(nI):
            assert np.array_equal(org_g[i], oracles[i].origin()), ("origin", cyc, i)
            assert np.array_equal(b[i], oracles[i].bounds()), ("box", cyc, i)
            assert np.array_equal(ol[i], oracles[i].layer(2)), ("layer", cyc, i)
            if voxel:
                assert np.array_equal(vx[i], oracles[i].voxels()), ("voxels", cyc, i)
            assert np.array_equal(m[i], oracles[i].master()), ("master", cyc, i)
    fl.close()


# ----------------------------------------------------------------------------------------------
# edge cases: empty / ragged inputs, robots and plans off the map, degenerate footprints, capacities
# ----------------------------------------------------------------------------------------------
def _planner_pair(nav, orc, n, master, cfg_kw, fp, max_fp=16):
    from navigation_amd import synth
    N = L(nav)
    cfg = nav.DwaConfig(**cfg_kw)
    fl = nav.Fleet(1, n, n, synth.RES, layers=N.LAYER_OBSTACLE, keep_sample_costs=True, max_sim_steps=64, max_footprint=max_fp)
    fl.configure_planner(cfg)
    fl.set_footprint(fp)
    fl.upload(N.GRID_MASTER, master)
    p = orc.DwaPlanner(master, synth.RES, 0.0, 0.0, orc.DwaConfig(**cfg.as_dict()))
    return fl, p


def _compare_cycle(fl, p, pos, vel, plan, fp):
    r = fl.find_best_path([pos], [vel], [plan])[0]
    o, otraj, _, cfull, ost = p.cycle(np.asarray(pos, np.float32), np.asarray(vel, np.float32), plan, fp)
    cost, status, vels = fl.samples(0)
    assert np.array_equal(vels.view(np.uint32), p.samples().view(np.uint32)), "sample velocities differ"
    assert np.array_equal(status, ost)
    sc = status == 1
    assert np.array_equal(cost[sc] < 0, cfull[sc] < 0)
    assert np.array_equal(cost[sc][cost[sc] < 0], cfull[sc][cfull[sc] < 0])
    assert np.allclose(cost[sc][cost[sc] >= 0], cfull[sc][cfull[sc] >= 0], rtol=0, atol=1e-5)
    assert (r.best_index, r.n_valid, r.n_scored) == (o.best_index, o.n_valid, o.n_scored)
    assert abs(r.cost - o.cost) <= 1e-5 and list(r.drive) == list(o.drive)
    return r, cost, status


def test_planner_edge_cases(nav, orc):
    from navigation_amd import synth
    n = 120
    cfgk = dict(vx_samples=6, vy_samples=4, vth_samples=7, sim_time=1.2, sim_granularity=0.1, discretize_by_time=1, min_vel_x=-0.2)
    ins = _inflated_instance(orc, n, 5, synth)
    m = ins["master"]
    plan_in = ins["plan"]
    fp = synth.FOOTPRINT
    fl, p = _planner_pair(nav, orc, n, m, cfgk, fp)
    size = n * synth.RES
    # robot close to the map border: samples leave the map (-4 / -6 codes)
    r, cost, st = _compare_cycle(fl, p, [size - 0.3, size / 2, 0.0], [0.3, 0.0, 0.0], plan_in, fp)
    assert (cost[st == 1] < 0).any()
    # robot off the map entirely: everything invalid, cost -7, zero drive
    r, cost, st = _compare_cycle(fl, p, [size + 1.0, size / 2, 0.5], [0.0, 0.0, 0.0], plan_in, fp)
    assert r.best_index == -1 and r.cost == -7.0 and list(r.drive) == [0.0, 0.0, 0.0]
    # plan entirely outside the map: MapGrids stay unreachable -> path/goal critics reject (-2)
    far = np.stack([np.linspace(50, 55, 20), np.linspace(50, 55, 20)], 1)
    r, cost, st = _compare_cycle(fl, p, [size / 2, size / 2, 0.0], [0.1, 0.0, 0.0], far, fp)
    assert r.best_index == -1
    # single-pose plan and a plan whose first poses are off the map
    _compare_cycle(fl, p, [size / 2, size / 2, 1.0], [0.1, 0.0, 0.1], plan_in[:1], fp)
    ragged = np.concatenate([np.array([[-3.0, -3.0], [-1.0, -1.0]]), plan_in[:60]])
    _compare_cycle(fl, p, [size / 2, size / 2, -2.0], [0.2, 0.0, -0.2], ragged, fp)
    # robot sitting inside a lethal blob: every sample collides (-6)
    m2 = m.copy()
    c = int(size / 2 / synth.RES)
    m2[c - 3:c + 4, c - 3:c + 4] = LETHAL
    fl.upload(L(nav).GRID_MASTER, m2)
    p.set_costmap(m2)
    r, cost, st = _compare_cycle(fl, p, [size / 2, size / 2, 0.3], [0.0, 0.0, 0.0], plan_in, fp)
    assert r.best_index == -1 and (cost[st == 1] == -6).all()
    fl.close()
    # degenerate footprints: 2 vertices (centre-cell rule incl. INSCRIBED) and none (-9)
    for fp2 in (np.array([[0.1, 0.0], [-0.1, 0.0]]), np.zeros((0, 2))):
        fl, p = _planner_pair(nav, orc, n, m, cfgk, fp2)
        r, cost, st = _compare_cycle(fl, p, [size / 2, size / 2, 0.0], [0.2, 0.0, 0.0], plan_in, fp2)
        if len(fp2) == 0:
            assert (cost[st == 1] == -9).all()
        fl.close()
    # 16-vertex circle (makeFootprintFromRadius, footprint.cpp:150-167)
    ang = np.arange(16) * 2 * np.pi / 16
    circ = np.stack([np.cos(ang) * 0.25, np.sin(ang) * 0.25], 1)
    fl, p = _planner_pair(nav, orc, n, m, cfgk, circ)
    _compare_cycle(fl, p, [size / 2, size / 2, 0.7], [0.2, 0.0, 0.1], plan_in, circ)
    fl.close()


def test_planner_cost_cloud(nav, orc):
    """DWAPlanner::getCellCosts over the map in MapGridVisualizer::publishCostCloud's order (dwa_planner.cpp:185-202,
    map_grid_visualizer.cpp:55-83), restated with numpy on the oracle's grids."""
    from navigation_amd import synth
    N = L(nav)
    n = 120
    cfg = nav.DwaConfig(vx_samples=4, vy_samples=2, vth_samples=4, sim_time=1.0, sim_granularity=0.1, discretize_by_time=1)
    ins = _inflated_instance(orc, n, 21, synth)
    fl, p = _planner_pair(nav, orc, n, ins["master"], cfg.as_dict(), synth.FOOTPRINT)
    fl.set_origin(np.array([[1.5, -2.0]]))
    p2 = orc.DwaPlanner(ins["master"], synth.RES, 1.5, -2.0, orc.DwaConfig(**cfg.as_dict()))
    p2.set_plan()
    fl.set_plan()
    pos = np.array(ins["pos"], np.float32) + np.array([1.5, -2.0, 0.0], np.float32)
    plan = ins["plan"] + np.array([1.5, -2.0])
    fl.find_best_path([pos], [ins["vel"]], [plan])
    p2.cycle(pos, ins["vel"], plan, synth.FOOTPRINT)
    path32, goal32 = p2.grid(0).astype(np.float32), p2.grid(1).astype(np.float32)
    occ32 = ins["master"].astype(np.float32)
    ok = ~((path32 == n * n) | (path32 == n * n + 1) | (occ32 >= 253))
    total = (cfg.path_distance_bias * synth.RES * path32.astype(np.float64) + cfg.goal_distance_bias * synth.RES * goal32.astype(np.float64) +
             cfg.occdist_scale * occ32.astype(np.float64)).astype(np.float32)
    want = []
    for cx in range(n):
        for cy in range(n):
            if ok[cy, cx]:
                want.append((np.float32(1.5 + (cx + 0.5) * synth.RES), np.float32(-2.0 + (cy + 0.5) * synth.RES), np.float32(0.0),
                             path32[cy, cx], goal32[cy, cx], occ32[cy, cx], total[cy, cx]))
    want = np.asarray(want, np.float32)
    got = fl.cost_cloud(0)
    assert got.shape == want.shape and len(got) > 1000
    assert np.array_equal(got, want)
    fl.close()


def test_costmap_publisher_export(nav):
    """Costmap2DPublisher's occupancy view (costmap_2d_publisher.cpp:57-74 table; :103-115 full grid; :146-156 window)."""
    N = L(nav)
    n = 96
    rs = np.random.RandomState(2)
    m = rs.randint(0, 256, (2, n, n)).astype(np.uint8)
    m[0, :4, :4] = [[0, 1, 2, 126], [127, 128, 251, 252], [253, 254, 255, 0], [1, 252, 253, 255]]
    table = np.zeros(256, np.int8)  # the reference's table, rebuilt from its text
    table[0], table[253], table[254], table[255] = 0, 99, 100, -1
    for i in range(1, 253):
        table[i] = 1 + (97 * (i - 1)) // 251
    fl = nav.Fleet(2, n, n, 0.05, layers=N.LAYER_OBSTACLE)
    fl.upload(N.GRID_MASTER, m)
    for k in range(2):
        assert np.array_equal(fl.export_occupancy(k), table[m[k]])
        assert np.array_equal(fl.export_occupancy(k, 5, 7, 61, 40), table[m[k][7:40, 5:61]])
        assert np.array_equal(fl.export_occupancy(k, n - 1, n - 1, n, n), table[m[k][n - 1:, n - 1:]])
    with pytest.raises(nav.NavgpuError):
        fl.export_occupancy(0, 10, 10, 10, 20)  # empty window
    with pytest.raises(nav.NavgpuError):
        fl.export_occupancy(0, 0, 0, n + 1, n)
    fl.close()


def test_costmap_edge_cases(nav, orc):
    from navigation_amd import synth
    N = L(nav)
    n = 80
    fl = nav.Fleet(1, n, n, synth.RES, layers=N.LAYER_OBSTACLE | N.LAYER_INFLATION, track_unknown=True, max_points=64,
                   max_observations=3)
    fl.configure_obstacle()
    fl.set_footprint(synth.FOOTPRINT)
    insc = synth.inscribed_radius(synth.FOOTPRINT)
    fl.configure_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, insc)
    o = orc.LayeredCostmap(True)
    o.resize(n, n, synth.RES, 0, 0)
    o.set_footprint(synth.FOOTPRINT)
    o.add_obstacle()
    o.add_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, exact=True)
    o.set_footprint(synth.FOOTPRINT)

    def cycle(pose, observations):
        o.clear_observations()
        for ob in observations:
            o.add_observation(ob["points"], origin=ob["origin"], obstacle_range=ob["obstacle_range"],
                              raytrace_range=ob["raytrace_range"], marking=ob.get("marking", True), clearing=ob.get("clearing", True))
        o.update_map(*pose)
        fl.stage_observations([pose], observations)
        fl.update_map()
        assert np.array_equal(fl.bounds()[0], o.bounds())
        assert np.array_equal(fl.download(N.GRID_OBSTACLE)[0], o.layer(2))
        assert np.array_equal(fl.master()[0], o.master())

    size = n * synth.RES
    cycle([2.0, 2.0, 0.3], [])  # no observation at all: only the footprint touches the bounds
    pts = np.array([[2.5, 2.0, 0.2], [9.0, 2.0, 0.2], [2.0, -1.0, 0.2], [2.2, 2.9, 3.0], [3.9, 3.9, 0.1], [2.0, 2.0, 0.1]], np.float32)
    cycle([2.0, 2.0, 0.3], [dict(instance=0, points=pts, origin=(2.0, 2.0, 0.3), obstacle_range=2.5, raytrace_range=3.0)])
    # sensor outside the map: no raytracing for that observation, marking still happens
    cycle([2.0, 2.0, 0.3], [dict(instance=0, points=pts, origin=(-0.5, 2.0, 0.3), obstacle_range=4.0, raytrace_range=3.0)])
    # marking-only + clearing-only + empty cloud in one cycle; footprint partly off the map (no clearing polygon)
    cycle([0.1, 0.1, 0.0], [dict(instance=0, points=pts[:3], origin=(1.0, 1.0, 0.3), obstacle_range=2.5, raytrace_range=3.0, clearing=False),
                            dict(instance=0, points=pts[3:], origin=(1.0, 1.0, 0.3), obstacle_range=2.5, raytrace_range=1.0, marking=False),
                            dict(instance=0, points=np.zeros((0, 3), np.float32), origin=(1.0, 1.0, 0.3), obstacle_range=2.5, raytrace_range=3.0)])
    cycle([size - 0.05, size - 0.05, 1.0], [])
    # capacities are enforced, not overrun
    with pytest.raises(nav.NavgpuError):
        fl.stage_observations([[2.0, 2.0, 0.0]], [dict(instance=0, points=np.zeros((65, 3), np.float32), origin=(2, 2, 0.3))])
    with pytest.raises(nav.NavgpuError):
        fl.stage_observations([[2.0, 2.0, 0.0]], [dict(instance=0, points=pts, origin=(2, 2, 0.3))] * 4)
    with pytest.raises(nav.NavgpuError):
        fl.set_footprint(np.zeros((40, 2)))
    fl.close()


# ----------------------------------------------------------------------------------------------
# MapGrid wavefronts: searches longer than one level epoch (> 1023 levels), odd and non-square sizes
# ----------------------------------------------------------------------------------------------
def _mapgrid_case(nav, orc, master, plan, pos):
    from navigation_amd import synth
    N = L(nav)
    ny, nx = master.shape
    cfg = nav.DwaConfig(vx_samples=3, vy_samples=1, vth_samples=3, sim_time=0.5, sim_granularity=0.1, discretize_by_time=1)
    fl = nav.Fleet(1, nx, ny, synth.RES, layers=N.LAYER_OBSTACLE, keep_sample_costs=True, max_sim_steps=16, max_plan=max(16, len(plan)))
    fl.configure_planner(cfg)
    fl.set_footprint(synth.FOOTPRINT)
    fl.upload(N.GRID_MASTER, master)
    p = orc.DwaPlanner(master, synth.RES, 0.0, 0.0, orc.DwaConfig(**cfg.as_dict()))
    fl.set_plan()
    p.set_plan()
    r = fl.find_best_path([pos], [[0.0, 0.0, 0.0]], [plan])[0]
    o = p.cycle(np.asarray(pos, np.float32), np.zeros(3, np.float32), plan, synth.FOOTPRINT)[0]
    levels = 0
    for gid, which in ((N.GRID_PATH, 0), (N.GRID_GOAL, 1), (N.GRID_GOAL_FRONT, 2)):
        g = fl.download(gid, 0, 1)[0].reshape(ny, nx)
        og = p.grid(which).reshape(ny, nx)
        assert np.array_equal(g.astype(np.float64), og), f"MapGrid {which} differs ({nx}x{ny})"
        levels = max(levels, int(g[g < nx * ny].max()))
    assert (r.best_index, r.n_valid) == (o.best_index, o.n_valid)
    fl.close()
    return levels


def test_mapgrid_long_searches_and_odd_sizes(nav, orc):
    from navigation_amd import synth
    res = synth.RES
    # serpentine corridors: the wavefront needs several thousand levels (level epochs are flushed and restarted)
    for n in (200, 400):
        m = np.zeros((n, n), np.uint8)
        for k, row in enumerate(range(4, n - 2, 4)):
            m[row, :] = LETHAL
            if k % 2:
                m[row, 1:3] = 0
            else:
                m[row, n - 3:n - 1] = 0
        plan = np.stack([np.linspace(0.1, 0.6, 12), np.full(12, 0.08)], 1)  # first corridor only
        levels = _mapgrid_case(nav, orc, m, plan, [0.3, 0.08, 0.0])
        assert levels > 2 * 1023, levels
    # odd and non-square sizes (ragged last bitmap word, partial last strip), random blobs, unknown cells
    rs = np.random.RandomState(5)
    for nx, ny in ((101, 77), (64, 64), (33, 250), (400, 37), (250, 399), (31, 31), (416, 416)):
        m = np.zeros((ny, nx), np.uint8)
        for _ in range(max(3, nx * ny // 900)):
            cx, cy, r = rs.randint(0, nx), rs.randint(0, ny), rs.randint(1, 4)
            m[max(0, cy - r):cy + r + 1, max(0, cx - r):cx + r + 1] = LETHAL
        m[rs.random_sample(m.shape) < 0.01] = NOINFO
        sx, sy = nx * res, ny * res
        plan = np.stack([np.linspace(0.1 * sx, 0.9 * sx, 40), np.linspace(0.2 * sy, 0.8 * sy, 40)], 1)
        for px, py in plan:  # keep the plan itself traversable
            m[int(py / res), int(px / res)] = 0
        _mapgrid_case(nav, orc, m, plan, [0.5 * sx, 0.5 * sy, 0.3])


def test_planner_reconfigure_same_fleet(nav, orc):
    """Window, heading tables and the per-robot LDS image are re-sized when the planner configuration or the
    footprint changes on a live fleet (navgpu_planner_configure / navgpu_set_footprint called again)."""
    from navigation_amd import synth
    N = L(nav)
    n = 160
    ins = _inflated_instance(orc, n, 11, synth)
    m = ins["master"]
    size = n * synth.RES
    fl = nav.Fleet(1, n, n, synth.RES, layers=N.LAYER_OBSTACLE, keep_sample_costs=True, max_sim_steps=64, max_footprint=16)
    fl.upload(N.GRID_MASTER, m)
    big = np.array([[0.45, 0.3], [0.45, -0.3], [-0.45, -0.3], [-0.45, 0.3]])
    steps = [(dict(vx_samples=5, vy_samples=3, vth_samples=6, sim_time=0.8, sim_granularity=0.1, discretize_by_time=1), synth.FOOTPRINT),
             (dict(vx_samples=5, vy_samples=3, vth_samples=6, sim_time=3.0, sim_granularity=0.1, discretize_by_time=1), synth.FOOTPRINT),
             (dict(vx_samples=5, vy_samples=3, vth_samples=6, sim_time=3.0, sim_granularity=0.1, discretize_by_time=1), big),
             (dict(vx_samples=4, vy_samples=2, vth_samples=9, sim_time=1.5, sim_granularity=0.05, discretize_by_time=0), big)]
    for cfgk, fp in steps:
        cfg = nav.DwaConfig(**cfgk)
        fl.configure_planner(cfg)
        fl.set_footprint(fp)
        fl.set_plan()
        p = orc.DwaPlanner(m, synth.RES, 0.0, 0.0, orc.DwaConfig(**cfg.as_dict()))
        p.set_plan()
        _compare_cycle(fl, p, [size / 2, size / 2, 0.4], [0.2, 0.0, 0.1], ins["plan"], fp)
    fl.close()


# ----------------------------------------------------------------------------------------------
# DWAPlannerROS control cycle: getLocalPlan -> updatePlanAndLocalCosts ->
# DWA or latched stop-rotate, closed loop until isGoalReached; every cycle compared with the oracle
# ----------------------------------------------------------------------------------------------
def test_dwa_planner_ros_control_cycle(nav, orc):
    from navigation_amd import synth
    from oracle import local_planner_oracle as lpo
    N = L(nav)
    n, n_inst = 200, 3
    size = n * synth.RES
    cfg = nav.DwaConfig(vx_samples=8, vy_samples=3, vth_samples=9, sim_time=1.2, sim_granularity=0.1, discretize_by_time=1)
    limits = dict(xy_goal_tolerance=0.15, yaw_goal_tolerance=0.08, rot_stopped_vel=0.01, trans_stopped_vel=0.01,
                  max_rot_vel=cfg.max_rot_vel, min_rot_vel=cfg.min_rot_vel, acc_lim_x=cfg.acc_lim_x, acc_lim_y=cfg.acc_lim_y,
                  acc_lim_theta=cfg.acc_lim_theta, sim_period=cfg.sim_period, prune_plan=1, latch_xy_goal_tolerance=0)
    insts = [_inflated_instance(orc, n, 40 + i, synth) for i in range(n_inst)]
    fl = nav.Fleet(n_inst, n, n, synth.RES, layers=N.LAYER_OBSTACLE, max_sim_steps=32, max_plan=512)
    fl.configure_planner(cfg)
    fl.set_footprint(synth.FOOTPRINT)
    fl.upload(N.GRID_MASTER, np.stack([i["master"] for i in insts]))
    fl.configure_local_planner(**limits)
    # before setPlan: every robot returns false, nothing is dispatched
    r0 = fl.compute_velocity_commands(np.zeros((n_inst, 3)), np.zeros((n_inst, 3)))
    assert all(not r.ok and r.branch == N.BRANCH_NONE for r in r0)
    oracles, plans, Ts = [], [], []
    for k, ins in enumerate(insts):
        lim = dict(limits)
        lim["latch_xy_goal_tolerance"] = 0
        o = lpo.DwaPlannerRos(orc.DwaPlanner(ins["master"], synth.RES, 0.0, 0.0, orc.DwaConfig(**cfg.as_dict())), lim, n, n,
                              synth.RES, synth.FOOTPRINT)
        cx = cy = size / 2
        length = 1.6 + 0.3 * k  # short straight-ish paths inside the obstacle-free disc around the robot
        s = np.linspace(0.0, length, int(length / 0.05) + 1)
        ang = 0.5 * k
        gx, gy = cx + s * np.cos(ang), cy + s * np.sin(ang) + 0.05 * np.sin(3 * s)
        gplan = np.stack([gx, gy, np.full_like(s, ang + 1.0)], 1)  # goal heading differs from the travel direction
        T = None
        if k == 1:  # this robot's plan arrives in another frame: plan = T^-1 (global)
            T = (0.7, -0.4, 0.3)
            c, sn = np.cos(T[2]), np.sin(T[2])
            dx, dy = gplan[:, 0] - T[0], gplan[:, 1] - T[1]
            gplan = np.stack([c * dx + sn * dy, -sn * dx + c * dy, gplan[:, 2] - T[2]], 1)
        plans.append(gplan)
        Ts.append(T)
        oracles.append(o)
        fl.set_global_plan(k, gplan, T)
        o.set_plan(gplan, T)
    pose = np.array([[size / 2, size / 2, 0.5 * k + 0.2] for k in range(n_inst)])
    vel = np.zeros((n_inst, 3))
    seen = [set() for _ in range(n_inst)]
    reached = [False] * n_inst
    dt = cfg.sim_period
    for cyc in range(400):
        have = [not (cyc == 5 and k == 2) for k in range(n_inst)]  # one robot loses its pose for a cycle
        got = fl.compute_velocity_commands(pose, vel, have_pose=have)
        for k in range(n_inst):
            want = oracles[k].compute_velocity_commands(pose[k], vel[k], have_pose=have[k])
            g = got[k]
            assert bool(g.ok) == bool(want["ok"]), (cyc, k, g.branch, want)
            assert g.branch == want["branch"], (cyc, k, g.branch, want["branch"])
            assert tuple(g.cmd_vel) == tuple(float(v) for v in want["cmd"]), (cyc, k, tuple(g.cmd_vel), want["cmd"])
            assert g.local_plan_points == want["local_plan_points"] and g.trajectory_points == want["trajectory_points"]
            assert np.array_equal(fl.global_plan(k), np.asarray(oracles[k].global_plan).reshape(-1, 3)), "pruned plan differs"
            seen[k].add(g.branch)
        gr = fl.is_goal_reached(pose, vel)
        for k in range(n_inst):
            assert gr[k] == oracles[k].is_goal_reached(pose[k], vel[k])
            reached[k] = reached[k] or gr[k]
        if all(reached):
            break
        # unicycle-with-strafe integration of the commanded velocity; odometry reports the command
        for k in range(n_inst):
            c = got[k].cmd_vel
            th = pose[k, 2]
            pose[k, 0] += (c[0] * np.cos(th) - c[1] * np.sin(th)) * dt
            pose[k, 1] += (c[0] * np.sin(th) + c[1] * np.cos(th)) * dt
            pose[k, 2] += c[2] * dt
            vel[k] = c
    # parity held on every cycle above; the scenario itself must have exercised every branch
    assert sum(reached) >= 2, (reached, [sorted(s) for s in seen], pose)
    for k in range(n_inst):
        assert N.BRANCH_DWA in seen[k]
        if reached[k]:
            assert {N.BRANCH_ROTATE, N.BRANCH_AT_GOAL} <= seen[k], sorted(seen[k])
    assert any(N.BRANCH_STOP in s for s in seen) and any(N.BRANCH_NONE in s for s in seen)
    fl.close()


# ----------------------------------------------------------------------------------------------
# Legacy base_local_planner::TrajectoryPlanner: MapGrids with within_robot, every
# generateTrajectory call (sample, cost, points), the stateful selection, scoreTrajectory
# ----------------------------------------------------------------------------------------------
def _tp_compare_cycle(fl, N, oracles, pos, vel, cyc=0):
    got = fl.tp_find_best_path(pos, vel)
    for k, o in enumerate(oracles):
        want, wtraj, wsamples = o.find_best_path(pos[k], vel[k], N.TpResult, N.TpSample)
        g = got[k]
        ny, nx = o.shape
        for gid, which in ((N.GRID_PATH, 0), (N.GRID_GOAL, 1)):
            gg = fl.download(gid, k, 1)[0].reshape(ny, nx).astype(np.float64)
            assert np.array_equal(gg, o.grid(which)), f"MapGrid {which} differs (robot {k}, cycle {cyc})"
        gs = fl.tp_samples(k)
        assert len(gs) == len(wsamples) == g.n_samples == want.n_samples, (cyc, k, len(gs), len(wsamples))
        for a, b in zip(gs, wsamples):
            assert a[:3] == b[:3], ("sample velocities differ", cyc, k, a, b)
            assert (a[3] < 0) == (b[3] < 0) and a[4] == b[4], ("legality / length differs", cyc, k, a, b)
            if b[3] < 0:
                assert a[3] == b[3], ("failure code differs", cyc, k, a, b)
            else:
                assert abs(a[3] - b[3]) <= 1e-9 * max(1.0, abs(b[3])), ("cost differs", cyc, k, a, b)
        assert (g.xv, g.yv, g.thetav) == (want.xv, want.yv, want.thetav), (cyc, k, (g.xv, g.yv, g.thetav), (want.xv, want.yv, want.thetav))
        assert abs(g.cost - want.cost) <= 1e-9 * max(1.0, abs(want.cost)) and tuple(g.drive) == tuple(want.drive)
        assert g.n_points == want.n_points
        t = fl.tp_trajectory(k)
        assert t.shape == wtraj.shape and np.allclose(t, wtraj, rtol=0, atol=1e-12)
        gst, wst = fl.tp_state(k, 1)[0], o.state(N.TpState)
        assert (gst.flags, gst.prev_x, gst.prev_y, gst.escape_x, gst.escape_y, gst.escape_theta) == \
               (wst.flags, wst.prev_x, wst.prev_y, wst.escape_x, wst.escape_y, wst.escape_theta), (cyc, k, gst.flags, wst.flags)
    return got


@pytest.mark.parametrize("holonomic,dwa", [(1, 0), (0, 1)])
def test_trajectory_planner_cycles(nav, orc, holonomic, dwa):
    from navigation_amd import synth
    N = L(nav)
    n, n_inst = 160, 3
    size = n * synth.RES
    cfg = N.TpConfig(vx_samples=6, vtheta_samples=9, sim_time=1.2, sim_granularity=0.05, angular_sim_granularity=0.05,
                     holonomic_robot=holonomic, dwa=dwa)
    insts = [_inflated_instance(orc, n, 60 + i, synth) for i in range(n_inst)]
    # robot 2 starts boxed in by a lethal ring: nothing legal but backing up -> escape mode
    c = n // 2
    ring = insts[2]["master"]
    d = np.abs(np.arange(-6, 7))
    ring[c - 6:c + 7, c - 6:c + 7] = np.where((d[:, None] >= 5) | (d[None, :] >= 5), LETHAL, ring[c - 6:c + 7, c - 6:c + 7])
    fl = nav.Fleet(n_inst, n, n, synth.RES, layers=N.LAYER_OBSTACLE, max_sim_steps=64, max_plan=256)
    fl.set_footprint(synth.FOOTPRINT)
    fl.upload(N.GRID_MASTER, np.stack([i["master"] for i in insts]))
    fl.configure_trajectory_planner(cfg)
    oracles = [orc.TrajectoryPlanner(i["master"], synth.RES, cfg, synth.FOOTPRINT) for i in insts]
    for k, ins in enumerate(insts):
        fl.tp_update_plan(k, ins["plan"], compute_dists=(k == 0))
        oracles[k].update_plan(ins["plan"], compute_dists=(k == 0))
    # updatePlan(compute_dists=true) fills the grids without within_robot
    for gid, which in ((N.GRID_PATH, 0), (N.GRID_GOAL, 1)):
        assert np.array_equal(fl.download(gid, 0, 1)[0].reshape(n, n).astype(np.float64), oracles[0].grid(which))
    pos = np.array([[size / 2, size / 2, 0.3 + 0.9 * k] for k in range(n_inst)], np.float32)
    vel = np.zeros((n_inst, 3), np.float32)
    vel[0] = (0.2, 0.0, 0.1)
    stages = set()
    for cyc in range(14):
        got = _tp_compare_cycle(fl, N, oracles, pos, vel, cyc)
        # scoreTrajectory / checkTrajectory against the grids of this cycle
        for k in range(n_inst):
            for vs in ((0.3, 0.0, 0.2), (0.0, 0.0, -0.8), (-0.1, 0.0, 0.0)):
                a = fl.tp_score_trajectory(k, pos[k].astype(np.float64), vel[k].astype(np.float64), vs)
                b = oracles[k].score_trajectory(pos[k].astype(np.float64), vel[k].astype(np.float64), vs)
                assert (a < 0 and a == b) or (a >= 0 and abs(a - b) <= 1e-9 * max(1.0, abs(b))), (cyc, k, vs, a, b)
        dt = 0.1
        for k in range(n_inst):
            r = got[k]
            stages.add((k, r.xv > 0, r.thetav != 0, r.cost))
            th = float(pos[k, 2])
            pos[k, 0] += (r.drive[0] * np.cos(th) - r.drive[1] * np.sin(th)) * dt
            pos[k, 1] += (r.drive[0] * np.sin(th) + r.drive[1] * np.cos(th)) * dt
            pos[k, 2] += r.drive[2] * dt
            vel[k] = r.drive
        if cyc == 6:  # a new, shorter plan for robot 1 (final-goal speed cap); an empty plan for nobody
            fl.tp_update_plan(1, insts[1]["plan"][:25])
            oracles[1].update_plan(insts[1]["plan"][:25])
    # state equality is asserted every cycle above; the boxed-in robot must have gone through escape mode
    assert any(c == 1.0 for (k, fwd, rot, c) in stages if k == 2), sorted(stages)
    fl.close()


def test_trajectory_planner_edge_cases(nav, orc):
    """Other wavefront kernels (600x600 -> k_bfs<12>, 1000x1000 -> k_bfs_global) with within_robot, a circular
    (2-vertex) footprint, an empty plan, a robot next to the map border, unsupported options."""
    from navigation_amd import synth
    N = L(nav)
    for n, fp in ((600, synth.FOOTPRINT), (1000, synth.FOOTPRINT5), (120, np.array([[0.1, 0.0], [-0.1, 0.0]]))):
        ins = _inflated_instance(orc, n, 70, synth)
        m = ins["master"]
        size = n * synth.RES
        cc = n // 2
        m[cc - 2:cc + 3, cc - 2:cc + 3] = LETHAL  # an obstacle under the robot: within_robot lets the wavefront through
        cfg = N.TpConfig(vx_samples=4, vtheta_samples=5, sim_time=1.0, sim_granularity=0.05, angular_sim_granularity=0.05)
        fl = nav.Fleet(1, n, n, synth.RES, layers=N.LAYER_OBSTACLE, max_sim_steps=64, max_plan=256, max_footprint=16)
        fl.set_footprint(fp)
        fl.upload(N.GRID_MASTER, m)
        fl.configure_trajectory_planner(cfg)
        o = orc.TrajectoryPlanner(m, synth.RES, cfg, fp)
        fl.tp_update_plan(0, ins["plan"])
        o.update_plan(ins["plan"])
        pos = np.array([[size / 2, size / 2, 0.7]], np.float32)
        _tp_compare_cycle(fl, N, [o], pos, np.zeros((1, 3), np.float32))
        if n == 120:
            # empty plan: every cell unreachable; nothing forward is legal, the planner falls through to backing up
            fl.tp_update_plan(0, np.zeros((0, 2)))
            o.update_plan(np.zeros((0, 2)))
            got = _tp_compare_cycle(fl, N, [o], pos, np.zeros((1, 3), np.float32), 1)
            assert got[0].xv == cfg.backup_vel and got[0].best_sample == got[0].n_samples - 1
            # robot at the map border: samples leave the map
            fl.tp_update_plan(0, ins["plan"])
            o.update_plan(ins["plan"])
            edge = np.array([[size - 0.12, size / 2, 0.0]], np.float32)
            _tp_compare_cycle(fl, N, [o], edge, np.array([[0.3, 0.0, 0.0]], np.float32), 2)
            with pytest.raises(nav.NavgpuError):
                fl.configure_trajectory_planner(N.TpConfig(heading_scoring=1))
            # degenerate sample counts: dvx / dvtheta divide by zero exactly as the reference does (inf / nan steps)
            for vxs, vths in ((1, 5), (4, 1), (1, 1), (0, 0)):
                cfg1 = N.TpConfig(vx_samples=vxs, vtheta_samples=vths, sim_time=1.0, sim_granularity=0.05, angular_sim_granularity=0.05)
                fl.configure_trajectory_planner(cfg1)
                o1 = orc.TrajectoryPlanner(m, synth.RES, cfg1, fp)
                o1.update_plan(ins["plan"])
                o1.set_state(fl.tp_state(0, 1)[0])
                _tp_compare_cycle(fl, N, [o1], pos, np.array([[0.1, 0.0, 0.2]], np.float32), 3)
        fl.close()


def test_mapgrid_fleet_soak(nav, orc):
    """Many wavefront workgroups in flight at once (384 per launch, more than the chip has CUs), several cycles with moving robots: every
    grid against the oracle.  Guards the barrier-less neighbour-wave synchronisation of k_bfs_wave under load."""
    from navigation_amd import synth
    N = L(nav)
    n, n_inst, cycles = 400, 128, 5
    cfg = nav.DwaConfig(vx_samples=4, vy_samples=2, vth_samples=4, sim_time=1.0, sim_granularity=0.1, discretize_by_time=1)
    ocfg = orc.DwaConfig(**cfg.as_dict())
    insts = [_inflated_instance(orc, n, 200 + i, synth) for i in range(n_inst)]
    fl = nav.Fleet(n_inst, n, n, synth.RES, layers=N.LAYER_OBSTACLE, max_sim_steps=32, max_plan=256)
    fl.configure_planner(cfg)
    fl.set_footprint(synth.FOOTPRINT)
    fl.upload(N.GRID_MASTER, np.stack([i["master"] for i in insts]))
    fl.set_plan()
    planners = [orc.DwaPlanner(i["master"], synth.RES, 0.0, 0.0, ocfg) for i in insts]
    for p in planners:
        p.set_plan()
    rs = np.random.RandomState(12)
    for cyc in range(cycles):
        pos = np.stack([i["pos"] for i in insts]).astype(np.float32)
        pos[:, :2] += rs.uniform(-0.4, 0.4, (n_inst, 2)).astype(np.float32)
        pos[:, 2] += rs.uniform(-1.0, 1.0, n_inst).astype(np.float32)
        vel = np.stack([i["vel"] for i in insts]).astype(np.float32)
        cut = 200 - 30 * cyc  # shorter plans in later cycles: other seeds, other goals
        plans = np.stack([i["plan"][:cut] for i in insts])
        res = fl.find_best_path(pos, vel, plans)
        for k in range(n_inst):
            ores = planners[k].cycle(pos[k], vel[k], plans[k], synth.FOOTPRINT, want_samples=False)[0]
            for gid, which in ((N.GRID_PATH, 0), (N.GRID_GOAL, 1), (N.GRID_GOAL_FRONT, 2)):
                g = fl.download(gid, k, 1)[0]
                assert np.array_equal(g.astype(np.float64), planners[k].grid(which)), f"MapGrid {which} differs (robot {k}, cycle {cyc})"
            assert res[k].best_index == ores.best_index and abs(res[k].cost - ores.cost) <= 1e-5
    fl.close()


# ---- bounded MapGrid wavefronts (navgpu_planner_set_bounded_map_grids): same planner results, grids completed on demand
def test_bounded_map_grids_equal_complete(nav, orc):
    from navigation_amd import synth
    from navigation_amd._lib import NavgpuError
    N = L(nav)
    n, n_inst = 400, 6
    cfg = nav.DwaConfig(vx_samples=8, vy_samples=5, vth_samples=9, sim_time=1.7, sim_granularity=0.085, discretize_by_time=1)
    insts = [_inflated_instance(orc, n, 40 + i, synth) for i in range(n_inst)]
    masters = np.stack([i["master"] for i in insts])
    pos = np.stack([i["pos"] for i in insts])
    vel = np.stack([i["vel"] for i in insts])
    plans = np.stack([i["plan"] for i in insts])
    out = {}
    for bounded in (1, 0):
        fl = nav.Fleet(n_inst, n, n, synth.RES, layers=N.LAYER_OBSTACLE, keep_sample_costs=True, max_sim_steps=32, max_plan=256)
        fl.configure_planner(cfg)
        fl.set_footprint(synth.FOOTPRINT)
        fl.set_bounded_map_grids(bounded)
        fl.upload(N.GRID_MASTER, masters)
        fl.set_plan()
        res = fl.find_best_path(pos, vel, plans)
        lv = fl.wavefront_levels()
        samples = [fl.samples(k) for k in range(n_inst)]
        grids = [fl.download(g) for g in (N.GRID_PATH, N.GRID_GOAL, N.GRID_GOAL_FRONT)]  # completes the bounded ones
        out[bounded] = (res, lv, samples, grids)
        if bounded:
            # a second cycle leaves bounded grids again; once the costmap they came from is replaced they cannot be completed
            fl.find_best_path(pos, vel, plans)
            fl.upload(N.GRID_MASTER, masters)
            with pytest.raises(NavgpuError):
                fl.download(N.GRID_PATH)
            fl.set_bounded_map_grids(0)
            fl.find_best_path(pos, vel, plans)
            fl.upload(N.GRID_MASTER, masters)
            assert np.array_equal(fl.download(N.GRID_PATH), grids[0])
        fl.close()
    (rb, lb, sb, gb), (rc, lc, sc, gc) = out[1], out[0]
    assert (lb < lc).sum() >= 2 * n_inst, (lb, lc)  # the bounded searches did stop early (goal ~160 cells away, reach ~30)
    assert (lb <= lc).all()
    for k in range(n_inst):
        assert (rb[k].best_index, rb[k].n_valid, rb[k].n_scored, rb[k].cost) == (rc[k].best_index, rc[k].n_valid, rc[k].n_scored, rc[k].cost)
        assert np.array_equal(sb[k][0], sc[k][0]) and np.array_equal(sb[k][1], sc[k][1])  # every sample's cost and status
    for a, b in zip(gb, gc):
        assert np.array_equal(a, b)
    # and both equal the oracle (costs of robot 0)
    p = orc.DwaPlanner(masters[0], synth.RES, 0.0, 0.0, orc.DwaConfig(**cfg.as_dict()))
    p.set_plan()
    o, _, _, cfull, ost = p.cycle(pos[0], vel[0], plans[0], synth.FOOTPRINT)
    scored = ost == 1
    assert np.array_equal(sb[0][1], ost)
    assert np.allclose(sb[0][0][scored], cfull[scored], rtol=0, atol=1e-5)
    assert np.array_equal(gb[0][0].astype(np.float64), p.grid(0))


def test_bounded_map_grids_pockets(nav, orc):
    """Enclosed free cells inside the robot's box: left alone when nothing feeds them, filled when a plan pose lies on
    their wall (seeds expand whatever their cost, map_grid.cpp:160-187) - the bounded search has to wait for that."""
    from navigation_amd import synth
    N = L(nav)
    n = 240
    res = synth.RES
    cfg_kw = dict(vx_samples=9, vy_samples=7, vth_samples=9, sim_time=1.7, sim_granularity=0.085, discretize_by_time=1, max_vel_y=0.3, min_vel_y=-0.3)
    rx, ry = 60, 120  # robot cell
    for variant in range(5):
        m = np.zeros((n, n), np.uint8)
        # ring of lethal cells around a free 5 x 5 interior, 12 cells ahead of the robot; variant 1 puts the plan through it;
        # variants 3 and 4: a larger ring that straddles the rim of the robot's region (box = +-30 cells), off / on the plan
        cx, cy = rx + (12 if variant < 3 else 33), ry + (0 if variant in (1, 4) else 9)
        h = 4 if variant < 3 else 7
        m[cy - h:cy + h + 1, cx - h:cx + h + 1] = LETHAL
        m[cy - h + 2:cy + h - 1, cx - h + 2:cx + h - 1] = 0
        if variant == 2:  # a one-cell pocket next to the robot and a long wall that forces a detour of the goal wavefront
            m[ry - 3:ry, rx + 3:rx + 6] = INSCRIBED
            m[ry - 2, rx + 4] = 0
            m[20:200, 150] = LETHAL
        plan = np.stack([(rx + 0.5) * res + 0.04 * np.arange(200), np.full(200, (ry + 0.5) * res)], 1)
        fl, p = _planner_pair(nav, orc, n, m, cfg_kw, synth.FOOTPRINT)
        fl.set_plan()
        p.set_plan()
        pos, vel = [(rx + 0.5) * res, (ry + 0.5) * res, 0.2], [0.3, 0.0, 0.1]
        _compare_cycle(fl, p, pos, vel, plan, synth.FOOTPRINT)
        lv = fl.wavefront_levels()[0]
        # none of them ran over the whole 240 x 240 map (path > 230 levels, goal > 330, behind the wall of variant 2 > 500)
        # (variants 3, 4: the region is the large area, which the frontier takes longer to leave)
        assert lv[0] < (120 if variant < 3 else 200) and lv[1] < (320 if variant != 2 else 450) and lv[2] < (320 if variant != 2 else 450), (variant, lv)
        for gid, which in ((N.GRID_PATH, 0), (N.GRID_GOAL, 1), (N.GRID_GOAL_FRONT, 2)):
            assert np.array_equal(fl.download(gid, 0, 1)[0].astype(np.float64), p.grid(which)), (variant, which)
        # a checked trajectory far faster than the limits leaves the box: the grids are completed first
        for vs in ([3.0, 0.0, 0.0], [2.0, 0.3, 0.1], [0.2, 0.0, 0.0]):
            assert fl.check_trajectory(0, vs) == p.check_trajectory(np.asarray(pos, np.float32), np.asarray(vel, np.float32), vs), (variant, vs)
        fl.close()


def test_bounded_map_grids_geometry(nav, orc):
    """Bounded searches at the edges of what the pocket mask covers: a reach too long for it (plain region test), a robot in
    the corner of the map (region clipped), a robot off the map (whole-grid search), a rectangular map."""
    from navigation_amd import synth
    N = L(nav)
    res = synth.RES
    rs = np.random.RandomState(11)
    cases = [
        # nx, ny, robot cell, config overrides, expect the searches to be bounded
        (400, 400, (60, 200), dict(sim_time=3.4, sim_granularity=0.17, max_vel_x=0.9, max_trans_vel=0.9), True),   # reach ~ 70 cells: no pocket mask
        (400, 400, (5, 6), dict(), None),                                                                            # corner: region clipped (and the last cells any search reaches)
        (400, 400, (-20, 200), dict(), False),                                                                       # off the map
        (416, 250, (40, 100), dict(), True),                                                                         # rectangular, ragged strips
    ]
    for nx, ny, (rx, ry), over, expect_bounded in cases:
        m = np.zeros((ny, nx), np.uint8)
        for _ in range(nx * ny // 1500):
            cx, cy, r = rs.randint(0, nx), rs.randint(0, ny), rs.randint(1, 4)
            m[max(0, cy - r):cy + r + 1, max(0, cx - r):cx + r + 1] = LETHAL
        m[rs.random_sample(m.shape) < 0.003] = INSCRIBED  # single blocked cells: pockets between them are likely
        m[max(0, ry - 3):ry + 4, max(0, rx - 3):max(0, rx + 4)] = 0
        kw = dict(vx_samples=6, vy_samples=3, vth_samples=7, sim_time=1.7, sim_granularity=0.085, discretize_by_time=1)
        kw.update(over)
        cfg = nav.DwaConfig(**kw)
        fl = nav.Fleet(1, nx, ny, res, layers=N.LAYER_OBSTACLE, keep_sample_costs=True, max_sim_steps=64, max_plan=256)
        fl.configure_planner(cfg)
        fl.set_footprint(synth.FOOTPRINT)
        fl.upload(N.GRID_MASTER, m)
        p = orc.DwaPlanner(m, res, 0.0, 0.0, orc.DwaConfig(**cfg.as_dict()))
        fl.set_plan()
        p.set_plan()
        x0, y0 = (rx + 0.5) * res, (ry + 0.5) * res
        plan = np.stack([np.linspace(max(x0, 0.3), (nx - 12) * res, 200), np.linspace(y0, (ny // 2) * res, 200)], 1)
        pos, vel = [x0, y0, 0.1], [0.2, 0.0, 0.0]
        _compare_cycle(fl, p, pos, vel, plan, synth.FOOTPRINT)
        lv = fl.wavefront_levels()[0].astype(int)
        for gid, which in ((N.GRID_PATH, 0), (N.GRID_GOAL, 1), (N.GRID_GOAL_FRONT, 2)):
            og = p.grid(which)
            assert np.array_equal(fl.download(gid, 0, 1)[0].astype(np.float64), og), (nx, ny, rx, ry, which)
        full = [int(p.grid(w)[p.grid(w) < nx * ny].max()) for w in (0, 1, 2)]
        if expect_bounded:
            assert lv[1] < full[1] and lv[2] < full[2], (nx, ny, rx, ry, lv, full)
        elif expect_bounded is None:
            assert lv[0] < full[0] and all(lv[w] <= full[w] + 8 for w in range(3)), (lv, full)
        else:
            assert all(lv[w] >= full[w] for w in range(3)), (lv, full)
        fl.close()


def test_bounded_map_grids_reconfigure_between_stage_and_cycle(nav, orc):
    """DWAPlanner::reconfigure may come between staging and the cycle: the box of the bounded searches follows the new limits."""
    from navigation_amd import synth
    N = L(nav)
    n = 400
    ins = _inflated_instance(orc, n, 7, synth)
    slow = dict(vx_samples=7, vy_samples=3, vth_samples=7, sim_time=1.0, sim_granularity=0.1, discretize_by_time=1, max_vel_x=0.2, max_trans_vel=0.2)
    fast = dict(slow, sim_time=2.5, sim_granularity=0.125, max_vel_x=1.2, max_trans_vel=1.2, acc_lim_x=30.0)
    fl = nav.Fleet(1, n, n, synth.RES, layers=N.LAYER_OBSTACLE, keep_sample_costs=True, max_sim_steps=64, max_plan=256)
    fl.configure_planner(nav.DwaConfig(**slow))
    fl.set_footprint(synth.FOOTPRINT)
    fl.upload(N.GRID_MASTER, ins["master"])
    fl.set_plan()
    pos, vel = ins["pos"].copy(), np.array([0.2, 0.0, 0.0], np.float32)
    fl.stage_planner([pos], [vel], [ins["plan"]])
    cfg = nav.DwaConfig(**fast)
    fl.configure_planner(cfg)   # after the stage: reach 6 cells -> 66 cells
    fl.planner_cycle()
    r = fl.results()[0]
    cost, status, _ = fl.samples(0)
    p = orc.DwaPlanner(ins["master"], synth.RES, 0.0, 0.0, orc.DwaConfig(**cfg.as_dict()))
    p.set_plan()
    o, _, _, cfull, ost = p.cycle(pos, vel, ins["plan"], synth.FOOTPRINT)
    assert np.array_equal(status, ost)
    sc = status == 1
    assert np.array_equal(cost[sc] < 0, cfull[sc] < 0) and np.array_equal(cost[sc][cost[sc] < 0], cfull[sc][cfull[sc] < 0])
    assert np.allclose(cost[sc][cost[sc] >= 0], cfull[sc][cfull[sc] >= 0], rtol=0, atol=1e-5)
    assert (r.best_index, r.n_valid) == (o.best_index, o.n_valid)
    fl.close()


@pytest.mark.parametrize("sizes,rounds", [((160, 250, 400), 40), ((600,), 3), ((800,), 3)])  # k_bfs_wave<7>, <13>, k_bfs_global
def test_bounded_map_grids_random_stress(nav, orc, sizes, rounds):
    """Seeded random clutter, poses, velocities, limits and plans (some through obstacles, some ending near the robot): the
    bounded and the whole-grid searches must give every sample the same cost, and the completed grids must be equal."""
    from navigation_amd import synth
    N = L(nav)
    res = synth.RES
    rs = np.random.RandomState(2024)
    n_inst = 8
    shorter = total = 0
    for rnd in range(rounds):
        n = int(rs.choice(sizes))
        masters = np.zeros((n_inst, n, n), np.uint8)
        pos, vel, plans = [], [], []
        for k in range(n_inst):
            m = masters[k]
            for _ in range(rs.randint(5, 60)):   # blobs and thin walls, inscribed halo around some
                cx, cy = rs.randint(0, n, 2)
                if rs.rand() < 0.5:
                    r = rs.randint(1, 6)
                    m[max(0, cy - r - 1):cy + r + 2, max(0, cx - r - 1):cx + r + 2] = np.maximum(m[max(0, cy - r - 1):cy + r + 2, max(0, cx - r - 1):cx + r + 2], INSCRIBED)
                    m[max(0, cy - r):cy + r + 1, max(0, cx - r):cx + r + 1] = LETHAL
                else:
                    ln = rs.randint(5, 60)
                    if rs.rand() < 0.5:
                        m[cy, cx:cx + ln] = LETHAL
                    else:
                        m[cy:cy + ln, cx] = LETHAL
            m[rs.random_sample(m.shape) < rs.choice([0.0, 0.002, 0.02])] = INSCRIBED
            m[(rs.random_sample(m.shape) < rs.choice([0.0, 0.01])) & (m == 0)] = NOINFO
            rx, ry = rs.randint(3, n - 3, 2)
            m[ry - 2:ry + 3, rx - 2:rx + 3] = 0
            pos.append([(rx + rs.rand()) * res, (ry + rs.rand()) * res, rs.uniform(-3.1, 3.1)])
            vel.append([rs.uniform(-0.1, 0.6), rs.uniform(-0.1, 0.1), rs.uniform(-1, 1)])
            gx, gy = rs.uniform(0.05, 0.95, 2) * n * res
            if rs.rand() < 0.15:  # a goal close to the robot: whole-grid search by rule
                gx, gy = pos[-1][0] + rs.uniform(-1, 1), pos[-1][1] + rs.uniform(-1, 1)
            t = np.linspace(0, 1, 120)[:, None]
            plans.append(np.clip(np.array(pos[-1][:2]) * (1 - t) + np.array([gx, gy]) * t + 0.3 * np.sin(6 * t) * rs.uniform(-1, 1, 2), 0.01, n * res - 0.01))
        cfg = nav.DwaConfig(vx_samples=6, vy_samples=3, vth_samples=7, sim_time=float(rs.choice([1.0, 1.7, 2.5])), sim_granularity=0.1,
                            discretize_by_time=int(rs.rand() < 0.7), allow_unknown=int(rs.rand() < 0.5), max_vel_x=float(rs.choice([0.3, 0.55, 0.9])),
                            max_trans_vel=float(rs.choice([0.55, 0.9])), forward_point_distance=float(rs.choice([0.0, 0.325, 0.6])),
                            max_vel_y=0.2, min_vel_y=-0.2, use_dwa=int(rs.rand() < 0.8))
        fl = nav.Fleet(n_inst, n, n, res, layers=N.LAYER_OBSTACLE, keep_sample_costs=True, max_sim_steps=128, max_plan=256)
        fl.configure_planner(cfg)
        fl.set_footprint(synth.FOOTPRINT)
        fl.upload(N.GRID_MASTER, masters)
        fl.set_plan()
        out = {}
        for bounded in (1, 0):
            fl.set_bounded_map_grids(bounded)
            res_b = fl.find_best_path(np.array(pos), np.array(vel), np.stack(plans))
            lv = fl.wavefront_levels()
            samples = [fl.samples(k) for k in range(n_inst)]
            grids = [fl.download(g) for g in (N.GRID_PATH, N.GRID_GOAL, N.GRID_GOAL_FRONT)]
            out[bounded] = (res_b, samples, grids, lv)
        for k in range(n_inst):
            a, b = out[1][0][k], out[0][0][k]
            assert (a.best_index, a.n_valid, a.n_scored, a.cost) == (b.best_index, b.n_valid, b.n_scored, b.cost), (rnd, k)
            assert np.array_equal(out[1][1][k][1], out[0][1][k][1]) and np.array_equal(out[1][1][k][0], out[0][1][k][0]), (rnd, k)
        for ga, gb in zip(out[1][2], out[0][2]):
            assert np.array_equal(ga, gb), rnd
        assert (out[1][3] <= out[0][3] + 8).all(), rnd
        shorter += int((out[1][3] + 8 < out[0][3]).sum())
        total += out[1][3].size
        fl.close()
    assert shorter > total // 3, (shorter, total)  # the comparison is not vacuous: many of the searches did stop early
