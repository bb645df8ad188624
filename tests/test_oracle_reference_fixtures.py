"""Pins the CPU oracle (oracle/) against every golden vector / known-answer test the reference's
own test-suite holds for the hot path.  Each test names the reference test it
restates.  No GPU needed."""
import math

import numpy as np
import pytest

LETHAL, INSCRIBED, NOINFO, FREE = 254, 253, 255, 0
MAX_Z = 1.0


def count(m, v, equal=True):
    return int((m == v).sum()) if equal else int((m != v).sum())


# ------------------------------------------------------------------ oracle/_ref: reference headers
def test_ref_line_iterator_matches_oracle(orc):
    R = orc.ref()
    if R is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.default_rng(7)
    for _ in range(3000):
        x0, y0, x1, y1 = [int(v) for v in rng.integers(-40, 40, 4)]
        cap = max(abs(x1 - x0), abs(y1 - y0)) + 2
        out = np.zeros((cap, 2), np.int32)
        n = R.ref_line_cells(x0, y0, x1, y1, out, cap)
        mine = orc.line_cells(x0, y0, x1, y1)
        assert n == len(mine)
        assert np.array_equal(out[:n], mine)


def test_ref_velocity_iterator_matches_oracle(orc):
    R = orc.ref()
    if R is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.default_rng(11)
    cases = [(0.0, 0.0, 1), (-30, 30, 4), (-10.00001, 10, 3), (-0.1, 0.1, 10), (0.0, 0.55, 32)]
    for _ in range(2000):
        a, b = sorted(rng.uniform(-2, 2, 2))
        cases.append((float(a), float(b), int(rng.integers(0, 40))))
    for mn, mx, n in cases:
        out = np.zeros(n + 8, np.float64)
        k = R.ref_velocity_samples(mn, mx, n, out, len(out))
        mine = orc.velocity_samples(mn, mx, n)
        assert k == len(mine)
        assert np.array_equal(out[:k], mine)  # bit-exact doubles


def test_ref_cost_values(orc):
    R = orc.ref()
    if R is None:
        pytest.skip("oracle/_ref not built")
    v = np.zeros(4, np.uint8)
    R.ref_cost_values(v)
    assert list(v) == [NOINFO, LETHAL, INSCRIBED, FREE]


# ------------------------------------------------------------------ base_local_planner/test/line_iterator_test.cpp:34-76
def test_line_iterator_south(orc):
    assert orc.line_cells(1, 2, 1, 4).tolist() == [[1, 2], [1, 3], [1, 4]]


def test_line_iterator_north_north_west(orc):
    assert orc.line_cells(0, 0, -2, -4).tolist() == [[0, 0], [-1, -1], [-1, -2], [-2, -3], [-2, -4]]


# ------------------------------------------------------------------ base_local_planner/test/velocity_iterator_test.cpp:45-176
@pytest.mark.parametrize("mn,mx,n,expected", [
    (0.0, 0.0, 1, [0.0]), (2.2, 2.2, 1, [2.2]), (-3.3, -3.3, 1, [-3.3]),
    (-30, 30, 1, [-30.0, 0.0, 30.0]), (10, 30, 1, [10.0, 30.0]), (-30, -10, 1, [-30.0, -10.0]),
    (-30, 30, 3, [-30.0, 0.0, 30.0]), (-30, 30, 4, [-30.0, -10.0, 0.0, 10.0, 30.0]),
    (-10, 50, 4, [-10.0, 0.0, 10.0, 30.0, 50.0])])
def test_velocity_iterator(orc, mn, mx, n, expected):
    assert orc.velocity_samples(mn, mx, n).tolist() == expected


def test_velocity_iterator_cranky(orc):
    got = orc.velocity_samples(-10.00001, 10, 3)
    assert len(got) == 4
    assert np.allclose(got, [-10.00001, -0.000005, 0.0, 10.0], rtol=1e-6, atol=1e-12)


# ------------------------------------------------------------------ base_local_planner/test/map_grid_test.cpp:113-160
def test_map_grid_adjust_plan(orc):
    assert len(orc.adjust_plan(np.zeros((0, 2)), 0.0)) == 0
    out = orc.adjust_plan([[1, 1], [5, 5]], 1.0)
    assert out.tolist() == [[1, 1], [3, 3], [5, 5]]


def test_map_grid_distance_propagation(orc):
    g = np.zeros((10, 10), np.uint8)
    d = orc.map_grid_seeded(g, [], True)
    assert (d == 101).all()  # nothing marked, everything unreachable
    d = orc.map_grid_seeded(g, [0], True)
    assert d[0, 0] == 0 and d[1, 1] == 2 and d[4, 0] == 4 and d[0, 4] == 4 and d[9, 9] == 18


# ------------------------------------------------------------------ base_local_planner/test/utest.cpp:104-166
def test_trajectory_planner_goal_distance(orc):
    g = np.zeros((10, 10), np.uint8)  # g[y, x]
    g[6, 4] = LETHAL  # footprintObstacles() put it there
    for (x, y) in [(1, 2), (1, 1), (1, 0), (2, 0), (3, 0), (3, 1), (3, 2), (2, 2)]:
        g[y, x] = LETHAL
    # wall from the footprintObstacles test (path_map_(7, y).target_dist = 1 then synchronize)
    for y in (1, 3, 4, 5, 6, 7):
        g[y, 7] = LETHAL
    d = orc.map_grid_seeded(g, [9 * 10 + 4], True)
    at = lambda x, y: d[y, x]
    assert at(4, 8) == 1 and at(4, 7) == 2 and at(4, 6) == 100
    assert at(4, 5) == 6 and at(4, 4) == 7 and at(4, 3) == 8 and at(4, 2) == 9 and at(4, 1) == 10 and at(4, 0) == 11
    assert at(5, 8) == 2 and at(9, 4) == 10
    assert at(2, 2) == 100  # the boxed-in wall cell itself is an obstacle cell
    assert at(2, 1) == 101  # and the cell it boxes in is never reached


# ------------------------------------------------------------------ base_local_planner/test/footprint_helper_test.cpp:52-130
def test_footprint_outline_cells_square(orc):
    # 4x4 square footprint at (4.5, 4.5): the outline cells the reference expects, as a set, via
    # the CostmapModel edge walk (LineIterator over worldToMap'ed rotated vertices)
    expected = {(6, y) for y in range(2, 7)} | {(x, 2) for x in range(2, 7)} | {(2, y) for y in range(2, 7)} | \
               {(x, 6) for x in range(2, 7)}
    fp = np.array([[2, 2], [2, -2], [-2, -2], [-2, 2]], float)
    for th in (0.0, math.pi / 2):
        # probe which cells the edge walk reads: a lethal cell makes footprintCost -1 iff it is on the outline
        hit = set()
        for y in range(10):
            for x in range(10):
                g = np.zeros((10, 10), np.uint8)
                g[y, x] = LETHAL
                if orc.footprint_cost(g, 1.0, 0.0, 0.0, 4.5, 4.5, th, fp) < 0:
                    hit.add((x, y))
        assert hit == expected


# ------------------------------------------------------------------ voxel_grid/test/voxel_grid_tests.cpp:40-140
def test_voxel_grid_basic_marking_and_clearing(orc):
    L = orc.lib()
    sx, sy, sz = 50, 10, 16
    vg = L.orc_vg_create(sx, sy, sz)
    tz, xmin, xmax, ymin, ymax = 12, 5, 15, 0, 3
    for x in range(xmin, xmax + 1):
        L.orc_vg_mark_line(vg, x, ymin, tz, x, ymax, tz)
    for i in range(xmin, xmax + 1):
        for j in range(ymin, ymax + 1):
            assert L.orc_vg_get_voxel(vg, i, j, tz) == 2

    def census():
        c = [0, 0, 0]
        for i in range(sx):
            for j in range(sy):
                for k in range(sz):
                    c[L.orc_vg_get_voxel(vg, i, j, k)] += 1
        return c
    free, unknown, marked = census()
    assert marked == 44 and unknown == sx * sy * sz - 44
    L.orc_vg_clear_line(vg, xmin, ymin, tz, xmax, ymin, tz)
    free, unknown, marked = census()
    assert marked == 33 and free == 11 and unknown == sx * sy * sz - 44
    for k in range(sz):
        L.orc_vg_mark_voxel(vg, 0, 0, k)
        assert L.orc_vg_get_voxel(vg, 0, 0, k) == 2
    L.orc_vg_clear_line(vg, 0, 0, 0, 0, 0, sz - 1)
    for k in range(sz):
        assert L.orc_vg_get_voxel(vg, 0, 0, k) == 0
    L.orc_vg_destroy(vg)


# ------------------------------------------------------------------ costmap_2d/test/obstacle_tests.cpp
def _static_obstacle(orc, ten_by_ten, track_unknown=False):
    lc = orc.LayeredCostmap(track_unknown)
    lc.add_static(ten_by_ten)
    lc.add_obstacle()
    return lc


def test_obstacle_raytracing(orc, ten_by_ten):  # :74-92
    lc = _static_obstacle(orc, ten_by_ten)
    lc.add_observation([[0.0, 0.0, MAX_Z / 2]], origin=(0, 0, MAX_Z / 2))
    lc.update_map(0, 0, 0)
    assert count(lc.master(), LETHAL) == 21


def test_obstacle_raytracing2(orc, ten_by_ten):  # :97-148
    lc = _static_obstacle(orc, ten_by_ten)
    lc.update_map(0, 0, 0)
    before = count(lc.master(), LETHAL)
    assert before == 20
    lc.add_observation([[9.5, 9.5, MAX_Z / 2]], origin=(0.5, 0.5, MAX_Z / 2))
    lc.update_map(0, 0, 0)
    after = count(lc.master(), LETHAL)
    assert after == before + 1
    layer = lc.layer(2)
    for i in range(10):
        layer[i, i] = LETHAL
    lc.set_layer(layer, 2)
    lc.update_map(0, 0, 0)
    m = lc.master()
    assert count(m, LETHAL) == after
    assert count(m, FREE) == 79


def test_obstacle_wave_interference(orc):  # :153-177
    lc = orc.LayeredCostmap(True)
    lc.resize(10, 10, 1, 0, 0)
    lc.add_obstacle()
    for p in (3.0, 5.0, 7.0):
        lc.add_observation([[p, p, MAX_Z]])
    lc.update_map(0, 0, 0)
    m = lc.master()
    assert count(m, LETHAL) == 3 and count(m, NOINFO) == 92 and count(m, FREE) == 5


def test_obstacle_z_threshold(orc):  # :182-198
    lc = orc.LayeredCostmap(True)
    lc.resize(10, 10, 1, 0, 0)
    lc.add_obstacle()
    lc.add_observation([[0.0, 5.0, 0.4]])
    lc.add_observation([[1.0, 5.0, 2.2]])
    lc.update_map(0, 0, 0)
    assert count(lc.master(), LETHAL) == 1


def test_obstacle_dynamic_obstacles(orc, ten_by_ten):  # :204-226
    lc = _static_obstacle(orc, ten_by_ten)
    for _ in range(3):
        lc.add_observation([[0.0, 0.0, 0.0]])
    lc.update_map(0, 0, 0)
    assert count(lc.master(), LETHAL) == 21


def test_obstacle_multiple_additions(orc, ten_by_ten):  # :232-245
    lc = _static_obstacle(orc, ten_by_ten)
    lc.add_observation([[9.5, 0.0, 0.0]])
    lc.update_map(0, 0, 0)
    assert count(lc.master(), LETHAL) == 20


# ------------------------------------------------------------------ costmap_2d/test/inflation_tests.cpp (cost_scaling_factor = 1)
def _radii(length, width):
    return [[width, length], [width, -length], [-width, -length], [-width, length]]


def _inflation_map(orc, polygon, radius, static=None):
    lc = orc.LayeredCostmap(False)
    if static is None:
        lc.resize(10, 10, 1, 0, 0)
    lc.set_footprint(polygon)
    if static is not None:
        lc.add_static(static)
    lc.add_obstacle()
    lc.add_inflation(radius, 1.0)
    lc.set_footprint(polygon)
    return lc


def test_inflation_adjacent_to_obstacle_can_still_move(orc):  # :130-154
    lc = _inflation_map(orc, _radii(2.1, 2.3), 4.1)
    lc.add_observation([[0, 0, MAX_Z]])
    lc.update_map(0, 0, 0)
    m = lc.master()
    c = lambda x, y: int(m[y, x])
    assert c(0, 0) == LETHAL and c(1, 0) == INSCRIBED and c(2, 0) == INSCRIBED
    assert c(3, 0) < INSCRIBED and c(2, 1) < INSCRIBED and c(1, 1) == INSCRIBED
    assert count(m, NOINFO) == 0  # testInflationShouldNotCreateUnknowns :156-175


def test_inflation_cost_function_correctness(orc):  # :181-222
    lc = orc.LayeredCostmap(False)
    lc.resize(100, 100, 1, 0, 0)
    poly = _radii(5.0, 6.25)
    lc.set_footprint(poly)
    lc.add_obstacle()
    lc.add_inflation(10.5, 1.0)
    lc.set_footprint(poly)
    lc.add_observation([[50, 50, MAX_Z]])
    lc.update_map(0, 0, 0)
    m = lc.master()
    for i in range(0, 6):
        assert m[50, 50 + i] >= INSCRIBED and m[50, 50 - i] >= INSCRIBED
        assert m[50 + i, 50] >= INSCRIBED and m[50 - i, 50] >= INSCRIBED
    for i in range(6, 12):
        expected = orc.lib().orc_compute_cost(1.0, 1.0, lc.inscribed_radius, float(i))
        assert m[50, 50 + i] == expected
    assert lc.inscribed_radius == 5.0


def test_inflation_priority_queue_use_correctness(orc):  # :229-272 (validatePointInflation: cost >= expected)
    lc = _inflation_map(orc, _radii(2.1, 2.3), 4.1)
    lc.add_observation([[4, 4, MAX_Z]])
    lc.add_observation([[5, 5, MAX_Z]])
    lc.update_map(0, 0, 0)
    m = lc.master()
    for (sx, sy) in ((4, 4), (5, 5)):
        for y in range(10):
            for x in range(10):
                d = math.hypot(x - sx, y - sy)
                if d <= 4.1 + 1:  # cells the validation walk visits
                    assert m[y, x] >= orc.lib().orc_compute_cost(1.0, 1.0, lc.inscribed_radius, d) or d > 5


def test_inflation_static_and_dynamic(orc, ten_by_ten):  # testInflation :277-336
    lc = _inflation_map(orc, _radii(1, 1), 1, static=ten_by_ten)
    lc.update_map(0, 0, 0)
    m = lc.master()
    assert count(m, LETHAL) == 20 and count(m, INSCRIBED) == 28
    lc.add_observation([[0, 0, 0.4]])
    lc.update_map(0, 0, 0)
    m = lc.master()
    assert count(m, LETHAL) + count(m, INSCRIBED) == 51
    lc.add_observation([[2, 0, 0.0]])
    lc.update_map(0, 0, 0)
    m = lc.master()
    assert count(m, LETHAL) + count(m, INSCRIBED) == 54
    lc.add_observation([[1, 9, 0.0]])
    lc.update_map(0, 0, 0)
    m = lc.master()
    assert m[9, 1] == LETHAL and m[9, 0] == INSCRIBED and m[9, 2] == INSCRIBED
    lc.add_observation([[0, 9, 0.0]])
    lc.update_map(0, 0, 0)
    assert lc.master()[9, 0] == LETHAL


def test_inflation2_l_shape(orc, ten_by_ten):  # :341-365
    lc = _inflation_map(orc, _radii(1, 1), 1, static=ten_by_ten)
    for p in ((1, 1), (2, 1), (2, 2)):
        lc.add_observation([[p[0], p[1], MAX_Z]])
    lc.update_map(0, 0, 0)
    m = lc.master()
    assert m[3, 2] == INSCRIBED and m[3, 3] == INSCRIBED


def test_inflation3_empty_map(orc):  # :370-403
    lc = _inflation_map(orc, _radii(1, 1.75), 3)
    m = lc.master()
    assert count(m, LETHAL) == 0 and count(m, INSCRIBED) == 0
    lc.add_observation([[5, 5, MAX_Z]])
    for _ in range(2):
        lc.update_map(0, 0, 0)
        m = lc.master()
        assert count(m, FREE, False) == 29 and count(m, LETHAL) == 1 and count(m, INSCRIBED) == 4


def test_inflation_exact_equals_pq_on_reference_fixtures(orc, ten_by_ten):
    """On every (tie-free) fixture of inflation_tests.cpp the order-independent exact-EDT
    specification and the reference's priority-queue walk give identical bytes."""
    scenarios = []
    for exact in (False, True):
        outs = []
        lc = _inflation_map(orc, _radii(1, 1), 1, static=ten_by_ten)
        lc.L.orc_lc_set_inflation_exact(lc.h, int(exact))
        lc.update_map(0, 0, 0)
        outs.append(lc.master())
        for p in ([0, 0, 0.4], [2, 0, 0.0], [1, 9, 0.0], [0, 9, 0.0]):
            lc.add_observation([p])
            lc.update_map(0, 0, 0)
            outs.append(lc.master())
        lc2 = _inflation_map(orc, _radii(2.1, 2.3), 4.1)
        lc2.L.orc_lc_set_inflation_exact(lc2.h, int(exact))
        lc2.add_observation([[4, 4, MAX_Z]])
        lc2.add_observation([[5, 5, MAX_Z]])
        lc2.update_map(0, 0, 0)
        outs.append(lc2.master())
        scenarios.append(outs)
    for a, b in zip(*scenarios):
        assert np.array_equal(a, b)


# ------------------------------------------------------------------ base_local_planner/test/footprint_helper_test.cpp:44-132
def test_footprint_helper_cells_in_reference_order(orc):
    fp = [[2, 2], [2, -2], [-2, -2], [-2, 2]]
    c = orc.footprint_cells(10, 10, 1.0, [4.5, 4.5, 0.0], fp, fill=False)
    assert len(c) == 20
    assert c[0:5] == [(6, 6), (6, 5), (6, 4), (6, 3), (6, 2)]      # front line first
    assert c[5:10] == [(6, 2), (5, 2), (4, 2), (3, 2), (2, 2)]     # right line
    assert c[10:15] == [(2, 2), (2, 3), (2, 4), (2, 5), (2, 6)]    # back line
    assert c[15:20] == [(2, 6), (3, 6), (4, 6), (5, 6), (6, 6)]    # left line
    c = orc.footprint_cells(10, 10, 1.0, [4.5, 4.5, math.pi / 2], fp, fill=False)
    assert c[0:5] == [(2, 6), (3, 6), (4, 6), (5, 6), (6, 6)]
    assert c[5:10] == [(6, 6), (6, 5), (6, 4), (6, 3), (6, 2)]
    assert c[10:15] == [(6, 2), (5, 2), (4, 2), (3, 2), (2, 2)]
    assert c[15:20] == [(2, 2), (2, 3), (2, 4), (2, 5), (2, 6)]
    # fill (used by TrajectoryPlanner::findBestPath for within_robot): outline + interior, every cell of the square covered
    filled = set(orc.footprint_cells(10, 10, 1.0, [4.5, 4.5, 0.0], fp, fill=True))
    assert filled == {(x, y) for x in range(2, 7) for y in range(2, 7)}


# ------------------------------------------------------------------ base_local_planner/test/utest.cpp:56-102
def test_trajectory_planner_footprint_obstacles(orc):
    """TrajectoryPlannerTest::footprintObstacles: tc(cm, map, footprint, acc 0/1/1, sim_time 1, sim_granularity 1, vx_samples 2)."""
    from navigation_amd import _lib as N  # POD layout of the configuration only
    cfg = N.TpConfig(acc_lim_x=0.0, acc_lim_y=1.0, acc_lim_theta=1.0, sim_time=1.0, sim_granularity=1.0, vx_samples=2)
    fp = [[2, 2], [2, -2], [-2, -2], [-2, 2]]
    g = np.zeros((10, 10), np.uint8)
    g[6, 4] = LETHAL  # map_(4, 6).target_dist = 1; wa->synchronize()
    tp = orc.TrajectoryPlanner(g, 1.0, cfg, fp)
    # generateTrajectory(4.5, 4.5, M_PI_2, 0, 0, 0, 4, 0, 0, 4, 0, 0, DBL_MAX, traj): drives into the obstacle
    assert tp.generate([4.5, 4.5, math.pi / 2], [0, 0, 0], [4, 0, 0], [4, 0, 0], 1.7976931348623157e308) == -1.0
    for y in (1, 3, 4, 5, 6, 7):  # the wall next to the footprint
        g[y, 7] = LETHAL
    tp.set_costmap(g)
    # generateTrajectory(4.5, 4.5, M_PI_2, 0, 0, 0, 0, 0, M_PI_2, 0, 0, M_PI_4, 100, traj): rotates into the wall
    assert tp.generate([4.5, 4.5, math.pi / 2], [0, 0, 0], [0, 0, math.pi / 2], [0, 0, math.pi / 4], 100) == -1.0
    # and without the obstacles the same two commands are legal (the -1 above comes from the footprint test)
    tp.set_costmap(np.zeros((10, 10), np.uint8))
    tp.update_plan([[4.5, 8.5]], compute_dists=True)
    assert tp.generate([4.5, 4.5, math.pi / 2], [0, 0, 0], [0, 0, math.pi / 2], [0, 0, math.pi / 4], 100) >= 0


def test_oracle_rolling_static_layer_identity_equals_shifted_copy(orc):
    """StaticLayer's rolling branch (static_layer.cpp:300-333) with the identity transform and a static map of the
    costmap's own resolution: the window is the static map's cells under it, the rest stays at the default."""
    occ = np.zeros((60, 80), np.int8)
    occ[10:20, 30:50] = 100
    occ[40, :] = -1
    o = orc.LayeredCostmap(True)
    o.resize(40, 40, 0.5, 0, 0)
    o.set_rolling(True)
    o.add_static_rolling(occ, 0.5, -5.0, -5.0)
    o.update_map(4.0, 3.0, 0.0)  # sizeInMeters = 19.75: the origin snaps to (-5.5, -6.5); cells left of / below the static map keep 255
    m = o.master()
    ox, oy = o.origin()
    assert (ox, oy) == (-5.5, -6.5)
    want = np.full((40, 40), 255, np.uint8)
    interp = np.where(occ == 100, 254, np.where(occ == -1, 255, 0)).astype(np.uint8)
    want[3:, 1:] = interp[:37, :39]  # master cell (i, j) <-> static cell (i - 1, j - 3)
    assert np.array_equal(m, want)
    assert np.array_equal(o.bounds(), [1, 40, 3, 40])  # the static map's extent (from its cell (0, 0) centre), clipped to the window, every cycle


def test_oracle_rollout_trig_switch_is_live(orc):
    """navgpu_dwa_config::rollout_trig (the float overload of computeNewPositions' cos / sin): the float product differs from the double
    one by ~1e-9 m per step, which moves the float-rounded pose in some steps of some samples - by one float ulp, never more per step."""
    rs = np.random.RandomState(3)
    n_diff = n_steps = 0
    worst = 0.0
    for _ in range(400):
        pos = np.array([rs.uniform(2, 8), rs.uniform(2, 8), rs.uniform(-3.1, 3.1)], np.float32)
        vel = np.array([0.2, 0.0, 0.1], np.float32)
        sample = np.array([rs.uniform(0.1, 0.5), rs.uniform(-0.1, 0.1), rs.uniform(-1, 1)], np.float32)
        tr = []
        for trig in (0, 1):
            cfg = orc.DwaConfig(sim_time=2.0, sim_granularity=0.1, discretize_by_time=1, rollout_trig=trig)
            k, pts, _ = orc.generate_trajectory(cfg, pos, vel, sample)
            assert k == 20
            tr.append(pts)
        assert np.array_equal(tr[0][:, 2], tr[1][:, 2])  # the heading recurrence has no trigonometry in it
        d = np.abs(tr[0][:, :2] - tr[1][:, :2])
        n_diff += int((d > 0).any(axis=1).sum())
        n_steps += k
        worst = max(worst, float(d.max()))
    assert 0 < n_diff < 0.5 * n_steps and worst <= 20 * 2.0 ** -21  # (a float ulp at 8 m is 2^-20 m... at most one per step, accumulated)
