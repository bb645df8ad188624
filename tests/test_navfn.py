"""navfn::NavFn: the oracle against the reference's own test (CPU), and the HIP path against the oracle
bit for bit (GPU): potentials, cycle counts, path points."""
import gzip
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def willow():
    """navfn/test/willow_costmap.pgm (the reference's own test data, gzip-compressed): P5, 1132 x 1217."""
    raw = gzip.open(os.path.join(ROOT, "tests", "golden", "willow_costmap.pgm.gz")).read()
    magic, w, h, maxval, data = raw.split(b"\n", 4)
    assert magic == b"P5" and maxval == b"255"
    nx, ny = int(w), int(h)
    return np.frombuffer(data, np.uint8)[:nx * ny].reshape(ny, nx).copy()


# navfn/test/path_calc_test.cpp:111-154: readPGM(raw = true) bytes copied straight into costarr, priInc = 2 * COST_NEUTRAL
WILLOW_CASES = [((428, 746), (350, 450)), ((350, 400), (350, 450))]  # (start, goal)


@pytest.mark.parametrize("start,goal", WILLOW_CASES)
def test_oracle_reference_path_calc(orc, willow, start, goal):
    """TEST(PathCalc, oscillate_in_pinch_point) / (easy_nav_should_always_work): calcNavFnDijkstra(true) finds a plan."""
    path, pot, cyc = orc.navfn_plan(willow, goal, start, cost_mode=0)
    assert len(path) > 0
    assert tuple(path[0]) == tuple(float(v) for v in start) and tuple(path[-1]) == tuple(float(v) for v in goal)
    step = np.hypot(*np.diff(path, axis=0).T)
    assert step.max() < 1.6  # half-cell steps, or a grid step where the potential has a boundary


def test_oracle_navfn_basic_properties(orc):
    """Open field: the potential grows away from the goal by about COST_NEUTRAL per cell, A* finds the same straight path."""
    n = 64
    cm = np.zeros((n, n), np.uint8)
    path, pot, cyc = orc.navfn_plan(cm, (10, 10), (50, 40), cost_mode=1, at_start=False)
    assert len(path) > 0 and pot[10, 10] == 0
    assert abs(pot[10, 30] - 20 * 50) < 1e-3  # along an axis the update is ta + hf exactly
    assert 1e9 < pot[0, 0]  # the border is an obstacle: never assigned
    pa, _, _ = orc.navfn_plan(cm, (10, 10), (50, 40), cost_mode=1, astar=True)
    assert len(pa) > 0
    # a wall with no gap: no plan
    cm[:, 32] = 254
    path, _, _ = orc.navfn_plan(cm, (10, 10), (50, 40), cost_mode=1)
    assert len(path) == 0


def _random_costmap(rs, n, density):
    cm = np.zeros((n, n), np.uint8)
    cm[rs.random_sample((n, n)) < density] = 254
    blur = (rs.random_sample((n, n)) < 0.15) & (cm == 0)
    cm[blur] = rs.randint(1, 253, blur.sum())
    cm[(rs.random_sample((n, n)) < 0.01) & (cm == 0)] = 255
    return cm


@pytest.mark.gpu
@pytest.mark.parametrize("astar", [False, True])
def test_navfn_batch_matches_oracle(orc, astar):
    import navigation_amd as nav
    nav.lib()
    rs = np.random.RandomState(11)
    n, nI = 120, 6
    maps = np.stack([_random_costmap(rs, n, 0.04) for _ in range(nI)])
    goals = rs.randint(8, n - 8, (nI, 2))
    starts = rs.randint(8, n - 8, (nI, 2))
    for k in range(nI):  # keep both ends on free cells
        maps[k][goals[k][1], goals[k][0]] = 0
        maps[k][starts[k][1], starts[k][0]] = 0
    nf = nav.NavFn(n, n, nI)
    for allow_unknown in (True, False):
        nf.set_costmap(maps, cost_mode=1, allow_unknown=allow_unknown)
        res = nf.plan(goals, starts, astar=astar)
        n_found = 0
        for k in range(nI):
            path, pot, cyc = orc.navfn_plan(maps[k], goals[k], starts[k], cost_mode=1, allow_unknown=allow_unknown, astar=astar)
            assert res[k].cycles == cyc, (k, res[k].cycles, cyc)
            assert np.array_equal(nf.potential(k).view(np.uint32), pot.view(np.uint32)), f"plan {k}: potential arrays differ"
            assert res[k].path_length == len(path) and bool(res[k].found) == (len(path) > 0)
            assert np.array_equal(nf.path(k).view(np.uint32), path.view(np.uint32)), f"plan {k}: paths differ"
            n_found += len(path) > 0
        assert n_found >= 2
    # at_start = False: the whole reachable field is expanded
    res = nf.plan(goals, starts, astar=False, at_start=False)
    for k in range(nI):
        path, pot, cyc = orc.navfn_plan(maps[k], goals[k], starts[k], cost_mode=1, allow_unknown=False, at_start=False)
        assert res[k].cycles == cyc and np.array_equal(nf.potential(k).view(np.uint32), pot.view(np.uint32))
        assert np.array_equal(nf.path(k).view(np.uint32), path.view(np.uint32))
    with pytest.raises(nav.NavgpuError):
        nf.plan([[0, 5]], [[5, 5]])  # goal on the border: the reference would index outside its arrays
    nf.close()


@pytest.mark.gpu
def test_navfn_reference_willow_cases_on_gpu(orc, willow):
    """The reference's own two searches (path_calc_test.cpp) through the C-ABI, both at once, bit-exact against the oracle."""
    import navigation_amd as nav
    ny, nx = willow.shape
    nf = nav.NavFn(nx, ny, 2)
    nf.set_costmap(willow, cost_mode=0)
    res = nf.plan([g for _, g in WILLOW_CASES], [s for s, _ in WILLOW_CASES])
    for k, (start, goal) in enumerate(WILLOW_CASES):
        path, pot, cyc = orc.navfn_plan(willow, goal, start, cost_mode=0)
        assert res[k].found and res[k].cycles == cyc
        assert np.array_equal(nf.potential(k).view(np.uint32), pot.view(np.uint32))
        assert np.array_equal(nf.path(k).view(np.uint32), path.view(np.uint32))
    nf.close()


@pytest.mark.gpu
def test_navfn_costmap_from_fleet(orc):
    """NavfnROS::makePlan's hand-over of the costmap, device to device: plans on the master grids a fleet has just updated."""
    import navigation_amd as nav
    from navigation_amd import _lib as N, synth
    n, nI = 200, 3
    insts = [synth.make_instance(n, 150 + i) for i in range(nI)]
    fl = nav.Fleet(nI, n, n, synth.RES, layers=N.LAYER_OBSTACLE | N.LAYER_INFLATION)
    fl.configure_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, synth.inscribed_radius(synth.FOOTPRINT))
    fl.upload(N.GRID_MASTER, np.stack([i["cells"] for i in insts]))
    fl.inflate(boxes=[[0, 0, n, n]] * nI)
    m = fl.master()
    nf = nav.NavFn(n, n, nI)
    nf.set_costmap_from_fleet(fl)
    goals, starts = [[100, 100]] * nI, [[170, 120], [30, 160], [150, 40]]
    res = nf.plan(goals, starts)
    for k in range(nI):
        path, pot, cyc = orc.navfn_plan(m[k], goals[k], starts[k], cost_mode=1)
        assert res[k].cycles == cyc and np.array_equal(nf.potential(k).view(np.uint32), pot.view(np.uint32))
        assert np.array_equal(nf.path(k).view(np.uint32), path.view(np.uint32))
    nf.close()
    fl.close()


# ------------------------------------------------------------------------------------------------ tiled wavefront mode
# navgpu_navfn_plan_wavefront relaxes NavFn::updateCell's rule to its fixed point instead of replaying the reference's
# priority buffers.  The contract (include/navgpu.h), stated on the oracle alone first (CPU) and then checked on
# the HIP path (GPU):
#   (1) the fixed point never lies above the reference's array (a relaxation only lowers a potential, never below the
#       fixed point; the reference stops early, drops buffer entries beyond 10 000 and skips some pushes);
#   (2) where the reference has had time to settle - well below the start cell's potential - the two agree closely;
#   (3) calcPath over the fixed point stays within a cell of calcPath over the reference's array, same end points.
def _hausdorff(a, b):
    d = np.hypot(a[:, None, 0] - b[None, :, 0], a[:, None, 1] - b[None, :, 1])
    return max(d.min(axis=1).max(), d.min(axis=0).max())


def _costarr(cm, cost_mode, allow_unknown=True):
    """NavFn::setCostmap (navfn.cpp:222-283) for cost_mode 1 (isROS), the bytes themselves for 0."""
    if cost_mode == 0:
        return cm.astype(np.int32)
    v = cm.astype(np.int32)
    out = np.full(cm.shape, 254, np.int32)
    lo = v < 253
    out[lo] = np.minimum((50 + 0.8 * v[lo]).astype(np.int32), 253)
    if allow_unknown:
        out[v == 255] = 253
    return out


def _path_cost(path, costarr):
    """Cell cost integrated along a path: what the expansion minimises, whichever cells the path picks."""
    seg = np.hypot(*np.diff(path, axis=0).T)
    mid = (path[1:] + path[:-1]) / 2
    return float((costarr[np.round(mid[:, 1]).astype(int), np.round(mid[:, 0]).astype(int)] * seg).sum())


def _maze_costmap(rs, n):
    """Rooms and doorways with inflated walls: the path has to wind, the wavefront meets tiles more than once."""
    cm = np.zeros((n, n), np.uint8)
    for k in range(24, n - 8, 24):
        cm[k, 4:n - 4] = 254
        cm[4:n - 4, k] = 254
        for _ in range(3):
            a = rs.randint(8, n - 12)
            cm[k, a:a + 4] = 0
            b = rs.randint(8, n - 12)
            cm[b:b + 4, k] = 0
    soft = (rs.random_sample((n, n)) < 0.2) & (cm == 0)
    cm[soft] = rs.randint(1, 200, soft.sum())
    return cm


@pytest.mark.parametrize("start,goal", WILLOW_CASES)
def test_oracle_fixed_point_against_reference_order(orc, willow, start, goal):
    path, pot, _ = orc.navfn_plan(willow, goal, start, cost_mode=0)
    fpath, fpot = orc.navfn_fixed_point(willow, goal, start, cost_mode=0)
    assert not (fpot > pot * (1 + 5e-4)).any()                                # (1)
    ps = pot[start[1], start[0]]
    assert fpot[start[1], start[0]] <= ps and ps - fpot[start[1], start[0]] < 100.0   # one priority block (2 x COST_NEUTRAL)
    iy, ix = np.clip(np.round(path[:, 1]).astype(int), 0, pot.shape[0] - 1), np.clip(np.round(path[:, 0]).astype(int), 0, pot.shape[1] - 1)
    assert np.abs(pot[iy, ix] - fpot[iy, ix]).max() < 100.0                   # (2) along the path: within one block
    assert len(fpath) > 0 and tuple(fpath[0]) == tuple(path[0]) and tuple(fpath[-1]) == tuple(path[-1])
    assert _hausdorff(path, fpath) <= 1.0                                     # (3) on the reference's own two searches
    assert _path_cost(fpath, _costarr(willow, 0)) <= 1.01 * _path_cost(path, _costarr(willow, 0))


def test_oracle_fixed_point_random_and_maze_maps(orc):
    """Away from the reference's own vectors: the rule is not monotone at dc = hf (its polynomial steps from 1.0046 down to 1), so
    which value a cell keeps depends on the order of the relaxations at the 1e-4 level, and where two routes round an obstacle
    at nearly the same cost the two arrays may choose differently - the paths then differ by cells, not their cost."""
    rs = np.random.RandomState(5)
    n_paths = 0
    for it in range(10):
        n = 140
        cm = _maze_costmap(rs, n) if it % 2 else _random_costmap(rs, n, 0.05)
        goal, start = rs.randint(8, n - 8, 2), rs.randint(8, n - 8, 2)
        cm[goal[1], goal[0]] = 0
        cm[start[1], start[0]] = 0
        path, pot, _ = orc.navfn_plan(cm, goal, start, cost_mode=1)
        fpath, fpot = orc.navfn_fixed_point(cm, goal, start, cost_mode=1)
        assert not (fpot > pot * (1 + 5e-4)).any()
        assert fpot[start[1], start[0]] <= pot[start[1], start[0]]
        assert (len(fpath) > 0) == (len(path) > 0)
        if len(path):
            assert tuple(fpath[0]) == tuple(path[0]) and tuple(fpath[-1]) == tuple(path[-1])
            assert _path_cost(fpath, _costarr(cm, 1)) <= 1.05 * _path_cost(path, _costarr(cm, 1))
            n_paths += 1
    assert n_paths >= 5


def _check_wavefront_plan(orc, nf, k, res, cm, goal, start, cost_mode, at_start=True, allow_unknown=True):
    """One plan of the HIP wavefront mode against the oracle: fixed point (potentials, path) and reference order (contract)."""
    fpath, fpot = orc.navfn_fixed_point(cm, goal, start, cost_mode=cost_mode, allow_unknown=allow_unknown)
    path, pot, _ = orc.navfn_plan(cm, goal, start, cost_mode=cost_mode, allow_unknown=allow_unknown, at_start=at_start)
    g = nf.potential(k)
    ps = g[start[1], start[0]]
    reached = ps < 1e9
    assert reached == (fpot[start[1], start[0]] < 1e9)
    # settled region: everything below the start cell's potential (the whole map when the start was never reached or at_start
    # is off).  There the HIP array is the rule's fixed point - equal to the oracle's FIFO relaxation up to the rule's order
    # dependence (a percent or two of the cells, <= 1e-3 relative); beyond it a cell holds POT_HIGH or a value on its way down.
    settled = (fpot < ps) if (reached and at_start) else np.ones_like(fpot, bool)
    a, b = g[settled], fpot[settled]
    assert ((a >= 1e9) == (b >= 1e9)).all()
    fin = b < 1e9
    rel = np.abs(a[fin] - b[fin]) / np.maximum(b[fin], 1.0)
    assert rel.max() <= 1e-3 if fin.any() else True, rel.max()  # (willow: 5.2e-4 at worst, 1.4 % of the cells differ at all)
    assert not (g[~settled] < fpot[~settled] * (1 - 1e-3)).any()  # never below the fixed point
    assert not (g[pot < 1e9] > pot[pot < 1e9] * (1 + 1e-3)).any() if at_start else True  # (1): never above the reference's array
    gpath = nf.path(k)
    assert bool(res.found) == (len(gpath) > 0) == (len(fpath) > 0)
    assert res.path_length == len(gpath)
    if len(gpath):
        assert tuple(gpath[0]) == tuple(float(v) for v in start) and tuple(gpath[-1]) == tuple(float(v) for v in goal)
        ca = _costarr(cm, cost_mode, allow_unknown)
        assert _path_cost(gpath, ca) <= 1.02 * _path_cost(fpath, ca)
        if len(path):
            assert _path_cost(gpath, ca) <= 1.05 * _path_cost(path, ca)          # (3) against the reference order
    return len(gpath) > 0


@pytest.mark.gpu
def test_navfn_wavefront_willow_cases(orc, willow):
    """The reference's two searches of path_calc_test.cpp in the tiled wavefront mode, both at once; run twice: identical bits."""
    import navigation_amd as nav
    ny, nx = willow.shape
    nf = nav.NavFn(nx, ny, 2)
    nf.set_costmap(willow, cost_mode=0)
    goals, starts = [g for _, g in WILLOW_CASES], [s for s, _ in WILLOW_CASES]
    res = nf.plan_wavefront(goals, starts)
    first = [(nf.potential(k).copy(), nf.path(k).copy(), res[k].cycles) for k in range(2)]
    for k, (start, goal) in enumerate(WILLOW_CASES):
        assert _check_wavefront_plan(orc, nf, k, res[k], willow, goal, start, 0)
        path, _, _ = orc.navfn_plan(willow, goal, start, cost_mode=0)
        assert _hausdorff(nf.path(k), path) <= 1.0
    # at_start off: the whole reachable map settles (no early stop), both searches at once
    res = nf.plan_wavefront(goals, starts, at_start=False)
    for k, (start, goal) in enumerate(WILLOW_CASES):
        assert _check_wavefront_plan(orc, nf, k, res[k], willow, goal, start, 0, at_start=False)
        assert res[k].cycles > first[k][2]
    res = nf.plan_wavefront(goals, starts)
    for k in range(2):
        assert res[k].cycles == first[k][2]
        assert np.array_equal(nf.potential(k).view(np.uint32), first[k][0].view(np.uint32))
        assert np.array_equal(nf.path(k).view(np.uint32), first[k][1].view(np.uint32))
    # the reference-order mode on the same handle afterwards: still bit-exact (the two modes share the arrays)
    res = nf.plan(goals, starts)
    for k, (start, goal) in enumerate(WILLOW_CASES):
        path, pot, cyc = orc.navfn_plan(willow, goal, start, cost_mode=0)
        assert res[k].cycles == cyc and np.array_equal(nf.potential(k).view(np.uint32), pot.view(np.uint32))
        assert np.array_equal(nf.path(k).view(np.uint32), path.view(np.uint32))
    nf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("at_start", [True, False])
def test_navfn_wavefront_batch_random_and_maze(orc, at_start):
    import navigation_amd as nav
    rs = np.random.RandomState(17)
    n, nI = 150, 8   # 150 = 4 tiles of 32 + a ragged fifth
    maps = np.stack([_maze_costmap(rs, n) if k % 2 else _random_costmap(rs, n, 0.05) for k in range(nI)])
    goals = rs.randint(8, n - 8, (nI, 2))
    starts = rs.randint(8, n - 8, (nI, 2))
    for k in range(nI):
        maps[k][goals[k][1], goals[k][0]] = 0
        maps[k][starts[k][1], starts[k][0]] = 0
    maps[nI - 1][starts[nI - 1][1] - 2:starts[nI - 1][1] + 3, starts[nI - 1][0] - 2:starts[nI - 1][0] + 3] = 254  # walled in: no plan
    maps[nI - 1][starts[nI - 1][1], starts[nI - 1][0]] = 0
    nf = nav.NavFn(n, n, nI)
    n_found = 0
    for allow_unknown in (True, False):
        nf.set_costmap(maps, cost_mode=1, allow_unknown=allow_unknown)
        res = nf.plan_wavefront(goals, starts, at_start=at_start)
        for k in range(nI):
            n_found += _check_wavefront_plan(orc, nf, k, res[k], maps[k], goals[k], starts[k], 1, at_start=at_start, allow_unknown=allow_unknown)
        assert not res[nI - 1].found
    assert n_found >= 8
    # a sub-range of the batch leaves the other plans' results alone
    keep = nf.potential(0).copy()
    nf.plan_wavefront(goals[3:5], starts[3:5], first=3, at_start=at_start)
    assert np.array_equal(nf.potential(0).view(np.uint32), keep.view(np.uint32))
    with pytest.raises(nav.NavgpuError):
        nf.plan_wavefront([[0, 5]], [[5, 5]])
    nf.close()


# ------------------------------------------------------------------------------------------------ global_planner
GP_VARIANTS = [dict(), dict(use_quadratic=0), dict(use_grid_path=1), dict(old_navfn_behavior=1), dict(use_dijkstra=0),
               dict(use_dijkstra=0, use_quadratic=0, use_grid_path=1), dict(allow_unknown=0, cost_factor=0.55, neutral_cost=66)]


def _gp_case(rs, n):
    cm = _random_costmap(rs, n, 0.03)
    start = rs.uniform(8, n - 9, 2)
    goal = rs.uniform(8, n - 9, 2)
    for x, y in (start, goal):
        cm[int(y) - 1:int(y) + 3, int(x) - 1:int(x) + 3] = 0
    return cm, start, goal


def test_oracle_global_planner_basic_properties(orc):
    """Open field: every variant reaches the goal; the traceback starts at the goal and ends at the start."""
    n = 60
    cm = np.zeros((n, n), np.uint8)
    for kw in GP_VARIANTS:
        start, goal = (12.3, 40.7), (47.6, 15.2)
        if kw.get("old_navfn_behavior"):
            start, goal = (12.0, 40.0), (47.0, 15.0)
        path, pot, legal, cyc = orc.global_planner_plan(cm, start, goal, (int(goal[0]), int(goal[1])), **kw)
        assert legal, kw
        if not kw.get("use_dijkstra", 1) and not kw.get("use_grid_path"):
            # A*'s narrow corridor + the interpolated gradient descent: the descent steps onto a cell A* never reached and
            # `int minp = potential[stc]` (gradient_path.cpp:119) ends the trace -- the reference's known A*/gradient failure
            assert len(path) == 0
            continue
        assert len(path) > 1, kw
        assert abs(path[0][0] - goal[0]) < 1.01 and abs(path[0][1] - goal[1]) < 1.01
        assert abs(path[-1][0] - start[0]) < 1.01 and abs(path[-1][1] - start[1]) < 1.01
        assert pot[0, 0] >= 1e9  # the outline is lethal: never assigned
    cm[:, 30] = 254
    path, pot, legal, cyc = orc.global_planner_plan(cm, (12.3, 40.7), (47.6, 15.2), (47, 15))
    assert not legal and len(path) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("kw", GP_VARIANTS)
def test_global_planner_batch_matches_oracle(orc, kw):
    import navigation_amd as nav
    rs = np.random.RandomState(21)
    n, nI = 110, 5
    cases = [_gp_case(rs, n) for _ in range(nI)]
    nf = nav.NavFn(n, n, nI)
    nf.set_costmap(np.stack([c[0] for c in cases]), cost_mode=0)
    starts = np.array([c[1] for c in cases])
    goals = np.array([c[2] for c in cases])
    if kw.get("old_navfn_behavior"):
        starts, goals = np.floor(starts), np.floor(goals)
    cells = goals.astype(np.int32)
    res = nf.global_planner_plan(starts, goals, cells, **kw)
    n_found = 0
    for k in range(nI):
        path, pot, legal, cyc = orc.global_planner_plan(cases[k][0], starts[k], goals[k], cells[k], **kw)
        assert res[k].cycles == cyc, (k, res[k].cycles, cyc)
        assert np.array_equal(nf.potential(k).view(np.uint32), pot.view(np.uint32)), f"plan {k}: potential arrays differ"
        assert bool(res[k].found) == (len(path) > 0) and res[k].path_length == len(path)
        assert np.array_equal(nf.path(k).view(np.uint32), path.view(np.uint32)), f"plan {k}: paths differ"
        n_found += len(path) > 0
    assert n_found >= 2
    nf.close()


@pytest.mark.gpu
def test_global_planner_on_willow(orc, willow):
    """The reference's willow map (as a costmap: 254 obstacles) through both expanders and both tracebacks."""
    import navigation_amd as nav
    ny, nx = willow.shape
    nf = nav.NavFn(nx, ny, 2)
    nf.set_costmap(willow, cost_mode=0)
    starts, goals = [[428.3, 746.6], [350.5, 400.5]], [[350.4, 450.2], [350.5, 450.5]]
    # the pinch point is a corridor of 253s: passable for navfn's tests (obstacles >= 254), for global_planner only with
    # lethal_cost 255 (dijkstra.h:80-92: c < lethal_cost - 1 is traversable)
    for kw in (dict(lethal_cost=255), dict(lethal_cost=255, use_dijkstra=0, use_grid_path=1), dict(lethal_cost=255, cost_factor=0.2, use_quadratic=0)):
        res = nf.global_planner_plan(starts, goals, np.floor(goals).astype(np.int32), **kw)
        for k in range(2):
            path, pot, legal, cyc = orc.global_planner_plan(willow, starts[k], goals[k], [int(goals[k][0]), int(goals[k][1])], **kw)
            assert legal and res[k].found and res[k].cycles == cyc
            assert np.array_equal(nf.potential(k).view(np.uint32), pot.view(np.uint32))
            assert np.array_equal(nf.path(k).view(np.uint32), path.view(np.uint32))
    nf.close()


@pytest.mark.gpu
def test_navfn_wavefront_tile_edges_and_small_maps(orc):
    """Seeds on tile borders (the neighbour tile has to start in round 0 even when the seed's own border cells are obstacles),
    maps smaller than a tile, ragged last tiles, a start next to the goal, a start on an obstacle."""
    import navigation_amd as nav
    cases = []
    cm = np.zeros((70, 70), np.uint8)        # goal in the corner cell of tile (1, 1); its tile-side neighbours are walls
    cm[33, 32] = cm[32, 33] = cm[33, 33] = 254
    cases.append((cm, (32, 32), (5, 60)))
    cm = np.zeros((70, 70), np.uint8)        # goal in the last column of tile (0, 0), walls above and below it
    cm[9, 31] = cm[11, 31] = cm[10, 30] = 254
    cases.append((cm, (31, 10), (60, 50)))
    cm = np.zeros((9, 9), np.uint8)          # one partial tile
    cases.append((cm, (2, 2), (6, 6)))
    cm = np.zeros((33, 65), np.uint8)        # 3 x 2 tiles, the last ones one cell wide / high
    cm[16, 5:60] = 254
    cases.append((cm, (10, 5), (60, 28)))
    cm = np.zeros((40, 40), np.uint8)        # start beside the goal
    cases.append((cm, (20, 20), (21, 20)))
    cm = np.zeros((40, 40), np.uint8)        # start on an obstacle: never reached
    cm[30, 30] = 254
    cases.append((cm, (8, 8), (30, 30)))
    for cm, goal, start in cases:
        ny, nx = cm.shape
        nf = nav.NavFn(nx, ny, 1)
        nf.set_costmap(cm, cost_mode=1)
        for at_start in (True, False):
            res = nf.plan_wavefront([goal], [start], at_start=at_start)
            _check_wavefront_plan(orc, nf, 0, res[0], cm, goal, start, 1, at_start=at_start)
        nf.close()


# ------------------------------------------------------------------------------------------------ global_planner, tiled wavefront
GP_WF_VARIANTS = [dict(), dict(use_quadratic=0), dict(use_grid_path=1), dict(old_navfn_behavior=1), dict(allow_unknown=0, cost_factor=0.55, neutral_cost=66)]


def _gp_cost(cm, kw):
    """DijkstraExpansion::getCost (dijkstra.h:78-87) of every cell, lethal where it is not traversable."""
    pr = dict(lethal_cost=253, neutral_cost=50, cost_factor=3.0, allow_unknown=1)
    pr.update({k: v for k, v in kw.items() if k in pr})
    c = cm.astype(np.float32)
    ok = (c < pr["lethal_cost"] - 1) | ((c == 255) & bool(pr["allow_unknown"]))
    out = np.full(cm.shape, float(pr["lethal_cost"]), np.float32)
    out[ok] = np.minimum(c[ok] * np.float32(pr["cost_factor"]) + pr["neutral_cost"], pr["lethal_cost"] - 1)
    return out


def _check_gp_wavefront(orc, nf, k, res, cm, start, goal, kw, against_reference_path=True):
    cell = [int(goal[0]), int(goal[1])]
    fpath, fpot, flegal, _ = orc.global_planner_plan(cm, start, goal, cell, fixed_point=True, **kw)
    path, pot, legal, _ = orc.global_planner_plan(cm, start, goal, cell, **kw)
    g = nf.potential(k)
    pg = g[cell[1], cell[0]]
    assert (pg < 1e9) == flegal == legal
    # Expander::clearEndpoint fills the 5 x 5 block round the goal with potentials of a cheaper cost (costs + neutral) wherever the
    # expansion had not reached: what it finds there depends on where the expansion stopped, so that block is left out
    far = np.ones(cm.shape, bool)
    far[max(cell[1] - 2, 0):cell[1] + 3, max(cell[0] - 2, 0):cell[0] + 3] = False
    settled = (fpot < pg) & far if flegal else far
    a, b = g[settled], fpot[settled]
    assert ((a >= 1e9) == (b >= 1e9)).all()
    fin = b < 1e9
    if fin.any():  # (the rule's order dependence is 0.46 % of a cell's cost per cell, and getCost's costs reach 252: 3e-3 observed)
        assert (np.abs(a[fin] - b[fin]) / np.maximum(b[fin], 1.0)).max() <= 1e-2
    m = (pot < 1e9) & far
    assert not (g[m] > pot[m] * (1 + 1e-2)).any()        # never above the reference-order array
    gpath = nf.path(k)
    assert bool(res.found) == (len(gpath) > 0) and res.path_length == len(gpath)
    assert (len(gpath) > 0) == (len(fpath) > 0)
    if len(gpath):
        ca = _gp_cost(cm, kw)
        assert np.array_equal(gpath[0], fpath[0]) and np.array_equal(gpath[-1], fpath[-1])
        assert _path_cost(gpath, ca) <= 1.03 * _path_cost(fpath, ca) + 1.0
        if len(path) and against_reference_path:
            assert _path_cost(gpath, ca) <= 1.06 * _path_cost(path, ca) + 1.0
    return len(gpath) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("kw", GP_WF_VARIANTS)
def test_global_planner_wavefront_batch(orc, kw):
    """navgpu_global_planner_plan_wavefront: DijkstraExpansion's rule relaxed to its fixed point by LDS tiles, then the reference's
    own clearEndpoint + traceback; against the oracle's fixed point (potentials, path cost) and its reference-order run (contract)."""
    import navigation_amd as nav
    rs = np.random.RandomState(33)
    n, nI = 110, 6
    cases = [_gp_case(rs, n) for _ in range(nI)]
    nf = nav.NavFn(n, n, nI)
    nf.set_costmap(np.stack([c[0] for c in cases]), cost_mode=0)
    starts = np.array([c[1] for c in cases])
    goals = np.array([c[2] for c in cases])
    if kw.get("old_navfn_behavior"):
        starts, goals = np.floor(starts), np.floor(goals)
    cells = goals.astype(np.int32)
    res = nf.global_planner_plan(starts, goals, cells, wavefront=True, **kw)
    # GradientPath's half-cell descent zig-zags for dozens of points next to the single-cell cost spikes of these maps - in the
    # reference-order array and in the fixed point alike, for different numbers of steps - so path against path is compared for
    # GridPath only here (and for both tracebacks on the reference's willow map below); potentials are compared for every variant
    n_found = sum(_check_gp_wavefront(orc, nf, k, res[k], cases[k][0], starts[k], goals[k], kw, against_reference_path=bool(kw.get("use_grid_path")))
                  for k in range(nI))
    assert n_found >= 2
    again = nf.global_planner_plan(starts, goals, cells, wavefront=True, **kw)
    assert [(r.found, r.path_length, r.cycles) for r in again] == [(r.found, r.path_length, r.cycles) for r in res]
    with pytest.raises(nav.NavgpuError):
        nf.global_planner_plan(starts, goals, cells, wavefront=True, use_dijkstra=0)
    # the reference-order call on the same handle afterwards: still bit-exact
    res = nf.global_planner_plan(starts, goals, cells, **kw)
    for k in range(nI):
        path, pot, legal, cyc = orc.global_planner_plan(cases[k][0], starts[k], goals[k], cells[k], **kw)
        assert res[k].cycles == cyc and np.array_equal(nf.potential(k).view(np.uint32), pot.view(np.uint32))
        assert np.array_equal(nf.path(k).view(np.uint32), path.view(np.uint32))
    nf.close()


@pytest.mark.gpu
def test_global_planner_wavefront_without_outline(orc):
    """outline_map = 0: nothing makes the border rows lethal.  The reference's updateCell would read outside its arrays for a cell of
    row 0 / ny - 1; the wavefront mode never updates those rows, like its checker (GlobalPlannerOracle::dijkstraFixedPoint) - free
    border cells all round, starts and goals six cells in (the expansion runs along the border rows on its way)."""
    import navigation_amd as nav
    rs = np.random.RandomState(41)
    n, nI = 96, 4
    cms = np.stack([_random_costmap(rs, n, 0.02) for _ in range(nI)])
    cms[:, :4, :] = 0
    cms[:, -4:, :] = 0
    cms[:, :, :4] = 0
    cms[:, :, -4:] = 0
    cms[:, :, 0] = cms[:, :, -1] = 254  # (the reference's arrays are flat: without lethal end columns a row's end is its neighbour's start)
    starts = np.array([[6.4, 6.6], [n - 7.5, 6.2], [40.3, n - 7.4], [6.7, 50.5]])
    goals = np.array([[n - 7.3, n - 7.6], [6.5, n - 7.5], [41.5, 6.6], [n - 7.2, 48.4]])
    for (x, y), cm in zip(np.concatenate([starts, goals]), np.concatenate([cms, cms])):
        cm[max(int(y) - 2, 0):int(y) + 3, max(int(x) - 2, 0):int(x) + 3] = 0
    nf = nav.NavFn(n, n, nI)
    nf.set_costmap(cms, cost_mode=0)
    cells = goals.astype(np.int32)
    kw = dict(outline_map=0)
    res = nf.global_planner_plan(starts, goals, cells, wavefront=True, **kw)
    for k in range(nI):
        cell = [int(cells[k][0]), int(cells[k][1])]
        fpath, fpot, flegal, _ = orc.global_planner_plan(cms[k], starts[k], goals[k], cell, fixed_point=True, **kw)
        g = nf.potential(k)
        pg = g[cell[1], cell[0]]
        assert flegal and pg < 1e9 and bool(res[k].found) == (len(fpath) > 0)  # (the traceback may give up beside the never-updated border rows: on both sides alike)
        assert (g[0] >= 1e9).all() and (g[-1] >= 1e9).all() and (fpot[0] >= 1e9).all() and (fpot[-1] >= 1e9).all()  # never updated, either side
        far = np.ones(cms[k].shape, bool)
        far[max(cell[1] - 2, 0):cell[1] + 3, max(cell[0] - 2, 0):cell[0] + 3] = False
        settled = (fpot < pg) & far
        a, b = g[settled], fpot[settled]
        assert ((a >= 1e9) == (b >= 1e9)).all()
        fin = b < 1e9
        assert fin.any() and (np.abs(a[fin] - b[fin]) / np.maximum(b[fin], 1.0)).max() <= 1e-2
        gpath = nf.path(k)
        assert len(gpath) == 0 if len(fpath) == 0 else (np.array_equal(gpath[0], fpath[0]) and np.array_equal(gpath[-1], fpath[-1]))
    nf.close()


@pytest.mark.gpu
def test_global_planner_wavefront_on_willow(orc, willow):
    import navigation_amd as nav
    ny, nx = willow.shape
    nf = nav.NavFn(nx, ny, 2)
    nf.set_costmap(willow, cost_mode=0)
    starts, goals = np.array([[428.3, 746.6], [350.5, 400.5]]), np.array([[350.4, 450.2], [350.5, 450.5]])
    for kw in (dict(lethal_cost=255), dict(lethal_cost=255, cost_factor=0.2, use_quadratic=0)):
        res = nf.global_planner_plan(starts, goals, np.floor(goals).astype(np.int32), wavefront=True, **kw)
        for k in range(2):
            assert _check_gp_wavefront(orc, nf, k, res[k], willow, starts[k], goals[k], kw)
    nf.close()


# ------------------------------------------------------------------------------------------------ oracle/_ref: global_planner pieces
def test_ref_potential_calculators_match_oracle(orc):
    """The reference's own PotentialCalculator / QuadraticCalculator (compiled in place into oracle/_ref/libref_gp.so)
    against the oracle's calculatePotential, bit for bit, over random neighbourhoods incl. POT_HIGH neighbours."""
    R = orc.ref_gp()
    if R is None:
        pytest.skip("oracle/_ref/libref_gp.so not built (reference tree absent)")
    rs = np.random.RandomState(3)
    nx, ny = 40, 30
    pot = rs.uniform(0, 4000, (ny, nx)).astype(np.float32)
    pot[rs.rand(ny, nx) < 0.3] = 1e10
    cells = (rs.randint(1, ny - 1, 5000) * nx + rs.randint(1, nx - 1, 5000)).astype(np.int32)
    cost = rs.randint(1, 255, 5000).astype(np.uint8)
    prev = np.where(rs.rand(5000) < 0.5, -1.0, rs.uniform(0, 4000, 5000)).astype(np.float32)
    for quadratic in (0, 1):
        a, b = np.zeros(5000, np.float32), np.zeros(5000, np.float32)
        R.ref_gp_calculate_potential(quadratic, pot.copy().reshape(-1), nx, ny, cost, cells, prev, 5000, a)
        orc.lib().orc_gp_calculate_potential(quadratic, pot.reshape(-1), nx, ny, cost, cells, prev, 5000, b)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), quadratic


def test_ref_grid_path_matches_oracle(orc):
    """The reference's own GridPath::getPath over potentials the oracle's expanders produced (Dijkstra and A*)."""
    R = orc.ref_gp()
    if R is None:
        pytest.skip("oracle/_ref/libref_gp.so not built (reference tree absent)")
    rs = np.random.RandomState(8)
    n, n_paths = 90, 0
    for it in range(12):
        cm, start, goal = _gp_case(rs, n)
        _, pot, legal, _ = orc.global_planner_plan(cm, start, goal, goal.astype(np.int32), use_dijkstra=it % 2, use_grid_path=1)
        if not legal:
            continue
        a, b = np.zeros((4 * n * n, 2), np.float32), np.zeros((4 * n * n, 2), np.float32)
        na = R.ref_gp_grid_path(pot.copy().reshape(-1), n, n, start[0], start[1], goal[0], goal[1], a.reshape(-1), len(a))
        nb = orc.lib().orc_gp_grid_path(pot.reshape(-1), n, n, start[0], start[1], goal[0], goal[1], b.reshape(-1), len(b))
        assert na == nb and np.array_equal(a[:na].view(np.uint32), b[:nb].view(np.uint32))
        n_paths += na > 0
    assert n_paths >= 4


def test_oracle_astar_fixed_point_bounds_the_reference(orc):
    """f-4, A*: AStarExpansion's rule relaxed to its fixed point (GlobalPlannerOracle::astarFixedPoint) reaches the goal exactly when
    the reference's expansion does and never lies above its array - but the reference sets a cell ONCE, when its first neighbour pops,
    so its array holds first-touch values and GridPath follows THAT: over 60 random maps the fixed point's path cost was 0.62 ... 1.47 x
    the reference path's (55 % within 6 %).  No path-level contract to hold a tiled wavefront to: A* stays on the
    reference-order replay (navgpu_global_planner_plan), and this test pins what does hold."""
    rs = np.random.RandomState(7)
    for _ in range(6):
        cm, start, goal = _gp_case(rs, 90)
        start, goal = np.floor(start), np.floor(goal)
        cell = [int(goal[0]), int(goal[1])]
        for kw in (dict(use_dijkstra=0, use_grid_path=1), dict(use_dijkstra=0, use_grid_path=1, use_quadratic=0)):
            p_ref, pot_ref, l_ref, _ = orc.global_planner_plan(cm, start, goal, cell, **kw)
            p_fp, pot_fp, l_fp, _ = orc.global_planner_plan(cm, start, goal, cell, fixed_point=True, **kw)
            assert l_ref == l_fp and (len(p_ref) > 0) == (len(p_fp) > 0)
            m = pot_ref < 1e9
            assert (pot_fp[m] <= pot_ref[m] * (1 + 1e-4) + 1e-3).all()
