"""Parity tests added in round 3 (same bar as the others: the HIP path through the C-ABI against the CPU oracle,
bit-exact on bytes / indices / codes, 1e-5 on cost floats): SURVEY 8(b)'s threading contract on the device path, a
reconfigure that fails half-way, and global_planner's border reads."""
import os
import threading
import time

import numpy as np
import pytest

from test_gpu_parity import L, _inflated_instance, _random_map, nav  # noqa: F401

pytestmark = pytest.mark.gpu


def _oracle_cycle(orc, cfg, master, pos, vel, plan, fp, res):
    p = orc.DwaPlanner(master, res, 0.0, 0.0, orc.DwaConfig(**cfg.as_dict()))
    p.set_plan()
    o, _, _, cfull, ost = p.cycle(np.asarray(pos, np.float32), np.asarray(vel, np.float32), plan, fp)
    return o


# ----------------------------------------------------------------------------------------------
# SURVEY 8(b) "Threading": dynamic_reconfigure calls DWAPlanner::reconfigure on the spinner thread while move_base's
# control thread is inside findBestPath; configuration_mutex_ (dwa_planner.cpp:55,301) orders them.  Here a second
# host thread calls navgpu_planner_configure on the SAME fleet - between the control thread's cycles (phase 1: both
# hold an application lock around whole cycles, as navgpu::DWAPlannerROS does, so every cycle's configuration is
# known) and during them (phase 2: no application lock, the library's per-fleet mutex alone; a cycle then ran under
# whichever configuration was in force when it was launched).  The two configurations differ in their sample counts:
# every switch re-allocates the planner's device tables.  A third thread reads state back all the while.
# ----------------------------------------------------------------------------------------------
def test_reconfigure_from_second_thread(nav, orc):
    from navigation_amd import synth
    N = L(nav)
    n, nI = 200, 3
    insts = [_inflated_instance(orc, n, 30 + i, synth) for i in range(nI)]
    masters = np.stack([i["master"] for i in insts])
    cfgs = [nav.DwaConfig(vx_samples=6, vy_samples=4, vth_samples=7, sim_time=1.2, sim_granularity=0.1, discretize_by_time=1),
            nav.DwaConfig(vx_samples=11, vy_samples=3, vth_samples=12, sim_time=1.6, sim_granularity=0.1, discretize_by_time=1, max_vel_x=0.7,
                          max_trans_vel=0.7)]
    fl = nav.Fleet(nI, n, n, synth.RES, layers=N.LAYER_OBSTACLE, max_sim_steps=64, max_plan=256)
    fl.configure_planner(cfgs[0])
    fl.set_footprint(synth.FOOTPRINT)
    fl.upload(N.GRID_MASTER, masters)
    fl.set_plan()
    pos0 = np.stack([i["pos"] for i in insts]).astype(np.float32)
    vel = np.stack([i["vel"] for i in insts]).astype(np.float32)
    plans = np.stack([i["plan"] for i in insts])
    rs = np.random.RandomState(5)

    app_lock = threading.Lock()
    state = {"cfg": 0, "stop": False, "use_lock": True, "switches": 0, "reads": 0, "errors": []}

    def reconfigurer():
        while not state["stop"]:
            try:
                if state["use_lock"]:
                    with app_lock:
                        k = 1 - state["cfg"]
                        fl.configure_planner(cfgs[k])
                        state["cfg"] = k
                else:
                    k = 1 - state["cfg"]
                    fl.configure_planner(cfgs[k])
                    state["cfg"] = k
                state["switches"] += 1
            except Exception as e:  # noqa: BLE001
                state["errors"].append(repr(e))
                return
            time.sleep(0.0005)

    def reader():
        while not state["stop"]:
            try:
                fl.wavefront_levels()
                fl.oscillation()
                state["reads"] += 1
            except Exception as e:  # noqa: BLE001
                state["errors"].append(repr(e))
                return
            time.sleep(0.0002)

    threads = [threading.Thread(target=reconfigurer), threading.Thread(target=reader)]
    for t in threads:
        t.start()
    record = []  # (phase, cfg index or None, positions, results)
    try:
        for phase, cycles in ((1, 24), (2, 24)):
            state["use_lock"] = phase == 1
            time.sleep(0.01)
            for c in range(cycles):
                pos = (pos0 + rs.normal(size=pos0.shape) * np.array([0.02, 0.02, 0.05])).astype(np.float32)
                if phase == 1:
                    with app_lock:
                        k = state["cfg"]
                        fl.set_plan()  # resetOscillationFlags: every cycle stands alone, as the oracle's does
                        res = fl.find_best_path(pos, vel, plans)
                else:
                    k = None
                    fl.set_plan()
                    res = fl.find_best_path(pos, vel, plans)
                record.append((phase, k, pos, [(r.best_index, r.cost, tuple(r.drive), r.n_valid, r.n_scored) for r in res]))
                time.sleep(0.0003)
    finally:
        state["stop"] = True
        for t in threads:
            t.join()
    assert not state["errors"], state["errors"]
    assert state["switches"] >= 8 and state["reads"] >= 8, (state["switches"], state["reads"])
    seen = {0: 0, 1: 0}
    for phase, k, pos, res in record:
        for i in range(nI):
            cands = [k] if k is not None else [0, 1]
            ok = False
            for kk in cands:
                o = _oracle_cycle(orc, cfgs[kk], masters[i], pos[i], vel[i], plans[i], synth.FOOTPRINT, synth.RES)
                if (res[i][0], res[i][3], res[i][4]) == (o.best_index, o.n_valid, o.n_scored) and abs(res[i][1] - o.cost) <= 1e-5 and \
                        list(res[i][2]) == list(o.drive):
                    ok = True
                    seen[kk] += 1
                    break
            assert ok, (phase, k, i, res[i])
    assert seen[0] > 0 and seen[1] > 0, seen  # both configurations really ran
    print(f"reconfigure thread: {state['switches']} switches, reader {state['reads']} reads, cycles under cfg0/cfg1: {seen}")
    fl.close()


# ----------------------------------------------------------------------------------------------
# A reconfigure that fails half-way must leave the configuration in use intact (round-2 advisor finding): the
# reference-order mode needs n_instances x 5 x cells x 16 B for its heaps; when that allocation fails the layer keeps
# inflating with the old radius, the old table and the old (exact) mode.
# ----------------------------------------------------------------------------------------------
def test_inflation_configure_failure_keeps_previous_configuration(nav, orc):
    N = L(nav)
    n = 128
    rs = np.random.RandomState(77)
    maps = np.stack([_random_map(rs, n, 0.01, 0.0) for _ in range(2)])
    fl = nav.Fleet(2, n, n, 0.05, layers=N.LAYER_INFLATION)
    fl.configure_inflation(0.55, 10.0, 0.2)
    fl.upload(N.GRID_MASTER, maps)
    fl.inflate(boxes=[[0, 0, n, n]] * 2)
    want = fl.master()
    assert np.array_equal(want[0], orc.inflate(maps[0], 0.05, 0.55, 10.0, 0.2, exact=True))
    fl.set_alloc_limit(1 << 20)  # the heaps (2 x 5 x 16384 x 16 B = 2.6 MB) cannot be allocated
    try:
        with pytest.raises(Exception):
            fl.configure_inflation(1.0, 3.0, 0.35, priority_queue_order=True)
    finally:
        fl.set_alloc_limit(0)
    fl.upload(N.GRID_MASTER, maps)
    fl.inflate(boxes=[[0, 0, n, n]] * 2)  # old radius, old table, exact mode - and no launch on null heaps
    assert np.array_equal(fl.master(), want)
    # the same request succeeds once memory is there, and then gives the reference's walk
    fl.configure_inflation(1.0, 3.0, 0.35, priority_queue_order=True)
    fl.upload(N.GRID_MASTER, maps)
    fl.inflate(boxes=[[0, 0, n, n]] * 2)
    assert np.array_equal(fl.master()[1], orc.inflate(maps[1], 0.05, 1.0, 3.0, 0.35, exact=False))
    fl.close()


# ----------------------------------------------------------------------------------------------
# global_planner with nothing outlining the map (outline_map = 0) or an outline below lethal_cost (A*, lethal_cost
# 255): border cells get expanded and the reference reads one row outside its arrays (planner_core.cpp:296,
# dijkstra.cpp:152-190, astar.cpp:88-104).  The device treats such neighbours as unreached lethal cells.  Free maps, so
# the expansion really runs along the border; a search that stays inside a lethal frame still equals the oracle.
# ----------------------------------------------------------------------------------------------
def test_global_planner_border_cells_stay_in_bounds(nav, orc):
    n, n_plans = 96, 4
    nf = nav.NavFn(n, n, n_plans)
    free = np.zeros((n_plans, n, n), np.uint8)
    nf.set_costmap(free, cost_mode=0)
    starts = np.array([[3.4, 3.6], [90.2, 4.5], [5.5, 90.5], [48.0, 48.0]])
    goals = np.array([[90.5, 91.5], [4.5, 92.3], [92.5, 3.5], [3.5, 3.5]])
    cells = np.floor(goals).astype(np.int32)
    for kw in (dict(outline_map=0), dict(outline_map=0, use_dijkstra=0, use_grid_path=1), dict(lethal_cost=255, use_dijkstra=0, use_grid_path=1),
               dict(outline_map=0, use_quadratic=0), dict(outline_map=0, use_dijkstra=0)):
        res = nf.global_planner_plan(starts, goals, cells, **kw)
        for k in range(n_plans):
            pot = nf.potential(k)
            assert np.isfinite(pot).all() and pot[cells[k][1], cells[k][0]] < 1e10, (kw, k)
            if kw.get("use_grid_path") or kw.get("use_dijkstra", 1):
                assert res[k].found and len(nf.path(k)) == res[k].path_length > 0, (kw, k)
    # inside a lethal frame the border is never entered: outline_map 0 == the oracle with the outline drawn
    framed = np.zeros((n, n), np.uint8)
    framed[0, :] = framed[-1, :] = framed[:, 0] = framed[:, -1] = 254
    framed[20:70, 40] = 254
    nf.set_costmap(np.stack([framed] * n_plans), cost_mode=0)
    for kw in (dict(outline_map=0), dict(outline_map=0, use_dijkstra=0, use_grid_path=1)):
        res = nf.global_planner_plan(starts, goals, cells, **kw)
        okw = dict(kw, outline_map=1)
        for k in range(n_plans):
            path, pot, legal, cyc = orc.global_planner_plan(framed, starts[k], goals[k], cells[k], **okw)
            assert bool(res[k].found) == bool(legal and len(path) > 0) and res[k].cycles == cyc
            assert np.array_equal(nf.potential(k).view(np.uint32), pot.view(np.uint32))
            if res[k].found:
                assert np.array_equal(nf.path(k).view(np.uint32), path.view(np.uint32))
    nf.close()


# ----------------------------------------------------------------------------------------------
# bench.py runs the fleet as stream groups (4 fleets of 64 robots on 4 HIP streams, their cycles interleaved and their
# kernels overlapping).  Robots are independent, so the partition and the interleaving must not show in any result: the
# same 12 robots as ONE fleet and as 3 groups of 4 whose cycles are queued round-robin without waiting for one another
# give the same winners, costs, counters and master grids - and both equal the oracle.
# ----------------------------------------------------------------------------------------------
def test_stream_groups_equal_one_fleet(nav, orc):
    import bench
    from navigation_amd import synth
    n, n_rob, G = 400, 12, 3
    one, insts, cfg = bench.build_fleet(nav, n_rob, n, seed0=500)
    g_one = bench.Group(nav, one, insts, seed=9)
    groups = []
    for gi in range(G):
        fl, gi_insts, _ = bench.build_fleet(nav, n_rob // G, n, seed0=500 + gi * (n_rob // G))
        groups.append(bench.Group(nav, fl, gi_insts, seed=9))
    # the same pose schedule for a robot whichever fleet it is in
    for gi, g in enumerate(groups):
        lo = gi * g.n
        g.sched.pos = [p[lo:lo + g.n].copy() for p in g_one.sched.pos]
        for c, st in enumerate(g.states):
            for i in range(g.n):
                st[i].pos[:] = list(g_one.states[c][lo + i].pos)
    res_one, res_grp = [], []
    m_first = None
    for k in range(4):
        g_one.cycle(k)
        g_one.collect()
        res_one.append([(r.best_index, r.cost, tuple(r.drive), r.n_valid, r.n_scored) for r in g_one.rbuf])
        if k == 0:
            m_first = one.master()
    for k in range(4):  # queued round-robin, results read one cycle later (the bench's loop)
        for g in groups:
            g.cycle(k)
        cyc = []
        for g in groups:
            g.collect()
            cyc += [(r.best_index, r.cost, tuple(r.drive), r.n_valid, r.n_scored) for r in g.rbuf]
        res_grp.append(cyc)
    assert res_one == res_grp
    m_one = one.master()
    m_grp = np.concatenate([g.fl.master() for g in groups])
    assert np.array_equal(m_one, m_grp)
    # ... and the first cycle (fresh oscillation flags) against the oracle on that cycle's costmap
    ocfg = orc.DwaConfig(**cfg.as_dict())
    for i in (0, 5, 11):
        p = orc.DwaPlanner(m_first[i], synth.RES, 0.0, 0.0, ocfg)
        p.set_plan()
        st = g_one.states[0][i]
        o, _, _, _, _ = p.cycle(np.array(list(st.pos), np.float32), np.array(list(st.vel), np.float32), insts[i]["plan"], synth.FOOTPRINT)
        assert (res_one[0][i][0], res_one[0][i][3], res_one[0][i][4]) == (o.best_index, o.n_valid, o.n_scored) and abs(res_one[0][i][1] - o.cost) <= 1e-5
    one.close()
    for g in groups:
        g.fl.close()


# ----------------------------------------------------------------------------------------------
# Two cycles in flight on a stream (navgpu_planner_set_cycles_in_flight): cycle k + 1 is handed over (new scan, pose, plan)
# and queued while cycle k runs, cycle k's results are read afterwards.  Every cycle must return exactly what the same
# cycle returns when it is run to completion before the next hand-over; the first cycle is checked against the oracle.
# ----------------------------------------------------------------------------------------------
def test_two_cycles_in_flight_equal_one(nav, orc):
    import bench
    from navigation_amd import synth
    from navigation_amd._lib import NavgpuError as NavGpuError
    n, n_rob, n_cyc = 400, 8, 7
    key = lambda rbuf: [(r.best_index, r.cost, tuple(r.drive), r.n_valid, r.n_scored) for r in rbuf]
    a, insts, cfg = bench.build_fleet(nav, n_rob, n, seed0=640)
    ga = bench.Group(nav, a, insts, seed=3)
    res_a, m_first = [], None
    for k in range(n_cyc):
        ga.cycle(k)
        ga.collect()
        res_a.append(key(ga.rbuf))
        if k == 0:
            m_first = a.master()
    b, insts_b, _ = bench.build_fleet(nav, n_rob, n, seed0=640)
    gb = bench.Group(nav, b, insts_b, seed=3)
    with pytest.raises(NavGpuError):  # one cycle in flight: there is no "previous" slot
        b.results_previous_into(gb.rbuf)
    gb.set_depth(2)
    with pytest.raises(NavGpuError):  # no cycle queued yet
        b.results_previous_into(gb.rbuf)
    res_b = []
    for k in range(n_cyc):
        gb.pending_before = gb.pending
        gb.cycle(k)            # queues cycle k, then reads cycle k - 1 into rbuf
        if gb.pending_before:
            res_b.append(key(gb.rbuf))
    with pytest.raises(NavGpuError):  # a sub-range cycle cannot alternate the result slot
        b.planner_cycle(0, 1)
    gb.collect()               # the last cycle: navgpu_planner_results = the latest queued cycle
    res_b.append(key(gb.rbuf))
    assert res_a == res_b
    assert ga.scored == gb.scored
    assert np.array_equal(a.master(), b.master())
    # back to one cycle in flight: the latest results stay readable, the next cycle behaves as before
    gb.set_depth(1)
    assert key(b.results_into(gb.rbuf)) == res_a[-1]
    ga.cycle(n_cyc)
    ga.collect()
    gb.cycle(n_cyc)
    gb.collect()
    assert key(ga.rbuf) == key(gb.rbuf)
    ocfg = orc.DwaConfig(**cfg.as_dict())
    for i in (0, 3, 7):
        p = orc.DwaPlanner(m_first[i], synth.RES, 0.0, 0.0, ocfg)
        p.set_plan()
        st = ga.states[0][i]
        o, _, _, _, _ = p.cycle(np.array(list(st.pos), np.float32), np.array(list(st.vel), np.float32), insts[i]["plan"], synth.FOOTPRINT)
        assert (res_b[0][i][0], res_b[0][i][3], res_b[0][i][4]) == (o.best_index, o.n_valid, o.n_scored) and abs(res_b[0][i][1] - o.cost) <= 1e-5
    a.close()
    b.close()


# ----------------------------------------------------------------------------------------------
# The floating-point contract's one library dependency: the rollout's cos(theta) / sin(theta)
# (simple_trajectory_generator.cpp:253-258) are the host libm's on the reference and ocml's on the device.  Neither is
# correctly rounded; both are faithful.  Measured here through navgpu_device_sincos: they never differ by more than one
# unit in the last place, and do so for about 3 % of the arguments.  A one-ulp change of a double factor moves the next
# pose - rounded to float - only when the double result lies within ~1e-16 of a float rounding boundary (float spacing:
# 6e-8 relative), i.e. about 2e-9 of the affected steps, which is why every cost / index comparison of this suite holds
# bit for bit.
# ----------------------------------------------------------------------------------------------
def test_device_sincos_against_host_libm(nav):
    import ctypes as C
    import math
    Lb = nav.lib()
    rs = np.random.RandomState(7)
    n = 1 << 17
    th = np.concatenate([rs.uniform(-2 * math.pi, 2 * math.pi, n).astype(np.float32).astype(np.float64),   # rollout headings are floats
                         math.pi / 2 + rs.uniform(-math.pi, math.pi, n // 4).astype(np.float32).astype(np.float64),  # M_PI_2 + theta
                         np.array([0.0, -0.0, math.pi / 2, math.pi, -math.pi, 1e-30, 1e6, -1e6])])
    sn, cs = np.empty_like(th), np.empty_like(th)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    assert Lb.navgpu_device_sincos(0, ptr(th), len(th), ptr(sn), ptr(cs)) == 0
    hs = np.array([math.sin(v) for v in th])
    hc = np.array([math.cos(v) for v in th])
    for dev, host in ((sn, hs), (cs, hc)):
        ulp = np.abs(dev.view(np.int64) - host.view(np.int64))
        assert ulp.max() <= 1, ulp.max()
        assert (ulp != 0).mean() < 0.06
    assert Lb.navgpu_device_sincos(0, None, 4, ptr(sn), ptr(cs)) == -1  # NAVGPU_ERR_INVALID
