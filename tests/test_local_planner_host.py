"""CPU tests of the host-side DWAPlannerROS mirror's pure functions (no GPU, no fleet): transformGlobalPlan +
prunePlan window and the angle helpers, against oracle/local_planner_oracle.py (goal_functions.cpp:69-174)."""
import ctypes as C
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def L():
    import navigation_amd as nav
    if not os.path.exists(nav.lib_path()):
        nav.build()
    return nav.lib()


def _window(L, plan, pose, T, thr, prune, capacity=None):
    plan = np.ascontiguousarray(plan, np.float64).reshape(-1, 3)
    pose = np.ascontiguousarray(pose, np.float64)
    cap = len(plan) + 1 if capacity is None else capacity
    out = np.zeros((max(cap, 1), 3))
    n, e = C.c_uint32(), C.c_uint32()
    Tp = None if T is None else np.ascontiguousarray(T, np.float64)
    rc = L.navgpu_local_plan_window(plan.ctypes.data, len(plan), pose.ctypes.data, None if T is None else Tp.ctypes.data,
                                    thr, int(prune), out.ctypes.data, cap, C.byref(n), C.byref(e))
    return rc, out[:n.value].copy(), e.value


def _oracle_window(plan, pose, T, thr, prune):
    from oracle import local_planner_oracle as lpo
    gp = [tuple(p) for p in np.asarray(plan, np.float64).reshape(-1, 3)]
    n0 = len(gp)
    loc = lpo.transform_global_plan(gp, pose, T, thr)
    if loc is None:
        return None, 0
    if prune:
        lpo.prune_plan(pose, loc, gp)
    return np.asarray(loc, np.float64).reshape(-1, 3), n0 - len(gp)


def test_plan_window_matches_reference_restatement(L):
    rs = np.random.RandomState(7)
    for case in range(300):
        n = int(rs.randint(1, 120))
        t = np.sort(rs.uniform(0, 12, n))
        plan = np.stack([t + rs.normal(0, 0.05, n), 0.7 * np.sin(t) + rs.normal(0, 0.05, n), rs.uniform(-4, 4, n)], 1)
        T = None if case % 3 == 0 else (rs.uniform(-2, 2), rs.uniform(-2, 2), rs.uniform(-3.5, 3.5))
        pose = np.array([rs.uniform(-1, 13), rs.uniform(-2, 2), rs.uniform(-3.2, 3.2)])
        thr = rs.choice([0.5, 2.0, 5.0, 10.0])
        prune = case % 2
        rc, got, erased = _window(L, plan, pose, T, thr, prune)
        want, werased = _oracle_window(plan, pose, T, thr, prune)
        assert rc == 0
        assert got.shape == want.shape and np.array_equal(got, want), (case, got.shape, want.shape)
        assert erased == werased


def test_plan_window_edge_cases(L):
    plan = np.stack([np.linspace(0, 10, 101), np.zeros(101), np.zeros(101)], 1)
    # empty plan: "Received plan with zero length" -> error, nothing written
    rc, got, erased = _window(L, np.zeros((0, 3)), [0, 0, 0], None, 2.0, 1, capacity=4)
    assert rc == -1 and len(got) == 0
    # robot far from every pose: empty local plan, nothing erased
    rc, got, erased = _window(L, plan, [50.0, 50.0, 0.0], None, 2.0, 1)
    assert rc == 0 and len(got) == 0 and erased == 0
    # the pose that leaves the reach is still taken (goal_functions.cpp:140-154)
    rc, got, _ = _window(L, plan, [0.0, 0.0, 0.0], None, 2.0, 0)
    assert rc == 0 and len(got) == 22 and got[-1][0] == pytest.approx(2.1)
    # prunePlan can erase the whole local plan when nothing is within 1 m (robot 1.5 m beside the path)
    rc, got, erased = _window(L, plan, [5.0, 1.5, 0.0], None, 2.0, 1)
    want, werased = _oracle_window(plan, [5.0, 1.5, 0.0], None, 2.0, 1)
    assert len(got) == len(want) == 0 and erased == werased > 0
    # capacity
    rc, _, _ = _window(L, plan, [5.0, 0.0, 0.0], None, 2.0, 0, capacity=5)
    assert rc == -4


def test_angle_helpers(L):
    from oracle import local_planner_oracle as lpo
    rs = np.random.RandomState(3)
    vals = list(rs.uniform(-20, 20, 500)) + [0.0, math.pi, -math.pi, 2 * math.pi, -2 * math.pi, 3 * math.pi, 1e-300]
    for a in vals:
        for b in (0.0, 1.0, -2.5, math.pi, float(rs.uniform(-10, 10))):
            got = L.navgpu_shortest_angular_distance(a, b)
            assert got == lpo.shortest_angular_distance(a, b)
            assert -math.pi <= got <= math.pi
