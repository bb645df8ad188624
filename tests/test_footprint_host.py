"""CPU tests of the pure host footprint helpers of the C-ABI (costmap_2d/src/footprint.cpp:41-67,138-167) against the
oracle's restatement and the reference's formulas."""
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


def _radii(L, fp):
    fp = np.ascontiguousarray(fp, np.float64).reshape(-1, 2)
    a, b = C.c_double(), C.c_double()
    assert L.navgpu_footprint_radii(fp.ctypes.data if len(fp) else None, len(fp), C.byref(a), C.byref(b)) == 0
    return a.value, b.value


def test_footprint_radii_match_oracle(L):
    from oracle import pyoracle as orc
    rs = np.random.RandomState(4)
    cases = [[[0.2, 0.2], [0.2, -0.2], [-0.2, -0.2], [-0.2, 0.2]],                      # square
             [[0.325, 0.325], [0.325, -0.325], [-0.325, -0.325], [-0.325, 0.325], [0.465, 0.0]][::-1],  # costmap_params.yaml-like
             [[1.0, 0.0], [0.0, 1.0], [-1.0, 0.0]]]
    for _ in range(200):
        n = int(rs.randint(3, 12))
        ang = np.sort(rs.uniform(0, 2 * math.pi, n))
        r = rs.uniform(0.05, 1.0, n)
        cases.append(np.stack([r * np.cos(ang), r * np.sin(ang)], 1).tolist())
    for fp in cases:
        got = _radii(L, fp)
        want = orc.min_max_distances(fp)
        assert got == tuple(want), (fp, got, want)
    # degenerate footprints: nothing to measure (footprint.cpp:46-49)
    for fp in ([], [[0.1, 0.0]], [[0.1, 0.0], [-0.1, 0.0]]):
        mn, mx = _radii(L, fp)
        assert mn == sys.float_info.max and mx == 0.0


def test_footprint_pad_and_circle(L):
    fp = np.array([[0.2, 0.1], [0.0, -0.3], [-0.25, 0.0], [-0.0, 0.0]], np.float64)
    want = fp + np.sign(fp) * 0.05  # sign0: 0 stays 0 (costmap_math.h:53-56)
    got = fp.copy()
    assert L.navgpu_footprint_pad(got.ctypes.data, len(got), 0.05) == 0
    assert np.array_equal(got, want)
    out = np.zeros((16, 2))
    assert L.navgpu_footprint_from_radius(0.46, out.ctypes.data) == 0
    for i in range(16):
        angle = i * 2 * math.pi / 16
        assert out[i, 0] == math.cos(angle) * 0.46 and out[i, 1] == math.sin(angle) * 0.46
    mn, mx = _radii(L, out)
    assert abs(mx - 0.46) < 1e-12 and 0.45 < mn < 0.46


def test_reference_footprint_test_expectations(L):
    """costmap_2d/test/footprint_tests.cpp: what the reference's own tests expect of padFootprint and
    makeFootprintFromRadius (the parameter-server parsing around them is ROS plumbing, out of scope)."""
    # padded_footprint_from_string_param (:64-81): footprint [[1, 1], [-1, 1], [-1, -1]], footprint_padding 0.5
    fp = np.array([[1.0, 1.0], [-1.0, 1.0], [-1.0, -1.0]], np.float64)
    assert L.navgpu_footprint_pad(fp.ctypes.data, len(fp), 0.5) == 0
    assert fp.astype(np.float32).tolist() == [[1.5, 1.5], [-1.5, 1.5], [-1.5, -1.5]]
    # unpadded_footprint_from_string_param (:45-62): padding 0 leaves it alone
    fp0 = np.array([[1.0, 1.0], [-1.0, 1.0], [-1.0, -1.0]], np.float64)
    assert L.navgpu_footprint_pad(fp0.ctypes.data, len(fp0), 0.0) == 0
    assert fp0.tolist() == [[1.0, 1.0], [-1.0, 1.0], [-1.0, -1.0]]
    # radius_param (:83-99): robot_radius 10 -> 16 points, the first (10, 0), the fifth a quarter turn on: (~0, 10)
    out = np.zeros((16, 2))
    assert L.navgpu_footprint_from_radius(10.0, out.ctypes.data) == 0
    assert np.float32(out[0, 0]) == np.float32(10.0) and np.float32(out[0, 1]) == np.float32(0.0)
    assert abs(out[4, 0]) < 1e-4 and abs(out[4, 1] - 10.0) < 1e-4
    # footprint_from_xmlrpc_param (:101-122): [[0.1, 0.1], [-0.1, 0.1], [-0.1, -0.1], [0.1, -0.1]] comes back as given
    fpx = np.array([[0.1, 0.1], [-0.1, 0.1], [-0.1, -0.1], [0.1, -0.1]], np.float64)
    assert L.navgpu_footprint_pad(fpx.ctypes.data, len(fpx), 0.0) == 0
    assert fpx.astype(np.float32).tolist() == np.array([[0.1, 0.1], [-0.1, 0.1], [-0.1, -0.1], [0.1, -0.1]], np.float32).tolist()
