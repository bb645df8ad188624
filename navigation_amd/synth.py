"""Synthetic fleet workload: seeded maps, poses, plans and LaserScan-like clouds.

Pure numpy data generation shared by bench.py and the tests; no compute of the hot path here.
  map      : N x N uint8 @ 0.05 m, Bernoulli(0.004) lethal cells outside a 1.5 m disc around the robot
             plus two axis-aligned wall segments, seed 1000 + instance
  robot    : map centre, yaw U(-pi, pi), vel (U(0,0.5), 0, U(-0.5,0.5))
  footprint: square, half-width 0.2 m
  plan     : 200 poses, x = cx + 0.04 i, y = cy + 0.5 sin(0.03 i)
  scan     : 720 beams over 270 deg, ray-cast against the static cells and 3 moving discs
             (r = 0.2 m), max range 10 m, sensor at the robot origin z = 0.3 -> float32 xyz
"""
import math

import numpy as np

RES = 0.05
FOOTPRINT = np.array([[0.2, 0.2], [0.2, -0.2], [-0.2, -0.2], [-0.2, 0.2]], np.float64)
FOOTPRINT5 = np.array([[-0.325, -0.325], [-0.325, 0.325], [0.325, 0.325], [0.46, 0.0], [0.325, -0.325]], np.float64)
INFLATION_RADIUS, COST_SCALING = 0.55, 10.0


def inscribed_radius(fp):
    """costmap_2d::calculateMinAndMaxDistances min_dist (footprint.cpp:41-67), numpy restatement."""
    fp = np.asarray(fp, np.float64)
    best = float("inf")
    n = len(fp)
    for i in range(n):
        x0, y0 = fp[i]
        x1, y1 = fp[(i + 1) % n]
        vd = math.hypot(x0, y0)
        c, d = x1 - x0, y1 - y0
        param = ((-x0) * c + (-y0) * d) / (c * c + d * d)
        if param < 0:
            xx, yy = x0, y0
        elif param > 1:
            xx, yy = x1, y1
        else:
            xx, yy = x0 + param * c, y0 + param * d
        best = min(best, vd, math.hypot(xx, yy))
    return best


def make_instance(n_cells, instance, density=0.004):
    """Static lethal cells + robot state + plan for one robot instance."""
    rs = np.random.RandomState(1000 + instance)
    size = n_cells * RES
    cx = cy = size / 2.0
    cells = np.zeros((n_cells, n_cells), np.uint8)
    lethal = rs.random_sample((n_cells, n_cells)) < density
    yy, xx = np.mgrid[0:n_cells, 0:n_cells]
    wx = (xx + 0.5) * RES
    wy = (yy + 0.5) * RES
    keep_out = (wx - cx) ** 2 + (wy - cy) ** 2 < 1.5 ** 2
    lethal &= ~keep_out
    # two wall segments, axis aligned, away from the robot
    wlen = int(n_cells * 0.3)
    r0 = int(n_cells * (0.15 + 0.1 * rs.random_sample()))
    c0 = int(n_cells * (0.1 + 0.5 * rs.random_sample()))
    lethal[r0, c0:c0 + wlen] = True
    c1 = int(n_cells * (0.8 + 0.1 * rs.random_sample()))
    r1 = int(n_cells * (0.1 + 0.5 * rs.random_sample()))
    lethal[r1:r1 + wlen, c1] = True
    lethal &= ~keep_out
    cells[lethal] = 254
    yaw = rs.uniform(-math.pi, math.pi)
    vel = np.array([rs.uniform(0.0, 0.5), 0.0, rs.uniform(-0.5, 0.5)], np.float32)
    pos = np.array([cx, cy, yaw], np.float32)
    i = np.arange(200)
    plan = np.stack([cx + 0.04 * i, cy + 0.5 * np.sin(0.03 * i)], axis=1)
    discs = np.concatenate([rs.uniform(cx - 2.5, cx + 2.5, (3, 2)), rs.uniform(-0.3, 0.3, (3, 2))], axis=1)
    return dict(cells=cells, pos=pos, vel=vel, plan=plan, discs=discs, origin=np.zeros(2), size=size)


def laser_scan(inst, cycle=0, n_beams=720, fov=math.radians(270.0), max_range=10.0, z=0.3, z_jitter=None):
    """Ray-cast a planar scan from the robot against static lethal cells and the moving discs.
    Returns float32 (k, 3) hit points in the global frame (beams with no return are dropped,
    like laser_geometry does for out-of-range readings)."""
    cells = inst["cells"]
    n = cells.shape[0]
    x0, y0, yaw = [float(v) for v in inst["pos"]]
    ang = yaw + np.linspace(-fov / 2, fov / 2, n_beams)
    step = RES * 0.5
    r = np.arange(step, max_range, step)
    px = x0 + np.outer(np.cos(ang), r)
    py = y0 + np.outer(np.sin(ang), r)
    ix = np.floor(px / RES).astype(np.int64)
    iy = np.floor(py / RES).astype(np.int64)
    inside = (ix >= 0) & (iy >= 0) & (ix < n) & (iy < n)
    hit = np.zeros_like(inside)
    hit[inside] = cells[iy[inside], ix[inside]] == 254
    discs = inst["discs"]
    for d in discs:
        dx = d[0] + d[2] * 0.2 * cycle
        dy = d[1] + d[3] * 0.2 * cycle
        hit |= (px - dx) ** 2 + (py - dy) ** 2 < 0.2 ** 2
    first = np.argmax(hit, axis=1)
    has = hit[np.arange(n_beams), first]
    rr = r[first][has]
    a = ang[has]
    pts = np.stack([x0 + rr * np.cos(a), y0 + rr * np.sin(a), np.full(rr.shape, z)], axis=1)
    if z_jitter is not None:
        rs = np.random.RandomState(77 + cycle)
        pts[:, 2] = z + rs.uniform(0.0, z_jitter, len(pts))
    return pts.astype(np.float32)


def fleet_config(vx=32, vy=32, vth=16, sim_time=2.0, sim_granularity=0.1):
    """DWA parameters of the benchmark configs: reference defaults except the sample counts and
    discretize_by_time with T = sim_time / sim_granularity = 20 steps."""
    from ._lib import DwaConfig
    return DwaConfig(vx_samples=vx, vy_samples=vy, vth_samples=vth, sim_time=sim_time, sim_granularity=sim_granularity,
                     discretize_by_time=1)
