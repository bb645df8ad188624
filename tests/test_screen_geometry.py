"""The geometric claim behind the obstacle screens' structuring element (navgpu_host.cpp planWindow, PlannerDev::fp_halfw): every cell
CostmapModel::footprintCost looks at - the LineIterator cells between the vertex cells (costmap_model.cpp:74-119, line_iterator.h) - lies
within r + 1.803 cells (Euclidean) and ceil(r) + 1 cells (Chebyshev) of the cell of the robot's centre, r = the footprint's largest vertex
distance in cells.  Checked on the CPU with the oracle's LineIterator (pinned by the reference's own header compiled in place,
tests/test_oracle_reference_fixtures.py): random polygons, poses and headings, and the worst case the bound is made of."""
import math

import numpy as np
import pytest

from oracle import pyoracle as orc


def _halfw(r_cells):
    """planWindow's table, restated."""
    rc = int(math.ceil(r_cells)) + 1
    rho = r_cells + 1.803 + 0.01
    hw = {}
    for d in range(0, rc + 1):
        if d <= rho:
            hw[d] = min(rc, int(math.floor(math.sqrt(rho * rho - d * d))))
    return rc, hw


def _outline_offsets(fp, res, x, y, th):
    """Cells of the outline relative to the centre cell, as CostmapModel::footprintCost walks them (world_model.h:72-73 rotation,
    Costmap2D::worldToMap's cast, one LineIterator per edge, the last vertex back to the first)."""
    cs, sn = math.cos(th), math.sin(th)
    verts = [(int((x + (fx * cs - fy * sn)) / res), int((y + (fx * sn + fy * cs)) / res)) for fx, fy in fp]
    cx, cy = int(x / res), int(y / res)
    out = set()
    for a in range(len(verts)):
        b = (a + 1) % len(verts)
        for (qx, qy) in orc.line_cells(verts[a][0], verts[a][1], verts[b][0], verts[b][1]):
            out.add((int(qx) - cx, int(qy) - cy))
    return out


@pytest.mark.parametrize("seed", range(6))
def test_outline_cells_lie_in_the_screen_disc(seed):
    rs = np.random.RandomState(100 + seed)
    res = 0.05
    worst = 0.0
    for _ in range(120):
        nv = rs.randint(3, 8)
        ang = np.sort(rs.uniform(0, 2 * math.pi, nv))
        rad = rs.uniform(0.08, 0.75, nv)
        fp = [(float(r * math.cos(a)), float(r * math.sin(a))) for r, a in zip(rad, ang)]
        r_cells = max(math.hypot(fx, fy) for fx, fy in fp) / res
        rc, hw = _halfw(r_cells)
        for _ in range(40):
            # poses well inside the map (worldToMap's cast = floor there), with fractions that hug cell borders as often as not
            x = (rs.randint(40, 400) + rs.choice([rs.uniform(0, 1), 1e-9, 1 - 1e-9, 0.5])) * res
            y = (rs.randint(40, 400) + rs.choice([rs.uniform(0, 1), 1e-9, 1 - 1e-9, 0.5])) * res
            th = rs.uniform(-math.pi, math.pi)
            for (dx, dy) in _outline_offsets(fp, res, x, y, th):
                assert max(abs(dx), abs(dy)) <= rc, (fp, x, y, th, dx, dy)
                assert abs(dy) in hw and abs(dx) <= hw[abs(dy)], (fp, x, y, th, dx, dy, r_cells)
                worst = max(worst, math.hypot(dx, dy) - r_cells)
    assert worst < 1.803
    assert worst > 0.9  # (the margin is not slack: cells beyond r + 0.9 do occur)


def test_the_bound_is_nearly_attained():
    """A vertex at distance r on the diagonal, the pose just short of the next cell in x and y: the vertex cell is (1, 1) cells further
    from the centre cell than the vertex from the centre, and an edge leaving it along a diagonal puts a cell half a step beyond."""
    res = 0.05
    best = 0.0
    for r_cells in (5.657, 7.3, 9.2):
        r = r_cells * res
        d = r / math.sqrt(2.0)
        for eps in (1e-6, 1e-3):
            for fp in ([(d, d), (-0.1, 0.3), (0.3, -0.1)], [(d, d), (d - 0.3, d + 0.25), (-0.2, -0.2)]):
                x = y = (100 + eps) * res
                rr = max(math.hypot(fx, fy) for fx, fy in fp) / res
                for (dx, dy) in _outline_offsets(fp, res, x - 2 * eps * res, y - 2 * eps * res, 0.0) | _outline_offsets(fp, res, x, y, 0.0):
                    best = max(best, math.hypot(dx, dy) - rr)
    assert 1.0 < best < 1.803
