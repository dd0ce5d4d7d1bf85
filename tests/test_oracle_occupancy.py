"""The occupancy-map oracle (oracle/occupancy_oracle.c) against an independent plain-Python restatement of the
reference's ray casting (frequency_map.cpp:27-103, grid_line_traversal.cpp:31-154) and known answers."""
import math

import numpy as np
import pytest

from cg_mrslam_amd import synth


def py_grid_line(sx, sy, ex, ey):
    """Plain restatement of gridLineCore written from the reference's structure: step along the major axis from the
    end with the smaller major coordinate, Bresenham error term decides the minor step."""
    dx, dy = abs(ex - sx), abs(ey - sy)
    pts = []
    if dy <= dx:
        d, i1, i2 = 2 * dy - dx, 2 * dy, 2 * (dy - dx)
        if sx > ex:
            x, y, flag, xend = ex, ey, -1, sx
        else:
            x, y, flag, xend = sx, sy, 1, ex
        pts.append((x, y))
        inc = 1 if (ey - sy) * flag > 0 else -1
        while x < xend:
            x += 1
            if d < 0:
                d += i1
            else:
                y += inc
                d += i2
            pts.append((x, y))
    else:
        d, i1, i2 = 2 * dx - dy, 2 * dx, 2 * (dx - dy)
        if sy > ey:
            y, x, flag, yend = ey, ex, -1, sy
        else:
            y, x, flag, yend = sy, sx, 1, ey
        pts.append((x, y))
        inc = 1 if (ex - sx) * flag > 0 else -1
        while y < yend:
            y += 1
            if d < 0:
                d += i1
            else:
                x += inc
                d += i2
            pts.append((x, y))
    return pts


def test_grid_line_vs_python_and_endpoints(oracle):
    rng = np.random.default_rng(3)
    cases = [(0, 0, 0, 0), (0, 0, 5, 0), (0, 0, 0, -7), (3, 3, -4, -4), (2, 9, 11, 4), (-5, 2, -1, 30), (10, 10, 3, 12)]
    cases += [tuple(int(v) for v in rng.integers(-60, 60, size=4)) for _ in range(300)]
    for sx, sy, ex, ey in cases:
        got = [tuple(p) for p in oracle.grid_line(sx, sy, ex, ey)]
        assert got == py_grid_line(sx, sy, ex, ey)
        assert len(got) == max(abs(ex - sx), abs(ey - sy)) + 1
        assert (sx, sy) in (got[0], got[-1]) and (ex, ey) in (got[0], got[-1])       # both end points, one per end
        steps = np.abs(np.diff(np.array(got), axis=0))
        assert steps.max(initial=0) <= 1                                              # 8-connected


def py_integrate(rows, cols, res, off, ranges, pose, a0, da, laser_max, max_range, usable, inf_fill, gain, sq):
    res, ox, oy = np.float32(res), np.float32(off[0]), np.float32(off[1])
    hits = np.zeros((rows, cols), dtype=np.int64)
    misses = np.zeros((rows, cols), dtype=np.int64)
    if max_range < 0:
        max_range = np.float32(laser_max)
    if usable < 0:
        usable = max_range
    w2m = lambda w, o: int(np.rint((np.float32(w) - o) / res))            # noqa: E731  float arithmetic, half-even
    inside = lambda x, y: 0 <= x < rows and 0 <= y < cols                  # noqa: E731
    cl, sl = math.cos(pose[2]), math.sin(pose[2])
    start = (w2m(pose[0], ox), w2m(pose[1], oy))
    for i, r in enumerate(ranges):
        r = np.float32(r)
        cropped = False
        if r > np.float32(usable):
            r, cropped = np.float32(usable), True
        if r >= np.float32(max_range) or r <= 0:
            if inf_fill > 0:
                r, cropped = np.float32(inf_fill), True
            else:
                continue
        ang = np.float32(a0 + i * da)
        bx, by = float(r * np.float32(np.cos(ang))), float(r * np.float32(np.sin(ang)))
        wx, wy = (cl * bx - sl * by) + pose[0], (sl * bx + cl * by) + pose[1]
        end = (w2m(wx, ox), w2m(wy, oy))
        for (x, y) in py_grid_line(*start, *end):
            if inside(x, y):
                misses[x, y] += 1
        if not inside(*end) or cropped:
            continue
        for c in range(-sq, sq + 1):
            for q in range(-sq, sq + 1):
                if inside(end[0] + q, end[1] + c):
                    hits[end[0] + q, end[1] + c] += gain
    rg = (w2m(pose[0], ox), w2m(pose[1], oy))
    for c in range(-4, 5):
        for q in range(-4, 5):
            if inside(rg[0] + q, rg[1] + c):
                misses[rg[0] + q, rg[1] + c] += 1
    return hits, misses


@pytest.mark.parametrize("inf_fill,usable,sq", [(5.0, -1.0, 0), (-1.0, 6.0, 1)])
def test_integrate_scan_vs_python(oracle, inf_fill, usable, sq):
    """np.cos / np.sin on float32 arrays call the same libm cosf / sinf the oracle calls."""
    tr = synth.make_trajectory(8, laps=0.02, n_beams=181)
    a0, da = -1.5, 3.0 / 180
    rng = np.random.default_rng(1)
    for k in (0, 5):
        pose = tr["truth"][k] + np.array([0.3, -0.2, 0.4])
        ranges = tr["scans"][k].copy()
        ranges[rng.integers(0, 181, size=12)] = 40.0                     # beyond the maximum range
        ranges[7] = 0.0
        res, off, rows, cols = 0.05, (pose[0] - 9.0, pose[1] - 7.5), 330, 310
        h, m = oracle.occupancy_integrate(rows, cols, res, off, ranges[None], pose[None], a0, da, 30.0, max_range=-1.0,
                                          usable_range=usable, infinity_filling_range=inf_fill, gain=3, square_size=sq)
        h2, m2 = py_integrate(rows, cols, res, off, ranges, pose, a0, da, 30.0, -1.0, usable, inf_fill, 3, sq)
        np.testing.assert_array_equal(h, h2)
        np.testing.assert_array_equal(m, m2)
        assert h.sum() > 0 and m.sum() > 181


def test_image_thresholds(oracle):
    hits = np.array([[0, 0, 3, 1, 7]], dtype=np.int32)
    misses = np.array([[0, 5, 1, 4, 13]], dtype=np.int32)          # fractions: -, 0, .75, .2, .35
    img = oracle.occupancy_image(hits, misses, 0.65, 0.196)
    np.testing.assert_array_equal(img, [[255, 0, 100, 255, 255]])
