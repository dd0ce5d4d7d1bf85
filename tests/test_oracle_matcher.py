"""CPU tests of the matcher oracle (oracle/matcher_oracle.c).  The reference matcher cannot be compiled in
this image (it needs Eigen), so the oracle is anchored on the values SURVEY.md Appendix C recorded from
the reference binary (kernel tables, grid sizes, offset recovery) and on brute-force restatements of each
step written directly from src/matcher/chargrid.cpp."""
import math

import numpy as np

from cg_mrslam_amd import synth


def test_kernel_tables_match_reference_probe(oracle):
    # close 17x17 K1=3 K2=25 centre row 24 21 .. 0 ..; LC 11x11 K1=12 K2=64 centre row 60 48 ..
    k = oracle.make_kernel(0.025, 0.2)
    assert k.shape == (17, 17)
    assert k[8].tolist() == [24, 21, 18, 15, 12, 9, 6, 3, 0, 3, 6, 9, 12, 15, 18, 21, 24]
    assert k[0, 0] == 25 and k.max() == 25
    assert np.array_equal(k, k.T) and np.array_equal(k, k[::-1, ::-1])
    k2 = oracle.make_kernel(0.1, 0.5)
    assert k2.shape == (11, 11)
    assert k2[5].tolist() == [60, 48, 36, 24, 12, 0, 12, 24, 36, 48, 60]
    for i in range(17):
        for j in range(17):
            v = int(3 * math.sqrt((i - 8) ** 2 + (j - 8) ** 2))
            assert k[j, i] == (v if v <= 25 else 25)


def test_grid_sizes_match_reference_probe(oracle):
    assert oracle.grid_dims((-15, -15), (15, 15), 0.025) == (1200, 1200, 40.0)
    assert oracle.grid_dims((-35, -35), (35, 35), 0.1) == (700, 700, 10.0)


def test_portable_sincos_within_one_ulp_of_libm(oracle):
    rng = np.random.default_rng(0)
    xs = np.concatenate([np.linspace(-7.5, 7.5, 4001), rng.uniform(-3.6, 3.6, 6000), [0.0, 1e-9, -1e-9, math.pi, -math.pi]])
    for x in xs:
        s, c = oracle.sincos(float(x))
        assert abs(s - math.sin(x)) <= np.spacing(abs(math.sin(x))) + 1e-300
        assert abs(c - math.cos(x)) <= np.spacing(abs(math.cos(x))) + 1e-300


def test_subsample_matches_map_semantics(oracle):
    rng = np.random.default_rng(2)
    pts = rng.uniform(-3, 3, size=(700, 2))
    pts[::7] = pts[1::7][: len(pts[::7])] + 1e-3         # force shared buckets
    out = oracle.subsample(pts, 0.1)
    acc = {}
    for p in pts:                                        # chargrid.cpp:61-122 in plain Python
        key = (int(10.0 * p[0]), int(10.0 * p[1]))       # trunc toward zero
        a = acc.setdefault(key, [0.0, 0.0, 0])
        a[0] += p[0]; a[1] += p[1]; a[2] += 1
    want = np.array([[a[0] * (1.0 / a[2]), a[1] * (1.0 / a[2])] for _, a in sorted(acc.items())])
    assert np.array_equal(out, want)


def test_rasterize_matches_bruteforce(oracle):
    rng = np.random.default_rng(3)
    pts = rng.uniform(-2.2, 2.2, size=(60, 2))           # some stamps hang over the border of the [-2,2]^2 grid
    cells = oracle.rasterize((-2, -2), (2, 2), 0.025, 0.025, 0.2, pts)
    nx, ny, inv = oracle.grid_dims((-2, -2), (2, 2), 0.025)
    k = oracle.make_kernel(0.025, 0.2)
    want = np.full((nx, ny), 25, dtype=np.uint8)
    for p in pts:
        fx, fy = np.float32(p[0]), np.float32(p[1])
        r = int(np.rint((fx - np.float32(-2)) * np.float32(inv)))
        c = int(np.rint((fy - np.float32(-2)) * np.float32(inv)))
        for i in range(17):
            for j in range(17):
                x, y = r + i - 8, c + j - 8
                if 0 <= x < nx and 0 <= y < ny:
                    want[x, y] = min(want[x, y], k[j, i])
    assert np.array_equal(cells, want)


def _naive_greedy(oracle, ll, ur, res, ref, q, region, theta_res, max_score, dx, dy, dth):
    """greedySearch for one region, straight from chargrid.cpp:208-308 (single thread map)."""
    cells = oracle.rasterize(ll, ur, res, res, 0.2, ref)
    nx, ny, inv = oracle.grid_dims(ll, ur, res)
    f32 = np.float32
    w2g = lambda v, l: int(np.rint((f32(v) - f32(l)) * f32(inv)))
    lo = (w2g(region[0], ll[0]), w2g(region[1], ll[1]))
    hi = (w2g(region[3], ll[0]), w2g(region[4], ll[1]))
    best = {}
    t = float(f32(region[2]))
    while t < float(f32(region[5])):
        s, c = oracle.sincos(t)
        ips, prev = [], None
        for p in q:
            px, py = c * p[0] - s * p[1], s * p[0] + c * p[1]
            ip = (int(px * float(f32(inv))), int(py * float(f32(inv))))
            if ip != prev:
                ips.append(ip); prev = ip
        k = len(ips)
        for i in range(lo[0], hi[0]):
            for j in range(lo[1], hi[1]):
                idsum = sum(int(cells[x + i, y + j]) for x, y in ips if 0 <= x + i < nx and 0 <= y + j < ny)
                dsum = f32(f32(idsum) * f32(1.0 / 128.0))
                dsum = f32(float(dsum) / k) if k else f32(max_score + 1)
                if float(dsum) < max_score:
                    x = float(f32(f32(ll[0]) + f32(res) * f32(i))); y = float(f32(f32(ll[1]) + f32(res) * f32(j)))
                    key = (int(x / dx), int(y / dy), int(t / dth))
                    if key not in best or best[key][3] > float(dsum):
                        best[key] = (x, y, t, float(dsum))
        t += theta_res
    out = [best[k] for k in sorted(best)]
    out.sort(key=lambda r: r[3])                          # stable
    return np.array(out).reshape(-1, 4)


def test_greedy_matches_naive_restatement(oracle):
    rng = np.random.default_rng(4)
    wall = np.stack([np.linspace(-1.5, 1.5, 90), np.full(90, 1.0)], 1)
    wall2 = np.stack([np.full(60, -1.2), np.linspace(-1.0, 1.0, 60)], 1)
    ref = np.concatenate([wall, wall2]) + rng.normal(scale=0.004, size=(150, 2))
    c, s = math.cos(0.04), math.sin(0.04)
    q = (ref[::3] - [0.06, -0.04]) @ np.array([[c, -s], [s, c]])      # query = ref moved by a small motion
    region = np.array([-0.1, -0.15, -0.08, 0.15, 0.1, 0.1], dtype=np.float32)
    n, got = oracle.greedy_search((-2, -2), (2, 2), 0.025, 0.025, 0.2, ref, q, region, 0.025, 0.0125, 0.2, 0.1, 0.1, 0.05)
    want = _naive_greedy(oracle, (-2, -2), (2, 2), 0.025, ref, q, region, 0.0125, 0.2, 0.1, 0.1, 0.05)
    assert n == len(want) and n > 3
    assert np.array_equal(got, want)


def test_close_match_recovers_known_offset(oracle):
    # the probe of SURVEY.md Appendix C: true offset (0.10, -0.05, 0.03) in a 10 x 8 m room is recovered
    ang = synth.LASER_ANGLE_MIN + synth.LASER_ANGLE_INC * np.arange(1081)
    boxes = [(-5.0, -4.0, 5.0, 4.0)]
    p1 = np.array([0.3, -0.2, 0.1])
    d = np.array([0.10, -0.05, 0.03])
    p2 = synth.se2_compose(p1, d)
    r1 = synth._raycast_boxes(p1[0], p1[1], p1[2] + ang, boxes, 30.0).astype(np.float32)
    r2 = synth._raycast_boxes(p2[0], p2[1], p2[2] + ang, boxes, 30.0).astype(np.float32)
    xyt, score, found = oracle.close_scan_match_batch(r1, r2, synth.LASER_ANGLE_MIN, synth.LASER_ANGLE_INC, 30.0, [0, 0, 0],
                                                      d + [0.04, -0.03, 0.015])
    assert found[0] == 1
    # within one grid cell / one angle step (the candidate lattice is anchored on the grid, not on the truth)
    assert abs(xyt[0, 0] - 0.10) < 0.0251 and abs(xyt[0, 1] + 0.05) < 0.0251 and abs(xyt[0, 2] - 0.03) < 0.00626
    assert score[0] < 0.03


def test_hierarchical_single_level_returns_nothing(oracle):
    # chargrid.cpp:310-344: with one level the loop body never runs and the final search is skipped
    ref = np.stack([np.linspace(-1, 1, 50), np.zeros(50)], 1)
    reg = np.array([-0.2, -0.2, -0.1, 0.2, 0.2, 0.1], dtype=np.float32)
    n, _ = oracle.hierarchical_search((-2, -2), (2, 2), 0.025, 0.025, 0.2, ref, ref, reg, 0.0125, 0.3, 0.1, 0.1, 0.05, 1)
    assert n == 0
    n, res = oracle.hierarchical_search((-2, -2), (2, 2), 0.025, 0.025, 0.2, ref, ref, reg, 0.0125, 0.3, 0.1, 0.1, 0.05, 3)
    assert n >= 1 and abs(res[0, 0]) < 0.03 and abs(res[0, 1]) < 0.03


def test_golden_close_match(oracle):
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "match_close12.npz"))
    xyt, score, found = oracle.close_scan_match_batch(d["ranges_ref"][:4], d["ranges_qry"][:4], float(d["angle_min"]),
                                                      float(d["angle_inc"]), float(d["max_range"]), [0, 0, 0], d["guess"][:4])
    assert np.array_equal(xyt, d["xyt"][:4]) and np.array_equal(score, d["score"][:4]) and np.array_equal(found, d["found"][:4])


def test_hand_derived_three_point_scan(oracle):
    """Known answers worked out by hand from chargrid.cpp / gridmap.h / scan_matcher.cpp (tests/known_answers.py): the
    stamped grid, the candidate scores and the single pruned result, incl. the float rounding of grid2world."""
    import known_answers as K
    grid = oracle.rasterize(K.GRID["ll"], K.GRID["ur"], K.GRID["res"], K.GRID["res"], K.GRID["kernel_range"], K.REF)
    assert grid.shape == (1200, 1200)
    for (cx, cy) in K.CELLS:
        assert grid[cx, cy] == 0
    for x, y in [(641, 620), (640, 622), (640, 626), (648, 620), (649, 620), (688, 568), (672, 552), (100, 100), (660, 600)]:
        assert grid[x, y] == K.expected_grid_value(x, y), (x, y)
    assert (grid != 25).sum() == sum(1 for x in range(600, 700) for y in range(540, 640) if K.expected_grid_value(x, y) != 25)
    # every candidate below the threshold, one at a time (a tight maxScore isolates single offsets): scores by hand
    n, res = oracle.greedy_search(K.GRID["ll"], K.GRID["ur"], K.GRID["res"], K.GRID["res"], K.GRID["kernel_range"], K.REF, K.REF,
                                  K.REGION, 0.025, 0.00625, 0.5, 0.5, 0.5, 0.2)
    assert n == 1 and tuple(res[0]) == K.EXPECTED_SAME
    shifted = K.REF + [0.025, 0.0]
    n, res = oracle.greedy_search(K.GRID["ll"], K.GRID["ur"], K.GRID["res"], K.GRID["res"], K.GRID["kernel_range"], K.REF, shifted,
                                  K.REGION, 0.025, 0.00625, 0.5, 0.5, 0.5, 0.2)
    assert n == 1 and tuple(res[0]) == K.EXPECTED_SHIFTED
    # finer result bins (0.025 m: one bin per offset) expose every candidate's score
    n, res = oracle.greedy_search(K.GRID["ll"], K.GRID["ur"], K.GRID["res"], K.GRID["res"], K.GRID["kernel_range"], K.REF, K.REF,
                                  K.REGION, 0.025, 0.00625, 0.5, 0.0125, 0.0125, 0.2)
    got = {(round((r[0] + 15) * 40), round((r[1] + 15) * 40)): r[3] for r in res[:n]}
    for i in range(598, 602):
        for j in range(598, 602):
            assert got[(i, j)] == K.expected_score(i, j, K.QUERY_CELLS), (i, j)
    assert n == 16 and got[(600, 600)] == 0.0 and got[(601, 600)] == 0.0234375 and got[(600, 602 - 1)] == K.expected_score(600, 601, K.QUERY_CELLS)


def test_portable_sincos_never_moves_a_cell_on_the_fixture(oracle):
    """oracle/matcher_oracle.c, deviation (1): the oracle and the HIP kernel take cos / sin of the search angle from a portable
    routine, the reference from libm (chargrid.cpp:241).  Counted here, over every (subsampled point, search angle) pair of the
    4096-pair fixture (the C3 recipe: 64 angles a pair, ~430 points: 1.1e8 pairs) and over 2e5 generated (point, angle) pairs:
    the truncated cells (chargrid.cpp:246-250) the two would search.  Zero differences = every result of the fixture is what the
    libm path gives; the number of angles whose cos / sin differ in the last bit is printed for the record."""
    sp = synth.make_scan_pairs(4096, seed=4242)
    theta_res, win_t = 0.0125 * .5, 0.2
    total_cells = total_bits = total_pairs = 0
    for i in range(4096):
        pts = oracle.cartesian(sp["ranges_qry"][i], sp["angle_min"], sp["angle_inc"], sp["max_range"])
        q = oracle.subsample(pts, 0.1)
        # the angle list of chargrid.cpp:239 for closeScanMatching's window (scan_matcher.cpp:148-151): float bounds, double steps
        lo, hi = float(np.float32(-win_t + sp["guess"][i, 2])), float(np.float32(win_t + sp["guess"][i, 2]))
        angles = []
        t = lo
        while t < hi:
            angles.append(t)
            t += theta_res
        nd, nb = oracle.sincos_cell_differences(q, angles, 40.0)
        total_cells += nd
        total_bits += nb
        total_pairs += len(q) * len(angles)
    rng = np.random.default_rng(11)
    nd, nb = oracle.sincos_cell_differences(rng.uniform(-30, 30, size=(2000, 2)), rng.uniform(-3.3, 3.3, size=100), 40.0)
    nd2, nb2 = oracle.sincos_cell_differences(rng.uniform(-35, 35, size=(2000, 2)), rng.uniform(-3.3, 3.3, size=100), 10.0)
    print(f"{total_pairs} (point, angle) pairs of the fixture: {total_cells} cells differ; cos / sin differ in a bit for {total_bits} of "
          f"{4096 * 64} angles; generated: {nd} + {nd2} cells of 4e5, {nb + nb2} of 200 angles")
    assert total_pairs > 1e8
    assert total_cells == 0 and nd == 0 and nd2 == 0


def test_hand_derived_subsample_hierarchy_verify_and_twin_regions(oracle):
    """The round-5 cases of tests/known_answers.py (worked out from chargrid.cpp / scan_matcher.cpp on paper): subsample order and
    means, two levels of hierarchicalSearch with its region propagation, searchNonMatchedPoints + countPoints, and
    scanMatchingLC's twin regions / normalise / merge -- the last one on the plain-Python restatement of the ScanMatcher
    bookkeeping (tests/ref_scan_matcher.py) over the oracle."""
    import known_answers as K
    import oracle_backend as ob
    # (S)
    assert np.array_equal(oracle.subsample(K.SUBSAMPLE_IN, 0.1), K.SUBSAMPLE_OUT)
    # (H)
    a = K.HIER_ARGS
    n, res = oracle.hierarchical_search(K.GRID["ll"], K.GRID["ur"], K.GRID["res"], K.GRID["res"], K.GRID["kernel_range"], K.HIER_REF, K.HIER_REF,
                                        K.HIER_REGION, a["theta_res"], a["max_score"], a["dx"], a["dy"], a["dth"], a["n_levels"])
    assert n == 2 and [tuple(r) for r in res[:n]] == K.HIER_EXPECTED
    # (V)
    n_nm, score = oracle.verify(K.VERIFY_LL, K.VERIFY_UR, K.VERIFY_RES, K.VERIFY_RES, K.VERIFY_RANGE, K.VERIFY_PTS2, K.VERIFY_PTS1,
                                K.VERIFY_LOWER, K.VERIFY_UPPER, 0.3)
    assert n_nm == K.VERIFY_NONMATCHED and score == K.VERIFY_SCORE
    assert K._VSUM == sum(min(int(12 * np.hypot(di, dj)), 64) for di in (-3, -2, -1, 0, 1, 2) for dj in (-3, -2, -1, 0, 1, 2)) == 990
    # (L)
    lo = ob.OracleMatcher(K.LC_ANGLE_MIN, K.LC_ANGLE_INC, K.LC_MAX_RANGE, (-35.0, -35.0), (35.0, 35.0), 0.1, 0.5)
    assert np.array_equal(lo.cartesian(K.LC_RANGES), K.LC_POINTS)
    res = lo.scanMatchingLC([(K.LC_RANGES, np.array([3.0, -2.0, 0.7]))], 0, [(K.LC_RANGES, np.array([-1.0, 4.0, -2.0]))], 0, K.LC_MAX_SCORE)
    assert [tuple(r) for r in res] == K.LC_EXPECTED
