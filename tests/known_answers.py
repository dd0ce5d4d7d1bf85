"""Hand-derivable known answers of the correlative matcher, worked out from the reference source itself (not from any
implementation): a 3-point scan on the close matcher's grid.

Set-up (src/slam/graph_slam.cpp:58-59, src/srslam.cpp:83-84): grid [-15, 15]^2 at 0.025 m (1200 x 1200 cells, invRes
40), kernel 17 x 17 with K1 = int(0.025 * 128) = 3, K2 = int(0.2 * 128) = 25: ker(di, dj) = min(int(3 * sqrt(di^2 +
dj^2)), 25) (scan_matcher.cpp:38-61).

Reference points (already in the reference frame): A = (1.0, 0.5), B = (1.0, 0.6), C = (2.0, -1.0).
addAndConvolvePoints (chargrid.h:205-216, gridmap.h:24-33): cell = lrint((float(p) - (-15)) * 40) ->
A (640, 620), B (640, 624), C (680, 560); every cell within 8 of a point gets min(grid, ker) (chargrid.cpp:132-161).

greedySearch (chargrid.cpp:208-308) with the window [-0.05, 0.05)^2, theta from 0 while < 0.005f in steps of 0.00625 (one
angle: 0; an upper bound of 0.00625f would admit a second one, the float 0.00625f being larger than the double 0.00625)
and the points themselves as the query: offsets i, j in [lrint(14.95 * 40), lrint(15.05 * 40)) = [598, 602); ip = int(p * 40) = A (40, 20), B (40, 24),
C (80, -40) -- three distinct cells, so k = 3; cell read for offset (i, j) = ip + (i, j).
  * offset (600, 600): every point reads its own stamp centre: 0 + 0 + 0 = 0 -> score 0.
  * offset (601, 600): one cell to the right of each centre.  A reads min(ker(1,0) = 3 from A, ker(1,4) =
    int(3 * 4.123) = 12 from B) = 3; B likewise 3; C reads 3.  idsum = 9,
    score = float(9) * (1/128f) / 3 = 0.0234375 (chargrid.cpp:277-279).
  * offset (600, 602): two cells up.  A reads min(ker(0,2) = 6 from A, ker(0,2) = 6 from B) = 6; B reads
    min(ker(0,2) = 6 from B, ker(0,6) = 18 from A) = 6; C reads 6 -> idsum 18, score 18/128/3 = 0.046875.
All 16 candidates fall into the result bin (int(x / 0.5), int(y / 0.5), int(0 / 0.2)) = (0, 0, 0) (truncation towards
zero, chargrid.h:50-66), the lowest score wins (addToPrunedMap, chargrid.cpp:36-46): ONE result, score 0, at
grid2world(600) = -15 + 0.025f * 600 (float arithmetic, gridmap.h:35-48) = 0.0.

Shifted query (every query point one cell to the right, +0.025 m in x): the zero-cost offset is (599, 600), i.e.
x = float(-15 + float(0.025f * 599)) -- *not* -0.025: 0.025f * 599 = 14.97500022... rounds to the float 14.97500038,
and -15 + 14.97500038 = -0.02499962 (exactly representable).  The expected value below is computed with numpy float32
following gridmap.h:35-48 literally."""
import numpy as np

REF = np.array([[1.0, 0.5], [1.0, 0.6], [2.0, -1.0]])
REGION = np.array([[-0.05, -0.05, 0.0, 0.05, 0.05, 0.005]], dtype=np.float32)   # theta in [0, 0.005) step 0.00625: theta = 0 only
GRID = dict(ll=(-15.0, -15.0), ur=(15.0, 15.0), res=0.025, kernel_range=0.2)
CELLS = [(640, 620), (640, 624), (680, 560)]


def grid2world(i):
    """gridmap.h:35-48 in float: lowerLeft + resolution * i."""
    return float(np.float32(-15.0) + np.float32(0.025) * np.float32(i))


def ker(di, dj):
    return min(int(3 * np.sqrt(di * di + dj * dj)), 25)


def expected_grid_value(x, y):
    """min over the reference points of ker(dx, dy) inside the 17 x 17 stamp, else the fill value 25."""
    best = 25
    for cx, cy in CELLS:
        if abs(x - cx) <= 8 and abs(y - cy) <= 8:
            best = min(best, ker(x - cx, y - cy))
    return best


def expected_score(i, j, query_cells):
    idsum = sum(expected_grid_value(qx + i, qy + j) for qx, qy in query_cells)
    return float(np.float32(np.float32(idsum) * np.float32(1.0 / 128.0)) / np.float64(len(query_cells)))


QUERY_CELLS = [(40, 20), (40, 24), (80, -40)]
EXPECTED_SAME = (grid2world(600), grid2world(600), 0.0, 0.0)                      # x, y, theta, score
EXPECTED_SHIFTED = (grid2world(599), grid2world(600), 0.0, 0.0)
assert EXPECTED_SAME[:2] == (0.0, 0.0)
assert abs(EXPECTED_SHIFTED[0] + 0.02499961853027344) < 1e-17
assert expected_score(601, 600, QUERY_CELLS) == 0.0234375 and expected_score(600, 602, QUERY_CELLS) == 0.046875


# ======================================================================================================================
# Round 5: four more cases worked out from the reference source, and one Gauss-Newton step from SURVEY.md Appendix A.
# Every expected value below is either a literal derived in the comment next to it or the comment's formula written out
# in IEEE arithmetic (numpy float32 / Python float where the reference computes in float / double); no implementation
# -- oracle or product -- is involved.  tests/test_oracle_matcher.py and tests/test_oracle_gn.py run the oracle on them,
# tests/test_matcher_gpu.py and tests/test_gn_gpu.py the C ABI on the GPU.

# ---------------------------------------------------------------------------------------------- (S) CharGrid::subsample
# chargrid.cpp:98-122 with res = 0.1 (ires = 10): bucket key = (int(10 x), int(10 y)) -- truncation towards zero, so
# (-0.1, 0.1) is ONE bucket along each axis --, buckets kept in a std::map ordered by x, then y (chargrid.cpp:88-94), a
# bucket's output = acc * (1. / count) (chargrid.cpp:67-71: a multiplication by the reciprocal, not a division), members
# added in input order.
#   P0 ( 0.26,  0.14) -> ( 2, 1)      P3 ( 0.21,  0.19) -> ( 2, 1)   joins P0
#   P1 (-0.05,  0.31) -> ( 0, 3)      P4 (-0.15, -0.02) -> (-1, 0)   (-1.5 -> -1, -0.2 -> 0)
#   P2 ( 0.04,  0.39) -> ( 0, 3)      P5 ( 0.23,  0.05) -> ( 2, 0)
#      joins P1 although x has the other sign
# map order: (-1, 0), (0, 3), (2, 0), (2, 1).
SUBSAMPLE_IN = np.array([[0.26, 0.14], [-0.05, 0.31], [0.04, 0.39], [0.21, 0.19], [-0.15, -0.02], [0.23, 0.05]])
SUBSAMPLE_OUT = np.array([[-0.15 * (1. / 1), -0.02 * (1. / 1)],
                          [(-0.05 + 0.04) * (1. / 2), (0.31 + 0.39) * (1. / 2)],
                          [0.23 * (1. / 1), 0.05 * (1. / 1)],
                          [(0.26 + 0.21) * (1. / 2), (0.14 + 0.19) * (1. / 2)]])

# ------------------------------------------------------------------------------- (H) CharGrid::hierarchicalSearch, 2 levels
# chargrid.cpp:376-400 builds the level parameters for nLevels = 2: level m = 2: step 2 cells, theta step max(m/2, 1) = 1 x
# thetaRes, bins 2 x (dx, dy, dth); level m = 1: step 1 cell, theta step thetaRes (m/2 = 0 < 1 -> m), bins (dx, dy, dth).
# chargrid.cpp:310-344: search the first level; every result r becomes a region r -+ bins/2 (computed in double, stored
# as Vector3f); search the last level in those regions.
# Scene on the close matcher's grid (0.025 m, K1 = 3): three points a quarter cell inside their cells so that neither
# lrint (rasteriser) nor truncation (search) nor a rotation by 0.004 rad (<= 0.2 cells at these radii) is near a boundary:
#   A (1.00625, 0.50625): 40 p = (40.25, 20.25) -> int -> (40, 20); rasterised at lrint(640.25, 620.25) = (640, 620)
#   B (1.00625, 0.60625): (40.25, 24.25) -> (40, 24); (640, 624)
#   C (2.00625, -0.98125): (80.25, -39.25) -> (80, -39) (towards zero); lrint(680.25, 560.75) = (680, 561)
# so offset (600, 600) puts every query point on its own stamp centre.  dx = dy = 0.04, dth = 0.004, thetaRes = 0.00625,
# maxScore = 0.03, region [-0.1, 0.1)^2 x [0, 0.005).
# Level 1 (m = 2): offsets lrint((-0.1f + 15) 40) = 596 .. < lrint((0.1f + 15) 40) = 604 in steps of 2: 596, 598, 600, 602; one
#   angle (0).  (600, 600): 0.  (598, 600): every point 2 cells from its centre, ker(2, 0) = 6 (B's stamp reaches A's cell
#   with ker(2, 4) = int(3 sqrt 20) = 13 > 6): 18 / 128 / 3 = 0.046875 > 0.03.  Same for (602, 600), (600, 598), (600, 602);
#   diagonals ker(2, 2) = 8: 0.0625.  ONE result: (grid2world(600), grid2world(600), 0) = (0, 0, 0), score 0.
# Region for level 2: (0, 0, 0) -+ (0.04, 0.04, 0.004) as floats.
# Level 2 (m = 1): offsets lrint((-0.04f + 15) 40 = 598.4) = 598 .. < lrint(601.6) = 602; angles t0 = double(float(-0.004)),
#   t1 = t0 + 0.00625 = 0.00225 < float(0.004); t1 + 0.00625 is beyond.  Neither rotation moves a point out of its cell, so
#   both angles score alike: (600, 600): 0; (599 / 601, 600) and (600, 599 / 601): ker(1, 0) = 3 per point (the other
#   stamps give >= 9) -> 9 / 128 / 3 = 0.0234375 < 0.03: accepted; (+-1, +-1): ker(1, 1) = 4 -> 0.03125: rejected; 598: >= 6
#   per point: rejected.
#   Bins (int(x / 0.04), int(y / 0.04), int(t / 0.004)): x = grid2world(599 .. 601) = -0.025, 0, 0.025 -> 0 (towards zero);
#   t0 / 0.004 = -1.00000005 -> -1, t1 -> 0.  Two bins, (0, 0, -1) and (0, 0, 0), each won by offset (600, 600) with score 0;
#   equal scores stay in map order (chargrid.cpp:292-307: concatenated in map order, sorted by score).
HIER_REF = np.array([[1.00625, 0.50625], [1.00625, 0.60625], [2.00625, -0.98125]])
HIER_REGION = np.array([[-0.1, -0.1, 0.0, 0.1, 0.1, 0.005]], dtype=np.float32)
HIER_ARGS = dict(theta_res=0.00625, max_score=0.03, dx=0.04, dy=0.04, dth=0.004, n_levels=2)
_T0 = float(np.float32(-0.004))
HIER_EXPECTED = [(grid2world(600), grid2world(600), _T0, 0.0), (grid2world(600), grid2world(600), _T0 + 0.00625, 0.0)]
assert HIER_EXPECTED[0][:2] == (0.0, 0.0) and int(_T0 / 0.004) == -1 and int((_T0 + 0.00625) / 0.004) == 0
assert _T0 + 0.00625 < float(np.float32(0.004)) <= _T0 + 0.00625 + 0.00625
for _p in HIER_REF:                                     # the premises of the derivation: a quarter cell inside, whatever the sign
    assert all(abs(abs(40 * _c) % 1 - 0.25) < 1e-9 for _c in _p)

# ------------------------------------------------------- (V) CharGrid::searchNonMatchedPoints + countPoints (verifyMatching)
# The loop-closure matcher's grid ([-35, 35]^2 at 0.1 m, kernel range 0.5: K1 = int(0.1 128) = 12, K2 = 64, 11 x 11).
# scan_matcher.cpp:467-470 / chargrid.cpp:444-455: a point of set 1 is "not matched" if it lies inside the grid and
# float(cell) * (1 / 128f) > 0.3 in the grid rasterised from set 2, i.e. cell >= 39.  Map: ONE point M (1.03, 2.03) ->
# cell lrint(360.3, 370.3) = (360, 370).  Points of set 1:
#   Q1 (1.03, 2.03)  cell (360, 370): 0                -> matched
#   Q2 (1.33, 2.03)  cell (363, 370): ker(3, 0) = 36   -> 0.28125: matched
#   Q3 (1.43, 2.03)  cell (364, 370): ker(4, 0) = 48   -> 0.375:   NOT matched
#   Q4 (5.03, 5.03)  cell (400, 400): fill 64          -> 0.5:     NOT matched
#   Q5 (40.0, 0.0)   outside the grid (isInside fails): skipped
# The two unexplained points are rasterised into a fresh grid (scan_matcher.cpp:473) and countPoints (chargrid.cpp:417-441)
# averages its cells over [lrint((lower + 35) 10), lrint((upper + 35) 10)): with lower (1.13, 1.73), upper (1.73, 2.33):
# x cells 361 .. 366, y cells 367 .. 372 -- the offsets -3 .. 2 from Q3's cell in both directions; Q4's stamp is 30 cells
# away.  score = float(sum of min(int(12 sqrt(di^2 + dj^2)), 64)) / float(36).
VERIFY_LL, VERIFY_UR, VERIFY_RES, VERIFY_RANGE = (-35.0, -35.0), (35.0, 35.0), 0.1, 0.5
VERIFY_PTS2 = np.array([[1.03, 2.03]])
VERIFY_PTS1 = np.array([[1.03, 2.03], [1.33, 2.03], [1.43, 2.03], [5.03, 5.03], [40.0, 0.0]])
VERIFY_LOWER, VERIFY_UPPER = np.array([1.13, 1.73], dtype=np.float32), np.array([1.73, 2.33], dtype=np.float32)
VERIFY_NONMATCHED = 2
_VSUM = sum(min(int(12 * np.sqrt(di * di + dj * dj)), 64) for di in range(-3, 3) for dj in range(-3, 3))
VERIFY_SCORE = float(np.float32(_VSUM) / np.float32(36))

# ---------------------------------------------------------- (L) ScanMatcher::scanMatchingLC: twin regions, normalise, merge
# scan_matcher.cpp:220-294 on the loop-closure grid.  One reference vertex (relposv = 0): region (-0.5, -1.5, -0.8) ..
# (0.5, 1.5, 0.8) as Vector3f; its twin has `lower[2] += M_PI; upper[2] += M_PI` -- a float lvalue plus a double: the sum is
# formed in double and narrowed ONCE (not float + float(pi)).  thetaRes 0.025, bins (0.5, 0.5, 0.2).
# Scene: a 16-beam laser (first beam -pi + pi/32, step pi/8) with four returns, two pairs of opposite beams at equal range,
# so that the scene maps onto itself under a half turn: beams 1 and 9 at 10.23 m, beams 6 and 14 at 8.70 m.  The same scan is
# reference and current scan.  All eight coordinates are checked below to lie 0.05 .. 0.45 cells inside their cells on the
# side towards zero, where lrint((p + 35) 10) - int(10 p) = 350 whatever the sign: offset (350, 350) and NO rotation puts
# every point on its own stamp centre (score 0), and so does a half turn.  A step of 0.025 rad moves the points by 2.2 .. 2.6
# cells (ker >= 24 each), a cell step by one cell (ker(1, 0) = 12): the two zero-score candidates are the only ones.
#   first search:  t = double(float(-0.8)) + 32 x 0.025 (accumulated) = -1.2e-8: the only angle within 0.0125 of 0
#   twin search:   float(double(float(-0.8)) + pi) = 2.34159255 (float + float(pi) would be 2.34159279, one ulp up), + 32 x 0.025
#                  = pi - 1.0e-7: below pi, so normalize_theta [g2o-recalled: only values outside [-pi, pi) are wrapped] leaves it
#                  (with the other reading of `+=` the angle is pi + 1.4e-7 and wraps to -pi: the two readings differ by 2 pi here)
# Both best results go through addToPrunedMap (bins int(x / 0.5), int(y / 0.5), int(t / 0.2)): (0, 0, 0) and (0, 0, 15);
# the map iterates in key order, so trel = [first result, twin result]; x = y = grid2world(350) = -35 + 0.1f 350 = 0.
LC_N_BEAMS, LC_ANGLE_MIN, LC_ANGLE_INC, LC_MAX_RANGE = 16, -np.pi + np.pi / 32, np.pi / 8, 30.0
LC_RANGES = np.full(16, 31.0, dtype=np.float32)
LC_RANGES[[1, 9]] = 10.23
LC_RANGES[[6, 14]] = 8.70
LC_POINTS = np.array([[float(np.float32(r)) * np.cos(LC_ANGLE_MIN + i * LC_ANGLE_INC), float(np.float32(r)) * np.sin(LC_ANGLE_MIN + i * LC_ANGLE_INC)]
                      for i, r in enumerate(LC_RANGES) if r < LC_MAX_RANGE])
for _p in LC_POINTS:
    for _c in _p:
        assert 0.05 < abs(10 * _c) % 1 < 0.45, (_p, "a coordinate too close to a cell boundary for the hand derivation")
        assert int(np.rint(np.float32((np.float32(_c) + np.float32(35.0)) * np.float32(10.0)))) - int(10 * _c) == 350


def _accumulate(t, n, step):
    for _ in range(n):
        t += step
    return t


_LO = float(np.float32(-0.8))
_LO_PI = float(np.float32(np.float64(np.float32(-0.8)) + np.pi))
LC_T_FIRST = _accumulate(_LO, 32, 0.025)
_T_TWIN = _accumulate(_LO_PI, 32, 0.025)
assert abs(LC_T_FIRST) < 1e-7 and 0 < np.pi - _T_TWIN < 1e-6
LC_T_TWIN = _T_TWIN                                                        # normalize_theta: inside [-pi, pi)
assert int(LC_T_TWIN / 0.2) == 15 and int(LC_T_FIRST / 0.2) == 0
_X350 = float(np.float32(-35.0) + np.float32(0.1) * np.float32(350))
assert _X350 == 0.0
LC_EXPECTED = [(_X350, _X350, LC_T_FIRST), (_X350, _X350, LC_T_TWIN)]
LC_MAX_SCORE = 0.05

# ------------------------------------------------------------------------------------- (G) one Gauss-Newton step by hand
# SURVEY.md Appendix A.  v0 fixed at the origin; v1 = (1.1, 0.2, 0), v2 = (2.3, -0.1, 0); edges 0 -> 1 and 1 -> 2 with
# z = (1, 0, 0), Omega = diag(100, 100, 1000) (graph_slam.cpp:72-73) and the closure 0 -> 2 with z = (2.1, 0.1, 0) (it
# disagrees with the chain: the step is not the trivial one),
# Omega = diag(1000, 1000, 10000) (graph_slam.cpp:75-76).  All angles are 0, so R(theta_i) = R(z^-1) = I:
#   e = (t_j - t_i - z_t, theta_j - theta_i - z_theta);   J_j = B = I;   J_i = A = [[-1, 0, dt.y], [0, -1, -dt.x], [0, 0, -1]]
#   e01 = (0.1, 0.2, 0)     chi2 = 100 (0.01 + 0.04)             = 5
#   e12 = (0.2, -0.3, 0)    chi2 = 100 (0.04 + 0.09)             = 13     dt = (1.2, -0.3)
#   e02 = (0.2, -0.2, 0)    chi2 = 1000 (0.04 + 0.04)            = 80     -> chi2 before the step: 98
#   H11 = O1 + A^T O1 A,  H12 = A^T O1,  H22 = O1 + O2;   b1 = -O1 e01 - A^T O1 e12,  b2 = -O1 e12 - O2 e02
# (a fixed vertex contributes nothing).  The 6 x 6 system is solved below with LAPACK; x <- x + dx (VertexSE2::oplusImpl adds
# the translation in the global frame).
GN_POSES = np.array([[0.0, 0.0, 0.0], [1.1, 0.2, 0.0], [2.3, -0.1, 0.0]])
GN_FIXED = np.array([1, 0, 0], dtype=np.uint8)
GN_FROM, GN_TO = np.array([0, 1, 0], dtype=np.int32), np.array([1, 2, 2], dtype=np.int32)
GN_MEAS = np.array([[1.0, 0.0, 0.0], [1.0, 0.0, 0.0], [2.1, 0.1, 0.0]])
GN_INFO = np.array([[100.0, 0, 0, 100.0, 0, 1000.0], [100.0, 0, 0, 100.0, 0, 1000.0], [1000.0, 0, 0, 1000.0, 0, 10000.0]])
GN_CHI2_BEFORE = 98.0
_O1, _O2 = np.diag([100.0, 100.0, 1000.0]), np.diag([1000.0, 1000.0, 10000.0])
_A = np.array([[-1.0, 0.0, -0.3], [0.0, -1.0, -1.2], [0.0, 0.0, -1.0]])
_e01, _e12, _e02 = np.array([0.1, 0.2, 0.0]), np.array([0.2, -0.3, 0.0]), np.array([0.2, -0.2, 0.0])
GN_H = np.block([[_O1 + _A.T @ _O1 @ _A, _A.T @ _O1], [_O1 @ _A, _O1 + _O2]])
GN_B = np.concatenate([-_O1 @ _e01 - _A.T @ _O1 @ _e12, -_O1 @ _e12 - _O2 @ _e02])
GN_DX = np.linalg.solve(GN_H, GN_B)
GN_POSES_AFTER = GN_POSES.copy()
GN_POSES_AFTER[1] += GN_DX[:3]
GN_POSES_AFTER[2] += GN_DX[3:]
