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
