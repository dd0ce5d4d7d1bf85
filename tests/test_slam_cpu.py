"""Host logic of the key-frame driver (cg_mrslam_amd/slam.py) on the CPU: VerticesFinder against brute force,
ClosureBuffer / LoopClosureChecker known answers, and the whole loop on the oracle backend (no GPU, no libcgmr
compute calls)."""
import itertools
import math

import numpy as np
import pytest

from cg_mrslam_amd import synth
from cg_mrslam_amd.graph import PoseGraph
from cg_mrslam_amd.matcher import _se2_inv, _se2_mul
from cg_mrslam_amd.slam import (ClosureBuffer, GraphSLAMDriver, LoopClosureChecker, VerticesFinder, SM_INFO, run_srslam)

import oracle_backend as OB


def _random_graph(rng, n, extra):
    poses = np.concatenate([rng.uniform(-4, 4, size=(n, 2)), rng.uniform(-3, 3, size=(n, 1))], axis=1)
    ef = list(range(n - 1)) + list(rng.integers(0, n, size=extra))
    et = list(range(1, n)) + list(rng.integers(0, n, size=extra))
    keep = [(a, b) for a, b in zip(ef, et) if a != b]
    ef, et = np.array([a for a, _ in keep]), np.array([b for _, b in keep])
    E = len(ef)
    return PoseGraph(np.arange(n) + 100, poses, np.zeros(n, dtype=np.uint8), ef, et, np.zeros((E, 3)), np.tile(SM_INFO, (E, 1)))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_vertices_in_distance_vs_floyd_warshall(seed):
    rng = np.random.default_rng(seed)
    n = 40
    g = _random_graph(rng, n, 25)
    D = np.full((n, n), np.inf)
    np.fill_diagonal(D, 0.0)
    for a, b in zip(g.edge_from, g.edge_to):
        d = math.hypot(*(g.poses[a, :2] - g.poses[b, :2]))
        D[a, b] = D[b, a] = min(D[a, b], d)
    for k, i, j in itertools.product(range(n), range(n), range(n)):
        if D[i, k] + D[k, j] < D[i, j]:
            D[i, j] = D[i, k] + D[k, j]
    vf = VerticesFinder(g)
    for src in (0, 7, 23):
        for maxd in (2.0, 5.0, 11.0):
            want = {v for v in range(n) if D[src, v] < maxd - 2e-3} | {src}
            maybe = {v for v in range(n) if D[src, v] < maxd + 2e-3} | {src}       # the 1e-3 conditioner band
            got = vf.findVerticesInDistance(src, maxd)
            assert want <= got <= maybe


def test_sets_of_vertices_are_connected_components():
    g = _random_graph(np.random.default_rng(5), 30, 0)          # a chain 0-1-...-29
    vf = VerticesFinder(g)
    sets = vf.findSetsOfVertices({0, 1, 2, 5, 6, 9, 20, 21, 22, 23})
    assert [sorted(s) for s in sets] == [[0, 1, 2], [5, 6], [9], [20, 21, 22, 23]]
    assert vf.findClosestVertex({3, 4, 5}, 4) == 4


def test_closure_buffer_window():
    b = ClosureBuffer()
    e1 = {"from": 0, "to": 10, "meas": np.zeros(3)}
    e2 = {"from": 1, "to": 11, "meas": np.zeros(3)}
    b.addEdgeSet([e1]); b.addVertex(10)
    assert not b.checkList(3)
    b.updateList(3)
    b.addEdgeSet([e2]); b.addVertex(11)
    b.updateList(3)
    assert b.checkList(3)                    # vertex 10 has age 2 = window - 1
    b.updateList(3)                          # age 3 >= window: vertex 10 and its edge leave
    assert b.vertex_ids() == [11] and b.edges == [e2]


def test_loop_closure_checker_separates_consistent_from_outlier():
    # three recent vertices (floating part) shifted by a common error; candidates: two consistent with the shift, one not
    truth = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [0.1, 0.5, 0.0], [1.1, 0.5, 0.0], [2.1, 0.5, 0.0]], dtype=float)
    poses = truth.copy()
    shift = np.array([0.4, -0.3, 0.05])
    for v in (3, 4, 5):
        poses[v] = _se2_mul(shift, truth[v])
    def rel(a, b): return _se2_mul(_se2_inv(truth[a]), truth[b])
    edges = [{"from": 0, "to": 3, "meas": rel(0, 3)}, {"from": 1, "to": 4, "meas": rel(1, 4)},
             {"from": 2, "to": 5, "meas": _se2_mul(rel(2, 5), np.array([0.8, 0.0, 0.3]))}]
    lcc = LoopClosureChecker()
    lcc.init(poses, [3, 4, 5], edges, 2.0)
    lcc.check()
    assert lcc.inliers() == 2
    chi = [c for _, c in lcc.closures()]
    assert chi[0] < 1e-12 or chi[1] < 1e-12          # the winning hypothesis is exact for its own edge
    assert chi[0] < 2.0 and chi[1] < 2.0 and chi[2] > 2.0


def test_driver_on_oracle_backend_short_run(oracle, tmp_path):
    tr = synth.make_trajectory(60, laps=0.15)
    la = (tr["n_beams"], tr["angle_min"], tr["angle_inc"], tr["max_range"])
    slam = GraphSLAMDriver(OB.OracleContext(), OB.close_matcher(la), OB.lc_matcher(la))
    n = run_srslam(slam, tr["odom"], tr["scans"])
    g = slam.g
    assert n >= 15 and g.n_edges == n - 1
    assert slam.edge_kind.count("sm") >= n - 3                 # scan matching succeeds in this structured room
    assert int(g.ids[0]) == 0 and list(g.ids) == list(range(n))
    # the estimate stays on the true path (no loop closure in this short run, scan-match dead reckoning only)
    tp = tr["truth"]
    err = max(np.min(np.hypot(tp[:, 0] - p[0], tp[:, 1] - p[1])) for p in g.poses)
    assert err < 0.15
    # saveGraph writes every vertex with its ROBOTLASER1 line (graph_slam.cpp:620-623); the scans survive a round trip
    out = tmp_path / "robot-0-run.g2o"
    slam.saveGraph(out, precision=17)
    back = PoseGraph.load_g2o(out)
    assert len(back.lasers) == n
    np.testing.assert_array_equal(back.lasers[n - 1].ranges, slam.lasers[n - 1])
    np.testing.assert_array_equal(back.poses, g.poses)
