"""Key-frame driver: the single-robot SLAM loop of the reference around the GPU kernels.

Host-side mirror of
  * ``srslam`` main loop                 src/srslam.cpp:190-253      -> ``run_srslam``
  * ``GraphSLAM::{setInitialData,addDataSM,findConstraints,checkCovariance,addNeighboringVertices,
    checkHaveLaser,addClosures,checkClosures,updateClosures}``  src/slam/graph_slam.cpp:87-122,197-267,300-560
  * ``VerticesFinder``                   src/slam/vertices_finder.{h,cpp}  (g2o HyperDijkstra [g2o-recalled])
  * ``ClosureBuffer``                    src/slam/closure_buffer.cpp
  * ``LoopClosureChecker``               src/slam/closure_checker.cpp
from recorded arrays (odometry poses + laser ranges) instead of ROS topics.  Everything numeric goes through
``libcgmr.so``: ``optimize`` (GN), ``CovarianceEstimator`` (marginals), ``closeScanMatching`` and ``scanMatchingLC``
(the generic correlative search); this module is bookkeeping.

Where the reference iterates ``std::set<Vertex*>`` / ``std::set<Edge*>`` (address order, i.e. allocation
dependent) this code iterates in vertex-id / edge-creation order; the sets themselves are the same.
"""
from __future__ import annotations

import heapq
import math

import numpy as np

from .graph import GraphSLAM, PoseGraph, RobotLaser
from .matcher import _se2_inv, _se2_mul, normalize_theta

MAX_GRAPH_DIST_SM = 2.0     # vertices_finder.h:91-93
MIN_GRAPH_DIST_LC = 5.0
MAX_EUC_DIST_LC = 50.0
ODOM_INFO = np.array([100.0, 0, 0, 100.0, 0, 1000.0])      # graph_slam.cpp:72-76 (upper triangle xx xy xt yy yt tt)
SM_INFO = np.array([1000.0, 0, 0, 1000.0, 0, 10000.0])


def _edge_chi2(pi, pj, z, info_upper):
    """EdgeSE2::computeError + chi2 [g2o-recalled]: e = (z^-1 * (xi^-1 * xj)).toVector(), chi2 = e' Omega e."""
    e = _se2_mul(_se2_inv(z), _se2_mul(_se2_inv(pi), pj))
    a, b, c, d, f, g = info_upper
    return float(e[0] * (a * e[0] + b * e[1] + c * e[2]) + e[1] * (b * e[0] + d * e[1] + f * e[2]) +
                 e[2] * (c * e[0] + f * e[1] + g * e[2]))


class VerticesFinder:
    """vertices_finder.cpp over the flat graph arrays (vertex = index into ``graph.poses``)."""

    def __init__(self, graph: PoseGraph):
        self.g = graph

    def _adjacency(self):
        adj = [[] for _ in range(self.g.n_vertices)]
        for k in range(self.g.n_edges):
            i, j = int(self.g.edge_from[k]), int(self.g.edge_to[k])
            adj[i].append((k, j))
            adj[j].append((k, i))
        return adj

    def _dijkstra(self, source, cost, max_distance=float("inf"), conditioner=1e-3):
        """HyperDijkstra::shortestPaths [g2o-recalled]: undirected, an entry is improved only if the new distance is
        smaller by more than ``conditioner`` and below ``max_distance``; visited = every vertex popped."""
        adj = self._adjacency()
        dist = {source: 0.0}
        frontier = [(0.0, source)]
        visited = set()
        while frontier:
            _, u = heapq.heappop(frontier)
            visited.add(u)
            du = dist[u]
            for k, z in adj[u]:
                c = cost(k, u, z)
                if c == float("inf"):
                    continue
                dz = du + c
                if dz + conditioner < dist.get(z, float("inf")) and dz < max_distance:
                    dist[z] = dz
                    heapq.heappush(frontier, (dz, z))
        return visited

    def vertexDistance(self, a, b):   # noqa: N802
        d = self.g.poses[a, :2] - self.g.poses[b, :2]
        return float(math.sqrt(d[0] * d[0] + d[1] * d[1]))

    def findVerticesInDistance(self, current, graphdist):   # noqa: N802  (vertices_finder.cpp:35-43)
        return self._dijkstra(current, lambda k, u, z: self.vertexDistance(u, z), graphdist)

    def findVerticesLoopClosing(self, current, graphdist):   # noqa: N802  (vertices_finder.cpp:45-60)
        visited = self.findVerticesInDistance(current, graphdist)
        return {v for v in range(self.g.n_vertices)
                if v not in visited and self.vertexDistance(current, v) <= MAX_EUC_DIST_LC}

    def findVerticesScanMatching(self, current):   # noqa: N802  (vertices_finder.cpp:62-80)
        vset = self.findVerticesInDistance(current, MAX_GRAPH_DIST_SM)
        vset |= self.findVerticesLoopClosing(current, MIN_GRAPH_DIST_LC)
        vset.discard(current)
        return vset

    def findSetsOfVertices(self, mixedvset):   # noqa: N802  (vertices_finder.cpp:83-99)
        """Connected components of ``mixedvset`` under edges with both end points in the set (id order)."""
        rest = set(mixedvset)
        out = []
        while rest:
            start = min(rest, key=lambda v: self.g.ids[v])
            members = frozenset(rest)
            comp = self._dijkstra(start, lambda k, u, z: 1.0 if (u in members and z in members) else float("inf"))
            out.append(comp)
            rest -= comp
        return out

    def findClosestVertex(self, vset, current):   # noqa: N802  (vertices_finder.cpp:101-114)
        best, dist = None, float("inf")
        for v in sorted(vset, key=lambda q: self.g.ids[q]):
            d = self.vertexDistance(current, v)
            if d < dist:
                dist, best = d, v
        return best


class ClosureBuffer:
    """closure_buffer.cpp: loop-closure candidates of the last ``window`` key frames.  Edges are
    (from, to, meas) tuples kept in creation order; vertices carry an age."""

    def __init__(self):
        self.edges = []          # list of dict(from,to,meas,id)
        self.vertices = []       # list of [vertex, time]

    def addEdge(self, e):   # noqa: N802
        if not any(e is q for q in self.edges):
            self.edges.append(e)

    def removeEdge(self, e):   # noqa: N802
        self.edges = [q for q in self.edges if q is not e]

    def addEdgeSet(self, eset):   # noqa: N802
        for e in eset:
            self.addEdge(e)

    def findVertex(self, v):   # noqa: N802
        return any(q[0] == v for q in self.vertices)

    def addVertex(self, v):   # noqa: N802
        self.vertices.append([v, 0])

    def removeVertex(self, v):   # noqa: N802
        if any(q[0] == v for q in self.vertices):
            self.edges = [e for e in self.edges if e["from"] != v and e["to"] != v]
            self.vertices = [q for q in self.vertices if q[0] != v]

    def updateList(self, window):   # noqa: N802
        for q in self.vertices:
            q[1] += 1
        for v, t in list(self.vertices):
            if t >= window:
                self.removeVertex(v)

    def checkList(self, window):   # noqa: N802
        return any(t == window - 1 for _, t in self.vertices)

    def vertex_ids(self):
        return [v for v, _ in self.vertices]


class LoopClosureChecker:
    """closure_checker.cpp, matching type "2dPose": every candidate edge in turn is made exact by moving the
    floating vertices rigidly; the hypothesis that makes most candidates inliers (then lowest chi2) wins."""

    def init(self, poses, local_vertices, closing_edges, inlier_threshold):
        self.poses = poses
        self.local = list(local_vertices)
        self.edges = list(closing_edges)
        self.thr = inlier_threshold
        self.best_inliers = 0
        self.best_chi2 = float("inf")
        self.best_result = [float("inf")] * len(self.edges)

    def check(self):
        for e in self.edges:
            chi = self._apply_zero_error_transform(e)
            inl = [c for c in chi if c < self.thr]
            n, tot = len(inl), float(sum(inl))
            if n > self.best_inliers or (n == self.best_inliers and tot < self.best_chi2):
                self.best_inliers, self.best_chi2, self.best_result = n, tot, chi

    def _apply_zero_error_transform(self, e):
        local = set(self.local)
        root = None
        if e["from"] in local:
            root = e["from"]
        if e["to"] in local:
            root = e["to"]
        assert root is not None, "the loop closure does not have any vertex in the floating part of the map"
        pf, pt = self.poses[e["from"]], self.poses[e["to"]]
        new_root = _se2_mul(pt, _se2_inv(e["meas"])) if root == e["from"] else _se2_mul(pf, e["meas"])
        motion = _se2_mul(new_root, _se2_inv(self.poses[root]))
        moved = {v: _se2_mul(motion, self.poses[v]) for v in self.local}
        chi = []
        for q in self.edges:
            a = moved.get(q["from"], self.poses[q["from"]])
            b = moved.get(q["to"], self.poses[q["to"]])
            chi.append(_edge_chi2(a, b, q["meas"], q.get("info", SM_INFO)))
        return chi

    def inliers(self):
        return self.best_inliers

    def chi2(self):
        return self.best_chi2

    def closures(self):
        return list(zip(self.edges, self.best_result))


class GraphSLAMDriver(GraphSLAM):
    """``GraphSLAM`` with the key-frame front end.  ``close_matcher`` / ``lc_matcher`` are ``ScanMatcher`` /
    ``LCScanMatcher`` instances (graph_slam.cpp:58-62); ``ctx`` provides ``gn_optimize`` and
    ``covariance_estimate``."""

    def __init__(self, ctx, close_matcher, lc_matcher, idRobot=0, baseId=10000, windowLoopClosure=10, maxScore=0.15,   # noqa: N803
                 inlierThreshold=2.0, minInliers=7):   # noqa: N803
        g = PoseGraph(np.zeros(0, dtype=np.int32), np.zeros((0, 3)), np.zeros(0, dtype=np.uint8),
                      np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros((0, 3)), np.zeros((0, 6)))
        super().__init__(g, ctx=ctx)
        self.g = self.graph
        self.close_matcher, self.lc_matcher = close_matcher, lc_matcher
        self.idRobot, self.baseId = idRobot, baseId
        self.windowLoopClosure, self.maxScore = windowLoopClosure, maxScore
        self.inlierThreshold, self.minInliers = inlierThreshold, minInliers
        self.lasers = {}            # vertex index -> float32 ranges
        self._id_index = {}         # vertex id -> index (g2o's VertexIDMap lookup)
        self._running_vertex_id = 0
        self._running_edge_id = 0
        self.edge_ids = []
        self.edge_kind = []         # "odom" | "sm" | "lc"
        self._vf = VerticesFinder(g)
        self._closures = ClosureBuffer()
        self.lcc = LoopClosureChecker()
        self._last_vertex = None
        self._last_odom = None
        self.log = []

    # ------------------------------------------------------------------ graph bookkeeping
    def _index_of_id(self, vid):
        return self._id_index.get(int(vid))

    def _add_vertex(self, vid, pose, fixed, ranges):
        g = self.g
        g.ids = np.append(g.ids, np.int32(vid))
        g.poses = np.vstack([g.poses, np.asarray(pose, dtype=np.float64).reshape(1, 3)])
        g.fixed = np.append(g.fixed, np.uint8(1 if fixed else 0))
        idx = g.n_vertices - 1
        self._id_index[int(vid)] = idx
        scan = np.array(ranges, dtype=np.float32)                 # own, read-only copy: the matcher wrappers may cache the scan sets built from it
        scan.setflags(write=False)
        self.lasers[idx] = scan
        return idx

    def _add_edge(self, i, j, meas, info, kind, eid):
        g = self.g
        g.edge_from = np.append(g.edge_from, np.int32(i))
        g.edge_to = np.append(g.edge_to, np.int32(j))
        g.meas = np.vstack([g.meas, np.asarray(meas, dtype=np.float64).reshape(1, 3)])
        g.info = np.vstack([g.info, np.asarray(info, dtype=np.float64).reshape(1, 6)])
        g.edge_level = np.append(g.edge_level, np.int32(0))
        self.edge_ids.append(eid)
        self.edge_kind.append(kind)

    def lastVertex(self):   # noqa: N802
        return self._last_vertex

    def isMyVertex(self, v):   # noqa: N802
        return int(self.g.ids[v]) // self.baseId == self.idRobot

    def _scans(self, vset):
        order = sorted(vset, key=lambda q: self.g.ids[q])
        return order, [(self.lasers[v], self.g.poses[v].copy()) for v in order]

    def saveGraph(self, filename, precision=None):   # noqa: N802  (graph_slam.cpp:620-623: vertices carry their RobotLaser)
        cfg = self.close_matcher.cfg
        lp = [float(cfg.laser_pose[k]) for k in range(3)]
        for idx, r in self.lasers.items():                       # optimize() leaves odomPose = the vertex estimate (:569-574)
            self.g.lasers[idx] = RobotLaser(r, cfg.angle_min, cfg.angle_inc, cfg.max_range, odom_pose=self.g.poses[idx],
                                            laser_pose=lp)
        self.g.save_g2o(filename, precision)
        return True

    # ------------------------------------------------------------------ graph_slam.cpp:87-122
    def setInitialData(self, initialOdom, ranges):   # noqa: N802,N803
        self._last_odom = np.asarray(initialOdom, dtype=np.float64).copy()
        self._last_vertex = self._add_vertex(self.idRobot * self.baseId, self._last_odom, True, ranges)

    # ------------------------------------------------------------------ graph_slam.cpp:197-267
    def addDataSM(self, currentOdom, ranges):   # noqa: N802,N803
        currentOdom = np.asarray(currentOdom, dtype=np.float64)   # noqa: N806
        last = self._last_vertex
        displacement = _se2_mul(_se2_inv(self._last_odom), currentOdom)
        curr_est = _se2_mul(self.g.poses[last], displacement)
        self._running_vertex_id += 1
        v = self._add_vertex(self._running_vertex_id + self.idRobot * self.baseId, curr_est, False, ranges)
        self._running_edge_id += 1
        eid = self._running_edge_id + self.idRobot * self.baseId
        vset = {last}
        for j in range(1, 6):                                   # gap = 5
            vj = self._index_of_id(int(self.g.ids[last]) - j)
            if vj is None:
                break
            vset.add(vj)
        order, scans = self._scans(vset)
        found, transf = self.close_matcher.closeScanMatchingVSet(scans, order.index(last), ranges, curr_est, self.maxScore)
        if found:
            self._add_edge(last, v, transf, SM_INFO, "sm", eid)
        else:                                                   # trust the odometry
            self._add_edge(last, v, displacement, ODOM_INFO, "odom", eid)
        self.log.append(("addDataSM", int(self.g.ids[v]), bool(found)))
        self._last_odom = currentOdom.copy()
        self._last_vertex = v

    # ------------------------------------------------------------------ graph_slam.cpp:310-353
    def checkCovariance(self, vset):   # noqa: N802
        if not vset:
            return vset
        g = self.g
        last = self._last_vertex
        order = sorted(vset, key=lambda q: g.ids[q])
        cov = self.ctx.covariance_estimate(g.poses, g.edge_from, g.edge_to, g.meas, g.info, last,
                                           np.asarray(order, dtype=np.int32))
        keep = set()
        for k, v in enumerate(order):
            pxy = cov[k][:2, :2]
            delta = _se2_mul(_se2_inv(g.poses[v]), g.poses[last])
            hx, hy = float(delta[0]), float(delta[1])
            rng = 1.0                                            # perceptionRange
            hx = hx - rng if hx - rng > 0 else (hx + rng if hx + rng < 0 else 0.0)
            hy = hy - rng if hy - rng > 0 else (hy + rng if hy + rng < 0 else 0.0)
            det = pxy[0, 0] * pxy[1, 1] - pxy[0, 1] * pxy[1, 0]
            inv = np.array([[pxy[1, 1], -pxy[0, 1]], [-pxy[1, 0], pxy[0, 0]]]) / det
            d2 = hx * (inv[0, 0] * hx + inv[0, 1] * hy) + hy * (inv[1, 0] * hx + inv[1, 1] * hy)
            if not d2 > 5.99:
                keep.add(v)
        return keep

    # ------------------------------------------------------------------ graph_slam.cpp:355-382
    def addNeighboringVertices(self, vset, gap):   # noqa: N802
        vset = set(vset)
        last_id = int(self.g.ids[self._last_vertex])
        for vertex in sorted(set(vset), key=lambda q: self.g.ids[q]):
            for sign in (+1, -1):
                for i in range(1, gap + 1):
                    vid = int(self.g.ids[vertex]) + sign * i
                    v = self._index_of_id(vid)
                    if v is not None and vid != last_id:
                        if v not in vset:
                            vset.add(v)
                        else:
                            break
        return vset

    def checkHaveLaser(self, vset):   # noqa: N802  (graph_slam.cpp:300-307)
        return {v for v in vset if v in self.lasers}

    # ------------------------------------------------------------------ graph_slam.cpp:388-485
    def findConstraints(self):   # noqa: N802
        g = self.g
        last = self._last_vertex
        self.optimize(1)                                         # so that the last added edge is satisfied
        vset = self._vf.findVerticesScanMatching(last)
        vset = self.checkCovariance(vset)
        vset = self.addNeighboringVertices(vset, 8)
        vset = self.checkHaveLaser(vset)
        sets = self._vf.findSetsOfVertices(vset)
        loop_closing = []
        for myvset in sets:
            closest = self._vf.findClosestVertex(myvset, last)
            if int(g.ids[closest]) == int(g.ids[last]) - 1:      # already have this edge
                continue
            order, scans = self._scans(myvset)
            ref_index = order.index(closest)
            if (not self.isMyVertex(closest)) or abs(int(g.ids[last]) - int(g.ids[closest])) > 10:
                results = self.lc_matcher.scanMatchingLC(scans, ref_index, [(self.lasers[last], g.poses[last].copy())], 0,
                                                         self.maxScore)
                for r in results:
                    self._running_edge_id += 1
                    loop_closing.append({"from": closest, "to": last, "meas": np.asarray(r, dtype=np.float64).copy(),
                                         "id": self._running_edge_id + self.baseId, "added": False})
                self.log.append(("lc", int(g.ids[closest]), int(g.ids[last]), len(results)))
            else:
                found, transf = self.close_matcher.closeScanMatchingVSet(scans, ref_index, self.lasers[last], g.poses[last],
                                                                          self.maxScore)
                if found:
                    self._running_edge_id += 1
                    self._add_edge(closest, last, transf, SM_INFO, "sm", self._running_edge_id + self.baseId)
                self.log.append(("close", int(g.ids[closest]), int(g.ids[last]), bool(found)))
        if loop_closing:
            self.addClosures(loop_closing)
        self.checkClosures()
        self.updateClosures()

    def addClosures(self, edges):   # noqa: N802  (graph_slam.cpp:487-491)
        self._closures.addEdgeSet(edges)
        self._closures.addVertex(self._last_vertex)

    def checkClosures(self):   # noqa: N802  (graph_slam.cpp:493-533)
        if not self._closures.checkList(self.windowLoopClosure):
            return
        self.lcc.init(self.g.poses, self._closures.vertex_ids(), self._closures.edges, self.inlierThreshold)
        self.lcc.check()
        self.log.append(("lcc", self.lcc.inliers(), self.lcc.chi2()))
        if self.lcc.inliers() >= self.minInliers:
            for e, chi in self.lcc.closures():
                if chi < self.inlierThreshold and not e["added"]:      # HyperGraph::addEdge refuses an edge twice
                    e["added"] = True
                    self._add_edge(e["from"], e["to"], e["meas"], SM_INFO, "lc", e["id"])

    def updateClosures(self):   # noqa: N802  (graph_slam.cpp:536-560)
        self._closures.updateList(self.windowLoopClosure)


def run_srslam(slam: GraphSLAMDriver, odom, scans, initial_pose=None, linearUpdate=0.25, angularUpdate=math.pi / 4,   # noqa: N803
               iterations=5):
    """The loop of srslam.cpp:190-253 over recorded ``odom`` (T,3) and ``scans`` (T,B): a key frame whenever the
    dead-reckoned estimate moved more than ``linearUpdate`` or turned more than ``angularUpdate`` since the last
    vertex; per key frame addDataSM -> findConstraints -> optimize(5).  Returns the number of key frames."""
    odom = np.asarray(odom, dtype=np.float64)
    curr_est = odom[0].copy() if initial_pose is None else np.asarray(initial_pose, dtype=np.float64).copy()
    slam.setInitialData(curr_est, scans[0])
    odom_k1 = odom[0].copy()
    for k in range(1, len(odom)):
        rel = _se2_mul(_se2_inv(odom_k1), odom[k])
        curr_est = _se2_mul(curr_est, rel)
        odom_k1 = odom[k].copy()
        lastp = slam.g.poses[slam.lastVertex()]
        d = math.hypot(lastp[0] - curr_est[0], lastp[1] - curr_est[1])
        if d > linearUpdate or abs(lastp[2] - curr_est[2]) > angularUpdate:
            slam.addDataSM(curr_est, scans[k])
            slam.findConstraints()
            slam.optimize(iterations)
            curr_est = slam.g.poses[slam.lastVertex()].copy()
    slam.optimize(iterations)
    return slam.g.n_vertices


__all__ = ["VerticesFinder", "ClosureBuffer", "LoopClosureChecker", "GraphSLAMDriver", "run_srslam", "normalize_theta"]
