"""Test infrastructure: plain-numpy restatement of the reference's ``CondensedGraphBuffer`` bookkeeping and wire format
(src/mrslam/condensed_graph/condensed_graph_buffer.{h,cpp}; src/mrslam/msg_factory.h:78-112,200-218;
src/mrslam/mr_graph_slam.cpp:331-395), with the numeric step (``CondensedGraphCreator::compute``) delegated to whatever
``ctx.condense`` it is given -- the CPU oracle in the tests.  The product implements the same protocol in C++ / HIP
behind the C ABI (csrc/mrslam_api.cpp, ``cg_mrslam_amd.condensed.RobotGraph``); the GPU tests run both through the same
multi-robot rounds and compare edge by edge.  ``RefRobotGraph`` below gives this restatement the product's interface.
Nothing under ``cg_mrslam_amd/`` imports this file.
"""
from __future__ import annotations

import numpy as np

from cg_mrslam_amd.graph import PoseGraph

EDGE_DTYPE = np.dtype([("from", "<i4"), ("to", "<i4"), ("est", "<f4", (3,)), ("info", "<f4", (6,))])
assert EDGE_DTYPE.itemsize == 44            # CondensedGraphMessage: 44 bytes per edge


def select_gauge_centroid(poses_xy: np.ndarray) -> int:
    """``selectGaugeCentroid`` (condensed_graph_buffer.cpp:318-345): index of the vertex closest to the
    centroid of the requested vertices' translations (first one wins ties)."""
    c = poses_xy.sum(axis=0) / len(poses_xy)
    d = np.sqrt(((poses_xy - c) ** 2).sum(axis=1))
    return int(np.argmin(d))


class CondensedGraphBuffer:
    def __init__(self, graph: PoseGraph, robot: int, n_robots: int, ctx=None, cap_edges: int = 128,
                 base_id: int = 10000):
        self.g = graph
        self.robot = robot
        self.n_robots = n_robots
        self.ctx = ctx
        self.cap = cap_edges
        self.base_id = base_id
        self.out_closures = {}      # peer -> sorted int array of MY vertex ids the peer asked for
        self.in_closures = {}       # peer -> sorted int array of the peer's vertex ids I ask for
        self.out_condensed = {}     # peer -> structured array EDGE_DTYPE (ids), level peer+1 in g2o terms
        self.in_edge_src = np.full(graph.n_edges, -1, dtype=np.int32)   # peer that sent a level-0 edge, -1 = own

    # ------------------------------------------------------------------ closures
    def insertOutClosure(self, peer, vertex_ids):   # noqa: N802  (condensed_graph_buffer.cpp:152-170)
        cur = self.out_closures.get(peer, np.zeros(0, dtype=np.int64))
        self.out_closures[peer] = np.union1d(cur, np.asarray(vertex_ids, dtype=np.int64))

    def insertInClosure(self, peer, vertex_ids):    # noqa: N802  (condensed_graph_buffer.cpp:131-150)
        cur = self.in_closures.get(peer, np.zeros(0, dtype=np.int64))
        self.in_closures[peer] = np.union1d(cur, np.asarray(vertex_ids, dtype=np.int64))

    # ------------------------------------------------------------------ my edges
    def my_edge_mask(self):
        """``getMyEdges`` (condensed_graph_buffer.cpp:347-366): every edge except those received from other
        robots; edges built *for* other robots live at level peer+1 and are not in the arrays at all."""
        return (self.in_edge_src < 0) & (self.g.edge_level == 0)

    def _index_of_ids(self, ids):
        order = np.argsort(self.g.ids, kind="stable")
        pos = np.searchsorted(self.g.ids[order], ids)
        pos = np.minimum(pos, len(order) - 1)
        ok = self.g.ids[order][pos] == ids
        return np.where(ok, order[pos], -1)

    # ------------------------------------------------------------------ build
    def computeCondensedGraph(self, peer):   # noqa: N802  (condensed_graph_buffer.cpp:437-485)
        """Star of condensed edges over the vertices ``peer`` asked for; stored (ids) in out_condensed[peer]."""
        want = self.out_closures.get(peer)
        if want is None or len(want) < 2:
            self.out_condensed[peer] = np.zeros(0, dtype=EDGE_DTYPE)
            return self.out_condensed[peer]
        idx = self._index_of_ids(want)
        idx = idx[idx >= 0]
        if len(idx) < 2:
            self.out_condensed[peer] = np.zeros(0, dtype=EDGE_DTYPE)
            return self.out_condensed[peer]
        gauge = int(idx[select_gauge_centroid(self.g.poses[idx, :2])])
        m = self.my_edge_mask()
        to, est, iu, _ = self.ctx.condense(self.g.poses, self.g.edge_from[m], self.g.edge_to[m], self.g.meas[m],
                                           self.g.info[m], gauge, idx.astype(np.int32))
        e = np.zeros(len(to), dtype=EDGE_DTYPE)
        e["from"] = self.g.ids[gauge]
        e["to"] = self.g.ids[to]
        e["est"] = est.astype(np.float32)          # doubles are narrowed to float32 on the wire
        e["info"] = iu.astype(np.float32)
        self.out_condensed[peer] = e
        return e

    # ------------------------------------------------------------------ wire
    def wire_bytes(self):
        R, cap = self.n_robots, self.cap
        return 4 * (2 + 2 * R) + R * cap * EDGE_DTYPE.itemsize + R * cap * 4

    def pack(self) -> np.ndarray:
        """Fixed-capacity send buffer: header {robot, n_robots, n_edges[R], n_closures[R]} int32,
        edges[R][cap] (44 B each, slice p = edges for peer p), closures[R][cap] int32 (ids I request from p)."""
        R, cap = self.n_robots, self.cap
        hdr = np.zeros(2 + 2 * R, dtype=np.int32)
        hdr[0], hdr[1] = self.robot, R
        edges = np.zeros((R, cap), dtype=EDGE_DTYPE)
        clos = np.zeros((R, cap), dtype=np.int32)
        for p in range(R):
            e = self.out_condensed.get(p)
            c = self.in_closures.get(p)
            # a message that does not fit is not sent at all: ComboMessage::toCharArray returns 0 beyond MAX_LENGTH_MSG and
            # GraphComm::send skips it (graph_comm.cpp:112-122) -- never truncated
            if (e is not None and len(e) > cap) or (c is not None and len(c) > cap):
                self.skipped = getattr(self, "skipped", 0) + 1
                continue
            if e is not None and len(e):
                edges[p, :len(e)] = e
                hdr[2 + p] = len(e)
            if c is not None and len(c):
                clos[p, :len(c)] = c
                hdr[2 + R + p] = len(c)
        return np.concatenate([hdr.view(np.uint8), edges.reshape(-1).view(np.uint8), clos.reshape(-1).view(np.uint8)])

    def unpack(self, buf: np.ndarray):
        """Inverse of ``pack`` for ONE sender's buffer: (sender, edges addressed to me, closures it requests from me)."""
        R, cap = self.n_robots, self.cap
        hdr = buf[:4 * (2 + 2 * R)].view(np.int32)
        sender = int(hdr[0])
        o = 4 * (2 + 2 * R)
        edges = buf[o:o + R * cap * 44].view(EDGE_DTYPE).reshape(R, cap)
        o += R * cap * 44
        clos = buf[o:o + R * cap * 4].view(np.int32).reshape(R, cap)
        me = self.robot
        return sender, edges[me, :hdr[2 + me]].copy(), clos[me, :hdr[2 + R + me]].copy()

    # ------------------------------------------------------------------ receive
    def insertEdgesFromRobot(self, peer, edges):   # noqa: N802  (condensed_graph_buffer.cpp:487-510)
        """Replace the previous set received from ``peer`` by ``edges``; edges whose end points are not in my
        graph are skipped (src/mrslam/mr_graph_slam.cpp:363)."""
        g = self.g
        keep = self.in_edge_src != peer
        fi = self._index_of_ids(edges["from"].astype(np.int64)) if len(edges) else np.zeros(0, dtype=np.int64)
        ti = self._index_of_ids(edges["to"].astype(np.int64)) if len(edges) else np.zeros(0, dtype=np.int64)
        ok = (fi >= 0) & (ti >= 0)
        n_new = int(ok.sum())
        g.edge_from = np.concatenate([g.edge_from[keep], fi[ok].astype(np.int32)])
        g.edge_to = np.concatenate([g.edge_to[keep], ti[ok].astype(np.int32)])
        g.meas = np.concatenate([g.meas[keep], edges["est"][ok].astype(np.float64).reshape(-1, 3)])
        g.info = np.concatenate([g.info[keep], edges["info"][ok].astype(np.float64).reshape(-1, 6)])
        g.edge_level = np.concatenate([g.edge_level[keep], np.zeros(n_new, dtype=np.int32)])
        self.in_edge_src = np.concatenate([self.in_edge_src[keep], np.full(n_new, peer, dtype=np.int32)])
        return n_new

    def ingest_from(self, sender, edges, closures):
        """MRGraphSLAM::addInterRobotData (mr_graph_slam.cpp:331-395): requests for vertices I have become out-closures
        (nothing happens for an empty list, :345); the edges whose end points I know replace the previous set -- only
        if at least one survives (:393-394 ``if (edges.size()) insertEdgesFromRobot``)."""
        known = self._index_of_ids(np.asarray(closures, dtype=np.int64)) >= 0 if len(closures) else np.zeros(0, dtype=bool)
        if known.any():
            self.insertOutClosure(sender, np.asarray(closures)[known])
        if len(edges):
            fi = self._index_of_ids(edges["from"].astype(np.int64))
            ti = self._index_of_ids(edges["to"].astype(np.int64))
            if ((fi >= 0) & (ti >= 0)).any():
                return self.insertEdgesFromRobot(sender, edges)
        return 0

    # ------------------------------------------------------------------ one exchange round
    def exchange(self, group=None, device=None):
        """All-gather every rank's send buffer and ingest what is addressed to me.  Returns bytes gathered.
        With torch.distributed uninitialised (single robot) this is a no-op."""
        import torch
        import torch.distributed as dist
        if not dist.is_available() or not dist.is_initialized():
            return 0
        send = torch.from_numpy(self.pack())
        if device is not None:
            send = send.to(device)
        world = dist.get_world_size(group)
        recv = torch.empty(world * send.numel(), dtype=torch.uint8, device=send.device)
        dist.all_gather_into_tensor(recv, send, group=group)
        host = recv.cpu().numpy().reshape(world, -1)
        for src in range(world):
            sender, edges, closures = self.unpack(host[src])
            if sender == self.robot:
                continue
            self.ingest_from(sender, edges, closures)
        return int(recv.numel())


class RefRobotGraph:
    """The interface of ``cg_mrslam_amd.condensed.RobotGraph`` on the numpy restatement above; ``ctx`` provides
    ``gn_optimize`` and ``condense`` (tests/oracle_backend.OracleContext)."""

    def __init__(self, ctx, robot, n_robots, base_id=10000, cap_edges=128):
        z = np.zeros
        self.pg = PoseGraph(z(0, dtype=np.int64), z((0, 3)), z(0, dtype=np.uint8), z(0, dtype=np.int32), z(0, dtype=np.int32),
                            z((0, 3)), z((0, 6)))
        self.buf = CondensedGraphBuffer(self.pg, robot, n_robots, ctx=ctx, cap_edges=cap_edges, base_id=base_id)
        self.ctx, self.robot, self.n_robots, self.cap = ctx, robot, n_robots, cap_edges
        self.gauge = {}

    def add_vertices(self, ids, poses, fixed=None):
        g = self.pg
        ids = np.asarray(ids, dtype=np.int64)
        g.ids = np.concatenate([g.ids, ids])
        g.poses = np.vstack([g.poses, np.asarray(poses, dtype=np.float64).reshape(-1, 3)])
        g.fixed = np.concatenate([g.fixed, np.zeros(len(ids), dtype=np.uint8) if fixed is None else np.asarray(fixed, dtype=np.uint8)])

    def add_edges(self, from_ids, to_ids, meas, info):
        g, b = self.pg, self.buf
        fi = b._index_of_ids(np.asarray(from_ids, dtype=np.int64))
        ti = b._index_of_ids(np.asarray(to_ids, dtype=np.int64))
        assert (fi >= 0).all() and (ti >= 0).all()
        own = b.in_edge_src < 0                                   # own edges stay in front of the received ones
        n = len(fi)
        def ins(a, new):
            return np.concatenate([a[own], new, a[~own]])
        g.edge_from = ins(g.edge_from, fi.astype(np.int32)); g.edge_to = ins(g.edge_to, ti.astype(np.int32))
        g.meas = ins(g.meas, np.asarray(meas, dtype=np.float64).reshape(-1, 3))
        g.info = ins(g.info, np.asarray(info, dtype=np.float64).reshape(-1, 6))
        g.edge_level = ins(g.edge_level, np.zeros(n, dtype=np.int32))
        b.in_edge_src = ins(b.in_edge_src, np.full(n, -1, dtype=np.int32))

    def counts(self):
        b = self.buf
        return {"vertices": self.pg.n_vertices, "own_edges": int((b.in_edge_src < 0).sum()),
                "received_edges": int((b.in_edge_src >= 0).sum()), "peers_with_requests": len(b.out_closures)}

    def optimize(self, iters):
        g = self.pg
        ef, et, meas, info = g.level0()
        rc, poses, chi2 = self.ctx.gn_optimize(g.poses, g.fixed, ef, et, meas, info, int(iters), raise_on_cholesky=False)
        g.poses[:] = poses
        return rc, chi2

    def poses(self, first=0, n=None):
        n = self.pg.n_vertices - first if n is None else n
        return self.pg.poses[first:first + n].copy()

    def insertInClosure(self, peer, ids):   # noqa: N802
        self.buf.insertInClosure(peer, ids)

    def insertOutClosure(self, peer, ids):   # noqa: N802
        ids = np.asarray(ids, dtype=np.int64)
        ok = self.buf._index_of_ids(ids) >= 0 if len(ids) else np.zeros(0, dtype=bool)
        if ok.any():
            self.buf.insertOutClosure(peer, ids[ok])

    def computeCondensedGraph(self, peer=-1):   # noqa: N802
        built = 0
        for p in (range(self.n_robots) if peer < 0 else [peer]):
            if p == self.robot or p not in self.buf.out_closures:
                continue
            e, gauge = self._condense64(p)
            built += 1 if len(e[0]) else 0
        return built

    def _condense64(self, peer):
        """computeCondensedGraph keeping the float64 result next to the float32 wire edges."""
        b, g = self.buf, self.pg
        want = b.out_closures.get(peer)
        idx = b._index_of_ids(want)
        idx = idx[idx >= 0]
        if len(idx) < 2:
            b.out_condensed[peer] = np.zeros(0, dtype=EDGE_DTYPE)
            self.gauge[peer] = (None, (np.zeros(0, dtype=np.int64), np.zeros((0, 3)), np.zeros((0, 6))))
            return self.gauge[peer][1], None
        gauge = int(idx[select_gauge_centroid(g.poses[idx, :2])])
        m = b.my_edge_mask()
        if getattr(self, "optimal_gauge", False):
            # selectOptimalGauge (condensed_graph_buffer.cpp:252-288) with computeOverallUncertainty (:172-180)
            best = np.inf
            self.uncertainties = {}
            for cand in idx:
                _, _, iu_c, _ = self.ctx.condense(g.poses, g.edge_from[m], g.edge_to[m], g.meas[m], g.info[m], int(cand), idx.astype(np.int32))
                u = 0.0
                for a, bb, c, d, e, f in iu_c:
                    u += 1.0 / (a * (d * f - e * e) - bb * (bb * f - e * c) + c * (bb * e - d * c))
                self.uncertainties[int(g.ids[cand])] = u
                if u < best:
                    best, gauge = u, int(cand)
        to, est, iu, _ = self.ctx.condense(g.poses, g.edge_from[m], g.edge_to[m], g.meas[m], g.info[m], gauge, idx.astype(np.int32))
        e = np.zeros(len(to), dtype=EDGE_DTYPE)
        e["from"] = g.ids[gauge]; e["to"] = g.ids[to]
        e["est"] = est.astype(np.float32); e["info"] = iu.astype(np.float32)
        b.out_condensed[peer] = e
        self.gauge[peer] = (int(g.ids[gauge]), (g.ids[to].copy(), est, iu))
        return self.gauge[peer][1], gauge

    def condensed(self, peer):
        gid, (to, est, iu) = self.gauge.get(peer, (None, (np.zeros(0, dtype=np.int64), np.zeros((0, 3)), np.zeros((0, 6)))))
        return gid, to, est, iu

    def pack_host(self):
        return self.buf.pack()

    def ingest_host(self, recv):
        recv = np.asarray(recv, dtype=np.uint8).reshape(self.n_robots, -1)
        n = np.zeros(self.n_robots, dtype=np.int32)
        for s in range(self.n_robots):
            sender, edges, clos = self.buf.unpack(recv[s])
            if sender != s or s == self.robot:
                continue
            n[s] = self.buf.ingest_from(s, edges, clos)
        return n

    def closures(self, peer, which="out"):
        d = self.buf.out_closures if which == "out" else self.buf.in_closures
        return np.asarray(d.get(peer, np.zeros(0, dtype=np.int64)), dtype=np.int32)

    def message_for(self, peer):
        """constructCondensedGraphMessage (mr_graph_slam.cpp:607-670)."""
        from cg_mrslam_amd.messages import CondensedGraphMessage
        b = self.buf
        e = b.out_condensed.get(peer, np.zeros(0, dtype=EDGE_DTYPE))
        c = b.in_closures.get(peer)
        if not len(e) and (c is None or not len(c)):
            return None
        return CondensedGraphMessage(self.robot, e.copy(), np.zeros(0, dtype=np.int32) if c is None else c.astype(np.int32))

    def message_from(self, msg):
        """addInterRobotData(CondensedGraphMessage*) (mr_graph_slam.cpp:331-395)."""
        b = self.buf
        clos = np.asarray(msg.closures, dtype=np.int64)
        known = (b._index_of_ids(clos) >= 0).any() if len(clos) else False
        n = b.ingest_from(msg.robotId, np.asarray(msg.edges, dtype=EDGE_DTYPE), clos)
        if known:
            self.computeCondensedGraph(msg.robotId)
        return n

    def received_edges(self, peer):
        g, b = self.pg, self.buf
        m = b.in_edge_src == peer
        return g.ids[g.edge_from[m]], g.ids[g.edge_to[m]], g.meas[m], g.info[m]
