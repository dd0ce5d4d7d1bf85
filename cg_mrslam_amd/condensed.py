"""Condensed-graph construction and inter-robot exchange for the multi-robot path.

Host-side mirror of the reference's
  * ``CondensedGraphBuffer``  (src/mrslam/condensed_graph/condensed_graph_buffer.{h,cpp}): which of my
    vertices each peer asked for (out-closures), which foreign vertices I ask for (in-closures), the edges
    I built for / received from each peer, ``getMyEdges``, ``selectGaugeCentroid``,
    ``computeCondensedGraph`` and ``insertEdgesFromRobot`` (replace-on-receive);
  * the wire structs of ``msg_factory.h`` (``EdgeArrayMessage::ESE2Data``: int from, int to, 3 + 6 doubles
    narrowed to float32 on the wire, src/mrslam/msg_factory.h:78-112,200-218): 44 bytes per edge;
  * ``GraphComm``'s pairwise UDP send/receive (src/mrslam/graph_comm.cpp:103-193), replaced by ONE
    all-gather per round of a fixed-capacity byte buffer per rank over RCCL/xGMI (``torch.distributed`` with
    backend "nccl" on the GPUs, "gloo" in the CPU tests).

The numeric work (``CondensedGraphCreator::compute``) runs on the GPU through ``Context.condense``.
"""
from __future__ import annotations

import numpy as np

from .graph import PoseGraph

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
            if e is not None and len(e):
                n = min(len(e), cap)
                edges[p, :n] = e[:n]
                hdr[2 + p] = n
            c = self.in_closures.get(p)
            if c is not None and len(c):
                n = min(len(c), cap)
                clos[p, :n] = c[:n]
                hdr[2 + R + p] = n
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
            self.insertOutClosure(sender, closures)
            self.insertEdgesFromRobot(sender, edges)
        return int(recv.numel())
