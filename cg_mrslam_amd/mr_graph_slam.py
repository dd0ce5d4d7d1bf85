"""Multi-robot key-frame driver: the ``cg_mrslam`` node around the GPU kernels (BASELINE config C4).

Host-side mirror of
  * ``cg_mrslam`` main loop                         src/cg_mrslam.cpp:206-259          -> ``run_cg_mrslam`` / ``RobotLoop``
  * ``MRGraphSLAM``                                 src/mrslam/mr_graph_slam.{h,cpp}   -> ``MRGraphSLAMDriver``
        checkInterRobotClosures :60-112, updateInterRobotClosures :114-116, addInterRobotData(ComboMessage) :118-252,
        findInterRobotConstraints :254-329, addInterRobotData(CondensedGraphMessage) :331-395,
        constructComboMessage :564-605, constructCondensedGraphMessage :607-670
  * ``MRClosureBuffer``                             src/mrslam/mr_closure_buffer.cpp   -> ``MRClosureBuffer``
  * ``GraphComm`` (SIM modality)                    src/mrslam/graph_comm.cpp:62-208   -> ``GraphCommSim`` / ``GraphCommRanks``
from recorded arrays (odometry, ranges, ground truth) instead of ROS topics and UDP sockets.

Numerics all sit behind the C ABI: the robot's graph, its Gauss-Newton solve, the condensed graphs and the wire records
are a ``condensed.RobotGraph`` (``cgmr_graph_*``), inter-robot matches come from ``globalMatching`` / ``verifyMatching``
(batched: every candidate of a key frame in one launch per search level), the single-robot front
end is ``slam.GraphSLAMDriver``.  This module is bookkeeping: which vertex is in which buffer.

Timing model.  The reference runs three communication threads beside the 10 Hz main loop (send every 150 ms, receive,
process the queue).  Here one communication cycle follows every tick of the main loop, in a fixed order (all sends, then
every robot drains its queue in sender order); with one rank per robot the cycle is ONE all-gather of fixed-size byte
buffers (``GraphCommRanks``).  Messages travel as the reference's own byte strings (``messages.py``), so doubles are
narrowed to float exactly where the reference narrows them.

Where the reference iterates ``std::set<Vertex*>`` (address order) this code iterates in vertex-id order; maps keyed by
robot or vertex id iterate in key order as ``std::map`` does.
"""
from __future__ import annotations

import math

import numpy as np

from .matcher import _se2_inv, _se2_mul
from .messages import MAX_LENGTH_MSG, ComboMessage, CondensedGraphMessage, from_bytes
from .slam import ClosureBuffer, GraphSLAMDriver

INTER_ROBOT_INFO = np.array([100.0, 0, 0, 100.0, 0, 1000.0])     # mr_graph_slam.cpp:234-236, 310-312
SIM_COMM_RANGE = 5.0                                             # graph_comm.h:47
PEER_LASER_MAX_RANGE = 8.0                                       # mr_graph_slam.cpp:165: received scans are cut at 8 m


class MRClosureBuffer:
    """mr_closure_buffer.cpp: one ``ClosureBuffer`` per peer robot (vertices are the peer's vertex ids)."""

    def __init__(self):
        self.mrClosures = {}   # noqa: N815

    def findClosuresRobot(self, robotId):   # noqa: N802,N803
        return self.mrClosures.get(robotId)

    def insert(self, closures: ClosureBuffer, robotId):   # noqa: N803
        prev = self.findClosuresRobot(robotId)
        if prev is not None:
            for v in closures.vertex_ids():
                prev.addVertex(v)
            for e in closures.edges:
                prev.addEdge(e)
        else:
            cb = ClosureBuffer()
            cb.vertices = [list(q) for q in closures.vertices]
            cb.edges = list(closures.edges)
            self.mrClosures[robotId] = cb

    def remove(self, closures: ClosureBuffer, robotId):   # noqa: N803
        prev = self.findClosuresRobot(robotId)
        if prev is None:
            return
        for v in closures.vertex_ids():
            prev.removeVertex(v)
        for e in closures.edges:
            prev.removeEdge(e)
        if not prev.vertices:
            del self.mrClosures[robotId]

    def update(self, windowSize):   # noqa: N803
        for r in sorted(self.mrClosures):
            cb = self.mrClosures[r]
            cb.updateList(windowSize)
            if not cb.vertices:
                del self.mrClosures[r]

    def size(self):
        return len(self.mrClosures)


class MRGraphSLAMDriver(GraphSLAMDriver):
    """``MRGraphSLAM``: the single-robot front end plus inter-robot closures and condensed graphs.

    ``robot_graph``: a ``condensed.RobotGraph`` (or an object with its interface) created for (idRobot, nRobots); it holds
    the optimiser's copy of the graph, so ``optimize`` runs there and the estimates are read back."""

    def __init__(self, ctx, close_matcher, lc_matcher, robot_graph, idRobot, nRobots, **kw):   # noqa: N803
        self.rg = robot_graph
        self.nRobots = nRobots   # noqa: N815
        self._n_recv = 0                       # received condensed edges at the tail of the host edge arrays
        super().__init__(ctx, close_matcher, lc_matcher, idRobot=idRobot, **kw)
        self.maxScoreMR, self.minInliersMR, self.windowMRLoopClosure = self.maxScore, self.minInliers, self.windowLoopClosure   # noqa: N815
        self.detectRobotInRange = False   # noqa: N815
        self.interRobotClosures = MRClosureBuffer()    # matched vertices + edges   # noqa: N815
        self.interRobotVertices = MRClosureBuffer()    # vertices not matched yet   # noqa: N815
        self.peer = {}                                 # peer vertex id -> {"pose", "ranges"} while it is not in the graph

    def setInterRobotClosureParams(self, maxScoreMR, minInliersMR, windowMRLoopClosure):   # noqa: N802,N803
        self.maxScoreMR, self.minInliersMR, self.windowMRLoopClosure = maxScoreMR, minInliersMR, windowMRLoopClosure

    def setDetectRobotInRange(self, flag):   # noqa: N802
        self.detectRobotInRange = bool(flag)

    # ------------------------------------------------------------------ graph bookkeeping, mirrored into the robot graph
    def _pop_received(self):
        n = self._n_recv
        if n == 0:
            return None
        g = self.g
        tail = (g.edge_from[-n:], g.edge_to[-n:], g.meas[-n:], g.info[-n:], g.edge_level[-n:])
        g.edge_from, g.edge_to, g.meas, g.info, g.edge_level = g.edge_from[:-n], g.edge_to[:-n], g.meas[:-n], g.info[:-n], g.edge_level[:-n]
        del self.edge_ids[-n:], self.edge_kind[-n:]
        self._n_recv = 0
        return tail

    def _push_received(self, tail):
        if tail is None:
            return
        g = self.g
        g.edge_from, g.edge_to = np.concatenate([g.edge_from, tail[0]]), np.concatenate([g.edge_to, tail[1]])
        g.meas, g.info = np.vstack([g.meas, tail[2]]), np.vstack([g.info, tail[3]])
        g.edge_level = np.concatenate([g.edge_level, tail[4]])
        n = len(tail[0])
        self.edge_ids += [-1] * n
        self.edge_kind += ["cond"] * n
        self._n_recv = n

    def _sync_received(self):
        """The condensed edges currently held from the peers (level-0 edges of the graph, condensed_graph_buffer.cpp:
        487-510) as the tail of the host edge arrays: the graph searches and the covariance estimate see them."""
        self._pop_received()
        f, t, m, i = [], [], [], []
        idx = self._id_index
        for p in range(self.nRobots):
            if p == self.idRobot:
                continue
            pf, pt, pm, pi = self.rg.received_edges(p)
            if len(pf) == 0:
                continue
            f.append(np.fromiter((idx[int(a)] for a in pf), dtype=np.int32, count=len(pf)))
            t.append(np.fromiter((idx[int(b)] for b in pt), dtype=np.int32, count=len(pt)))
            m.append(pm)
            i.append(pi)
        if f:
            f, t = np.concatenate(f), np.concatenate(t)
            self._push_received((f, t, np.concatenate(m).reshape(-1, 3), np.concatenate(i).reshape(-1, 6), np.zeros(len(f), dtype=np.int32)))

    def _add_vertex(self, vid, pose, fixed, ranges):
        idx = super()._add_vertex(vid, pose, fixed, ranges)
        self.rg.add_vertices([vid], np.asarray(pose, dtype=np.float64).reshape(1, 3), [1 if fixed else 0])
        return idx

    def _add_edge(self, i, j, meas, info, kind, eid):
        tail = self._pop_received()                    # own edges stay in front of the received ones
        super()._add_edge(i, j, meas, info, kind, eid)
        self.rg.add_edges([int(self.g.ids[i])], [int(self.g.ids[j])], np.asarray(meas, dtype=np.float64).reshape(1, 3),
                          np.asarray(info, dtype=np.float64).reshape(1, 6))
        self._push_received(tail)

    def optimize(self, nrunnings: int) -> None:
        rc, chi2 = self.rg.optimize(int(nrunnings))
        self.g.poses[:] = self.rg.poses()
        self.last_status, self.last_chi2 = rc, chi2

    # ------------------------------------------------------------------ helpers
    def _peer_pose(self, vid):
        idx = self._index_of_id(vid)               # once accepted, the vertex lives in the graph and is optimised there
        return self.g.poses[idx].copy() if idx is not None else self.peer[vid]["pose"]

    def _reference_vset(self, ref, gap_back, gap_fwd):
        ref_id = int(self.g.ids[ref])
        vset = [ref]
        for j in range(1, gap_back + 1):
            v = self._index_of_id(ref_id - j)
            if v is None:
                break
            vset.append(v)
        for j in range(1, gap_fwd + 1):
            v = self._index_of_id(ref_id + j)
            if v is None:
                break
            vset.append(v)
        return vset

    def _global_matching(self, jobs):
        m = self.lc_matcher
        if len(jobs) > 1 and hasattr(m, "globalMatchingBatch"):
            return m.globalMatchingBatch(jobs, self.maxScoreMR)
        return [m.globalMatching(*j, self.maxScoreMR) for j in jobs]

    def _verify_matching(self, jobs, trel):
        m = self.lc_matcher
        if len(jobs) > 1 and hasattr(m, "verifyMatchingBatch"):
            return m.verifyMatchingBatch(jobs, trel)
        return [m.verifyMatching(*j, t) for j, t in zip(jobs, trel)]

    @staticmethod
    def _closure(ref_id, vid, transf):
        c = ClosureBuffer()
        c.addVertex(vid)
        c.addEdge({"from": ref_id, "to": vid, "meas": np.asarray(transf, dtype=np.float64).copy(), "info": INTER_ROBOT_INFO,
                   "added": False})
        return c

    # ------------------------------------------------------------------ mr_graph_slam.cpp:60-116
    def checkInterRobotClosures(self):   # noqa: N802
        for robot_id in sorted(self.interRobotClosures.mrClosures):
            cb = self.interRobotClosures.mrClosures[robot_id]
            if not cb.checkList(self.windowMRLoopClosure):
                continue
            local = sorted(set(cb.vertex_ids()))
            poses = {vid: self._peer_pose(vid) for vid in local}
            for e in cb.edges:
                poses[e["from"]] = self.g.poses[self._index_of_id(e["from"])].copy()
                poses.setdefault(e["to"], self._peer_pose(e["to"]))
            self.lcc.init(poses, local, cb.edges, self.inlierThreshold)
            self.lcc.check()
            self.log.append(("mr_lcc", robot_id, self.lcc.inliers(), self.lcc.chi2()))
            if self.lcc.inliers() < self.minInliersMR:
                continue
            in_closures = []
            for e, chi in self.lcc.closures():
                if not chi < self.inlierThreshold:
                    continue
                self._running_edge_id += 1                       # setId runs for every inlier, new or not (:83)
                vto = e["to"]
                idx = self._index_of_id(vto)
                if idx is None:
                    idx = self._add_vertex(vto, self.peer[vto]["pose"], False, self.peer[vto]["ranges"])
                elif idx not in self.lasers and vto in self.peer:
                    self.lasers[idx] = self.peer[vto]["ranges"]
                if not e["added"]:                               # HyperGraph::addEdge refuses an edge twice
                    e["added"] = True
                    self._add_edge(self._index_of_id(e["from"]), idx, e["meas"], INTER_ROBOT_INFO, "mr",
                                   self._running_edge_id + self.baseId)
                in_closures.append(vto)
            if in_closures:
                self.rg.insertInClosure(robot_id, np.asarray(sorted(set(in_closures))))

    def updateInterRobotClosures(self):   # noqa: N802
        self.interRobotClosures.update(self.windowMRLoopClosure)

    # ------------------------------------------------------------------ mr_graph_slam.cpp:118-252
    def _add_combo(self, cmsg: ComboMessage, ref_vertex):
        robot = cmsg.robotId
        vset = []
        for rec in cmsg.vertices:
            vid = int(rec["id"])
            vest = rec["estimate"].astype(np.float64)
            if self._index_of_id(vid) is not None:               # already in the graph
                continue
            known = False
            for buf in (self.interRobotClosures, self.interRobotVertices):
                cb = buf.findClosuresRobot(robot)
                if cb is not None and cb.findVertex(vid):
                    self.peer[vid]["pose"] = vest                # update estimate
                    vset.append(vid)
                    known = True
                    break
            if known:
                continue
            if vid == cmsg.nodeId:                               # new vertex, comes with its scan
                r = np.array(cmsg.readings, dtype=np.float32)
                r[r >= PEER_LASER_MAX_RANGE] = np.float32(2.0 * self.lc_matcher.cfg.max_range)   # LaserParameters(.., 8.0, ..)
                r.setflags(write=False)
                self.peer[vid] = {"pose": vest, "ranges": r}
                vset.append(vid)
        if not vset or cmsg.nodeId not in vset:
            # (the reference would match an empty VertexSE2 here, :214-221; a node is sent once, so this does not occur)
            return
        v = cmsg.nodeId
        refs = self._reference_vset(ref_vertex, 10, 10)
        order, ref_scans = self._scans(refs)
        cur_order = sorted(vset)
        cur_scans = [(self.peer[q]["ranges"], self._peer_pose(q)) for q in cur_order]
        job = (ref_scans, order.index(ref_vertex), cur_scans, cur_order.index(v))
        found, transf = self._global_matching([job])[0]
        ref_id = int(self.g.ids[ref_vertex])
        self.log.append(("combo", robot, v, ref_id, bool(found)))
        if found:
            if self.detectRobotInRange:
                ok, score = self._verify_matching([job], [transf])[0]
                self.log.append(("verify", robot, v, bool(ok), float(score)))
                if not ok:
                    return
            self.interRobotClosures.insert(self._closure(ref_id, v, transf), robot)
        else:
            c = ClosureBuffer()
            c.addVertex(v)
            self.interRobotVertices.insert(c, robot)

    # ------------------------------------------------------------------ mr_graph_slam.cpp:331-395
    def _add_condensed(self, gmsg: CondensedGraphMessage):
        n = self.rg.message_from(gmsg)
        if n:
            self._sync_received()
        self.log.append(("cond_in", gmsg.robotId, len(gmsg.closures), len(gmsg.edges), n))

    def addInterRobotData(self, msg, refVertex=None):   # noqa: N802,N803  (mr_graph_slam.cpp:485-501)
        if isinstance(msg, ComboMessage):
            self._add_combo(msg, self._last_vertex if refVertex is None else refVertex)
        elif isinstance(msg, CondensedGraphMessage):
            self._add_condensed(msg)

    # ------------------------------------------------------------------ mr_graph_slam.cpp:254-329
    def findInterRobotConstraints(self):   # noqa: N802
        last = self._last_vertex
        last_id = int(self.g.ids[last])
        refs = self._reference_vset(last, 20, 0)
        order, ref_scans = self._scans(refs)
        ri = order.index(last)
        todo = []                                                # every vertex of every peer that is still unmatched
        for robot_id in sorted(self.interRobotVertices.mrClosures):
            for vid in sorted(set(self.interRobotVertices.mrClosures[robot_id].vertex_ids())):
                todo.append((robot_id, vid))
        jobs = [(ref_scans, ri, [(self.peer[vid]["ranges"], self._peer_pose(vid))], 0) for _, vid in todo]
        results = self._global_matching(jobs) if jobs else []
        hit = [k for k, (found, _) in enumerate(results) if found]
        verified = {}
        if self.detectRobotInRange and hit:
            ver = self._verify_matching([jobs[k] for k in hit], [results[k][1] for k in hit])
            verified = {k: ok for k, (ok, _) in zip(hit, ver)}
            self.log += [("verify", todo[k][0], todo[k][1], bool(ok), float(sc)) for k, (ok, sc) in zip(hit, ver)]
        for k, (robot_id, vid) in enumerate(todo):
            found, transf = results[k]
            self.log.append(("mr_match", robot_id, vid, last_id, bool(found)))
            if not found or (self.detectRobotInRange and not verified[k]):
                continue
            closure = self._closure(last_id, vid, transf)
            self.interRobotClosures.insert(closure, robot_id)
            self.interRobotVertices.remove(closure, robot_id)
        self.checkInterRobotClosures()
        self.updateInterRobotClosures()
        self.interRobotVertices.update(self.windowMRLoopClosure)
        for vid in [q for q in self.peer if self._index_of_id(q) is None and not self._buffered(q)]:
            del self.peer[vid]                                   # dropped from both windows: the scan is not needed any more

    def _buffered(self, vid):
        return any(cb.findVertex(vid) for buf in (self.interRobotClosures, self.interRobotVertices) for cb in buf.mrClosures.values())

    # ------------------------------------------------------------------ mr_graph_slam.cpp:564-670
    def constructComboMessage(self):   # noqa: N802
        last = self._last_vertex
        last_id = int(self.g.ids[last])
        idx = [last]
        for i in range(1, 5):                                    # nVertices = 5
            v = self._index_of_id(last_id - i)
            if v is None:
                break
            idx.append(v)
        idx.sort(key=lambda q: self.g.ids[q])                    # VertexIDMap order
        cfg = self.close_matcher.cfg
        return ComboMessage(self.idRobot, [int(self.g.ids[q]) for q in idx], self.g.poses[idx], nodeId=last_id,
                            readings=self.lasers[last], minangle=cfg.angle_min, angleincrement=cfg.angle_inc,
                            maxrange=cfg.max_range, accuracy=0.1)

    def constructCondensedGraphMessage(self, idRobotTo):   # noqa: N802,N803
        return self.rg.message_for(idRobotTo)


# ---------------------------------------------------------------------------------------------------------------------
# cg_mrslam.cpp:206-259 and graph_comm.cpp, SIM modality

class RobotLoop:
    """One robot's main loop over recorded odometry and scans (cg_mrslam.cpp:206-259); ``tick(k)`` is one pass."""

    def __init__(self, slam: MRGraphSLAMDriver, odom, scans, initial_pose, linearUpdate=0.25, angularUpdate=math.pi / 4,   # noqa: N803
                 iterations=5):
        self.slam, self.odom, self.scans = slam, np.asarray(odom, dtype=np.float64), scans
        self.linearUpdate, self.angularUpdate, self.iterations = linearUpdate, angularUpdate, iterations   # noqa: N815
        self.curr_est = np.asarray(initial_pose, dtype=np.float64).copy()     # SIM: the ground-truth start (cg_mrslam.cpp:147-148)
        self.odom_k1 = self.odom[0].copy()
        slam.setInitialData(self.curr_est, scans[0])
        self.key_frames = 1

    def tick(self, k):
        s = self.slam
        rel = _se2_mul(_se2_inv(self.odom_k1), self.odom[k])
        self.curr_est = _se2_mul(self.curr_est, rel)
        self.odom_k1 = self.odom[k].copy()
        lastp = s.g.poses[s.lastVertex()]
        if (math.hypot(lastp[0] - self.curr_est[0], lastp[1] - self.curr_est[1]) > self.linearUpdate or
                abs(lastp[2] - self.curr_est[2]) > self.angularUpdate):
            s.addDataSM(self.curr_est, self.scans[k])
            s.findConstraints()
            s.findInterRobotConstraints()
            s.optimize(self.iterations)
            self.curr_est = s.g.poses[s.lastVertex()].copy()
            self.key_frames += 1
            return True
        return False

    def finish(self):
        self.slam.optimize(self.iterations)


def _in_range(truth_a, truth_b, comm_range):
    return math.hypot(truth_a[0] - truth_b[0], truth_a[1] - truth_b[1]) < comm_range   # distanceSE2 on the ground truth


class _Sender:
    """``GraphComm::sendToThrd`` (graph_comm.cpp:126-155) for one robot: the messages of one cycle as (dest, bytes)."""

    def __init__(self, slam, comm_range):
        self.slam, self.comm_range = slam, comm_range
        self.last_sent = -1
        self.bytes_sent = 0

    def outbox(self, truth_now):
        s = self.slam
        me = s.idRobot
        to = [r for r in range(s.nRobots) if r != me and _in_range(truth_now[me], truth_now[r], self.comm_range)]
        out = []
        if not to:
            return out
        last_id = int(s.g.ids[s.lastVertex()])
        if last_id != self.last_sent:
            self.last_sent = last_id
            b = s.constructComboMessage().to_bytes()
            if b:
                out += [(r, b) for r in to]
        for r in to:
            gmsg = s.constructCondensedGraphMessage(r)
            b = gmsg.to_bytes() if gmsg is not None else None
            if b:
                out.append((r, b))
        self.bytes_sent += sum(len(b) for _, b in out)
        return out


class GraphCommSim:
    """All robots in one process: SIM modality of ``GraphComm`` (robots talk when their ground-truth poses are closer
    than ``SIM_COMM_RANGE``, graph_comm.cpp:62-64,76-84), one cycle per call."""

    def __init__(self, slams, comm_range=SIM_COMM_RANGE):
        self.slams = slams
        self.senders = [_Sender(s, comm_range) for s in slams]
        self.delivered = 0

    def cycle(self, truth_now, pool=None):
        """pool (a ``concurrent.futures`` executor, optional): the robots build their outboxes side by side, then process
        their queues side by side -- what one process per robot does; every robot needs a context of its own then.  Same
        messages in the same per-robot order either way."""
        run = (lambda f, xs: list(pool.map(f, xs))) if pool is not None else (lambda f, xs: [f(x) for x in xs])
        queues = [[] for _ in self.slams]
        for out in run(lambda snd: snd.outbox(truth_now), self.senders):
            for dest, b in out:
                queues[dest].append(b)

        def process(sq):                                         # receiveFromThrd + processQueueThrd
            s, q = sq
            for b in q:
                s.addInterRobotData(from_bytes(b), s.lastVertex())
            return len(q)
        self.delivered += sum(run(process, list(zip(self.slams, queues))))


def pack_outbox(out, cap_bytes):
    """(dest, bytes) list -> one fixed-size uint8 record: int32 count, then per message {int32 dest, int32 len, payload}."""
    buf = np.zeros(cap_bytes, dtype=np.uint8)
    o = 4
    buf[:4] = np.frombuffer(np.int32(len(out)).tobytes(), dtype=np.uint8)
    for dest, b in out:
        if o + 8 + len(b) > cap_bytes:
            raise ValueError("outbox exceeds its all-gather slot")
        buf[o:o + 8] = np.frombuffer(np.array([dest, len(b)], dtype=np.int32).tobytes(), dtype=np.uint8)
        buf[o + 8:o + 8 + len(b)] = np.frombuffer(b, dtype=np.uint8)
        o += 8 + len(b)
    return buf


def unpack_outbox(buf):
    n = int(buf[:4].view(np.int32)[0])
    o, out = 4, []
    for _ in range(n):
        dest, ln = (int(v) for v in buf[o:o + 8].view(np.int32))
        out.append((dest, buf[o + 8:o + 8 + ln].tobytes()))
        o += 8 + ln
    return out


class GraphCommRanks:
    """One rank per robot: a communication cycle is ONE all-gather of every rank's outbox (fixed-size slot:
    one ComboMessage + one CondensedGraphMessage per peer); each rank then drains what is addressed to it in sender
    order.  Backend nccl (= RCCL): the slots live in HBM; gloo: host tensors."""

    def __init__(self, slam, comm_range=SIM_COMM_RANGE, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.slam = slam
        self.sender = _Sender(slam, comm_range)
        if dist.get_world_size(group) != slam.nRobots:
            raise ValueError("one rank per robot")
        self.device = device if device is not None else torch.device("cpu")
        self.slot, self.recv = 0, None
        self.delivered = 0

    def _allocate(self):
        s = self.slam
        r = s.nRobots
        n_beams = len(s.lasers[s.lastVertex()])
        combo = 8 + 8 + 5 * 16 + 4 + 8 + 4 * n_beams + 16
        cond = 8 + 8 + 44 * s.rg.cap + 8 + 4 * s.rg.cap
        slot = 4 + (r - 1) * (16 + min(combo, MAX_LENGTH_MSG) + min(cond, MAX_LENGTH_MSG))
        self.slot = (slot + 255) // 256 * 256
        self.recv = self.torch.empty(r * self.slot, dtype=self.torch.uint8, device=self.device)

    def cycle(self, truth_now):
        torch, s = self.torch, self.slam
        if self.recv is None:
            self._allocate()                                     # sized by the first vertex' scan
        send = torch.from_numpy(pack_outbox(self.sender.outbox(truth_now), self.slot)).to(self.device)
        self.dist.all_gather_into_tensor(self.recv, send, group=self.group)
        host = self.recv.cpu().numpy().reshape(s.nRobots, self.slot)
        for src in range(s.nRobots):
            if src == s.idRobot:
                continue
            for dest, b in unpack_outbox(host[src]):
                if dest == s.idRobot:
                    s.addInterRobotData(from_bytes(b), s.lastVertex())
                    self.delivered += 1


def run_cg_mrslam(slams, trajectories, comm=None, linearUpdate=0.25, angularUpdate=math.pi / 4, iterations=5, n_steps=None,   # noqa: N803
                  concurrent=False):
    """``cg_mrslam -modality sim`` for several robots in one process: every robot's main loop ticks through its recorded
    trajectory (``odom``, ``scans``, ``truth``), a communication cycle after every tick.  Returns the ``RobotLoop``s."""
    loops = [RobotLoop(s, tr["odom"], tr["scans"], tr["truth"][0], linearUpdate, angularUpdate, iterations)
             for s, tr in zip(slams, trajectories)]
    comm = comm or GraphCommSim(slams)
    n = min(len(tr["odom"]) for tr in trajectories) if n_steps is None else n_steps
    if not concurrent:
        for k in range(1, n):
            for lp in loops:
                lp.tick(k)
            comm.cycle([tr["truth"][k] for tr in trajectories])
        for lp in loops:
            lp.finish()
        return loops
    # one thread per robot between the communication cycles (the C-ABI calls release the interpreter lock): the robots of the
    # reference are processes of their own.  Every robot must have been built on a context of its own.
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(loops)) as pool:
        for k in range(1, n):
            list(pool.map(lambda lp: lp.tick(k), loops))
            comm.cycle([tr["truth"][k] for tr in trajectories], pool=pool)
        list(pool.map(lambda lp: lp.finish(), loops))
    return loops


def run_cg_mrslam_rank(slam, trajectories, comm=None, linearUpdate=0.25, angularUpdate=math.pi / 4, iterations=5, n_steps=None):   # noqa: N803
    """The same with one rank per robot (``torch.distributed`` initialised, rank = robot id): this rank runs robot
    ``slam.idRobot``; ``trajectories`` holds every robot's ground truth (the SIM modality's range test needs it)."""
    tr = trajectories[slam.idRobot]
    loop = RobotLoop(slam, tr["odom"], tr["scans"], tr["truth"][0], linearUpdate, angularUpdate, iterations)
    comm = comm or GraphCommRanks(slam)
    n = min(len(t["truth"]) for t in trajectories) if n_steps is None else n_steps
    for k in range(1, n):
        loop.tick(k)
        comm.cycle([t["truth"][k] for t in trajectories])
    loop.finish()
    return loop


__all__ = ["MRClosureBuffer", "MRGraphSLAMDriver", "RobotLoop", "GraphCommSim", "GraphCommRanks", "run_cg_mrslam",
           "run_cg_mrslam_rank", "pack_outbox", "unpack_outbox", "SIM_COMM_RANGE", "INTER_ROBOT_INFO"]
