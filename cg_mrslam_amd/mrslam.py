"""Round driver of the multi-robot path on synthetic worlds (BASELINE.json configs C4 / C5).

The reference's ``cg_mrslam`` node (src/cg_mrslam.cpp:206-259) adds a key frame, looks for constraints, optimises and
-- from its communication threads (src/mrslam/graph_comm.cpp:126-208) -- sends every robot in range a ComboMessage
with the condensed graph built for it and the list of vertices it wants condensed in return
(src/mrslam/mr_graph_slam.cpp:527-562, 607-670); incoming messages are digested by
``MRGraphSLAM::addInterRobotData`` (src/mrslam/mr_graph_slam.cpp:331-395).  Here the same protocol runs in rounds, one
round every ``chunk`` new vertices (C5: 50), one rank = one robot = one GPU:

    grow           the robot's own sub-graph gains ``chunk`` vertices, their odometry / loop-closure edges and the
                   inter-robot closures that became possible (both end points exist): a foreign vertex + an edge
                   own -> foreign, information diag(100, 100, 1000) (mr_graph_slam.cpp:234-236), a closure request
    optimize(5)    local solve on own + received level-0 edges
    ingest         what the previous round's all-gather delivered (one round of latency, like a UDP message that
                   arrives while the robot is busy): requests -> out-closures, newest edge set per peer replaces the old
    condense       one star of condensed edges per peer that has asked (own edges only)
    exchange       one all-gather of the 44-byte/edge wire buffers, issued on a side stream: it overlaps with the
                   next round's grow + optimize

``RobotRounds`` drives one robot through any object with the ``RobotGraph`` interface (the product:
``cg_mrslam_amd.condensed.RobotGraph``; the tests also run ``tests/ref_condensed.RefRobotGraph`` on the CPU oracle
through the very same rounds and compare edge by edge).  All numerics are behind that interface.
"""
from __future__ import annotations

import numpy as np

from . import synth

INTER_ROBOT_INFO = np.array([100.0, 0, 0, 100.0, 0, 1000.0])     # mr_graph_slam.cpp:234-236, 310-312


class RobotWorld:
    """Robot ``r``'s share of a ``synth.make_multi_robot`` world, served in increments of ``chunk`` vertices."""

    def __init__(self, robots, r, chunk=50, base_id=10000, closures=True):
        """``closures=False``: the robot alone in the same world -- its own vertices and edges, nobody to close loops with
        (the one-rank reference of a weak-scaling figure: the same sub-graph without the peers)."""
        g = robots[r]
        self.r, self.chunk, self.base_id = r, chunk, base_id
        self.n_own = int(g["n_own"])
        self.n_rounds = (self.n_own + chunk - 1) // chunk
        self.ids = g["ids"][:self.n_own].astype(np.int64)
        self.truth0 = g["truth"][0].copy()
        n_e = len(g["edge_from"])
        self.ef, self.et = g["edge_from"].astype(np.int64), g["edge_to"].astype(np.int64)
        self.meas, self.info = g["meas"], g["info"]
        # own edge k becomes available in the round that adds its later end point
        self.e_round = np.maximum(self.ef, self.et) // chunk
        self.e_order = np.argsort(self.e_round, kind="stable")
        self.e_ptr = np.searchsorted(self.e_round[self.e_order], np.arange(self.n_rounds + 1))
        # inter-robot closures: own vertex i -> copy of peer p's vertex j; possible once both exist
        ci = g["ef_all"][n_e:].astype(np.int64)
        fid = g["ids"][g["et_all"][n_e:]].astype(np.int64)
        self.c_own, self.c_fid = ci, fid
        self.c_meas, self.c_info = g["meas_all"][n_e:], g["info_all"][n_e:]
        if not closures:
            keep = np.zeros(len(ci), dtype=bool)
            ci, fid = ci[keep], fid[keep]
            self.c_own, self.c_fid = ci, fid
            self.c_meas, self.c_info = self.c_meas[keep], self.c_info[keep]
        self.c_round = np.maximum(ci, fid % base_id) // chunk
        self.c_order = np.argsort(self.c_round, kind="stable")
        self.c_ptr = np.searchsorted(self.c_round[self.c_order], np.arange(self.n_rounds + 1))

    def increment(self, t):
        """Round ``t`` (0-based): (first new own vertex, one past the last, own-edge indices, closure indices)."""
        v0, v1 = t * self.chunk, min(self.n_own, (t + 1) * self.chunk)
        return v0, v1, self.e_order[self.e_ptr[t]:self.e_ptr[t + 1]], self.c_order[self.c_ptr[t]:self.c_ptr[t + 1]]


def _dead_reckon(start, increments):
    """start * inc[0], start * inc[0] * inc[1], ...: ``synth.se2_compose`` step by step on scalars (same operations in the
    same order, bit for bit -- tests/test_host_cpu.py -- without an array round trip per pose: a round's 50 new poses were
    0.4 ms of a 2.5 ms solo round)."""
    x, y, th = float(start[0]), float(start[1]), float(start[2])
    two_pi = 2 * np.pi
    out = np.empty((len(increments), 3), dtype=np.float64)
    for k, (bx, by, bt) in enumerate(np.asarray(increments, dtype=np.float64).tolist()):
        c, s = float(np.cos(np.float64(th))), float(np.sin(np.float64(th)))
        x, y = x + c * bx - s * by, y + s * bx + c * by
        t = th + bt
        th = t if (-np.pi <= t < np.pi) else t - two_pi * float(np.floor((t + np.pi) / two_pi))
        out[k, 0], out[k, 1], out[k, 2] = x, y, th
    return list(out)


class RobotRounds:
    """One robot of the round protocol above on ``graph`` (``RobotGraph`` interface)."""

    def __init__(self, graph, world: RobotWorld, iterations: int = 5):
        self.g, self.w, self.iterations = graph, world, iterations
        self.t = 0
        self.foreign = {}             # foreign vertex id -> True once it is in the graph
        self.n_vertices = 0
        self.last_status, self.last_chi2 = 0, None

    def grow(self):
        """Add round ``t``'s increment.  New own vertices are dead-reckoned from the newest vertex' current estimate,
        new foreign vertices from the current estimate of the own vertex that saw them (closure measurement)."""
        g, w, t = self.g, self.w, self.t
        v0, v1, e_idx, c_idx = w.increment(t)
        if v0 == 0:
            prev = w.truth0.copy()
            new = [prev]
            first = 1
        else:
            prev = g.poses(self._own_slot[v0 - 1], 1)[0]
            new, first = [], v0
        if v1 > first:
            chain = _dead_reckon(prev, w.meas[first - 1:v1 - 1])               # odometry edge k-1: (k-1) -> k
            new.extend(chain)
            prev = chain[-1]
        fixed = np.zeros(v1 - v0, dtype=np.uint8)
        if v0 == 0:
            fixed[0] = 1                                                        # the robot's first vertex is its gauge
            self._own_slot = {}
        g.add_vertices(w.ids[v0:v1], np.array(new).reshape(-1, 3), fixed)
        for k in range(v0, v1):
            self._own_slot[k] = self.n_vertices + (k - v0)
        self.n_vertices += v1 - v0
        if len(e_idx):
            g.add_edges(w.ids[w.ef[e_idx]], w.ids[w.et[e_idx]], w.meas[e_idx], w.info[e_idx])
        # inter-robot closures that became possible
        if len(c_idx):
            new_f_ids, new_f_pose, req = [], [], {}
            for c in c_idx:
                fid = int(w.c_fid[c])
                if fid not in self.foreign:
                    self.foreign[fid] = True
                    own_pose = g.poses(self._own_slot[int(w.c_own[c])], 1)[0]
                    new_f_ids.append(fid)
                    new_f_pose.append(synth.se2_compose(own_pose[None], w.c_meas[c][None])[0])
                    req.setdefault(fid // w.base_id, []).append(fid)
            if new_f_ids:
                g.add_vertices(np.array(new_f_ids), np.array(new_f_pose).reshape(-1, 3), None)
                self.n_vertices += len(new_f_ids)
            g.add_edges(w.ids[w.c_own[c_idx]], w.c_fid[c_idx], w.c_meas[c_idx], w.c_info[c_idx])
            for p, ids in req.items():
                g.insertInClosure(p, np.array(ids))                            # ask p for condensed edges among them
        self.t += 1

    def optimize(self):
        self.last_status, self.last_chi2 = self.g.optimize(self.iterations)
        return self.last_status

    def condense(self):
        return self.g.computeCondensedGraph(-1)

    def round(self, exchange):
        """One full round against an ``Exchange``-like object (``start()`` / ``finish()``)."""
        self.grow()
        self.optimize()
        n_in = exchange.finish()            # previous round's all-gather: it ran while this round grew and solved
        built = self.condense()
        exchange.start()
        return n_in, built


class LoopbackExchange:
    """All robots in one process (tests, single-GPU dry runs).  ``device=False``: the 'all-gather' is a concatenation of host
    buffers (any object with ``pack_host`` / ``ingest_host``).  ``device=True`` (robots that share a GPU, a context each): every
    robot's packed message is copied into the others' receive buffers on the device (``RobotGraph.deliver``), nothing waits on
    the host -- what the all-gather on the communicator's stream does between ranks."""

    def __init__(self, graphs, device: bool = False):
        self.graphs = graphs
        self.wire = None
        self.device = device
        self.pending = False

    def start_all(self):
        if self.device:
            for g in self.graphs:
                g.pack(0)
            for src in self.graphs:
                for dst in self.graphs:
                    if dst is not src:
                        src.deliver(dst)
            self.pending = True
            return
        self.wire = np.concatenate([g.pack_host() for g in self.graphs])

    def finish_all(self):
        if self.device:
            if not self.pending:
                return None
            self.pending = False
            return [g.ingest_delivered() for g in self.graphs]
        if self.wire is None:
            return None
        out = [g.ingest_host(self.wire) for g in self.graphs]
        self.wire = None
        return out


class TakeTurns:
    """Robots that share a GPU run their rounds ONE AFTER THE OTHER -- robot r: grow, solve, ingest what the others delivered
    in their previous round, condensed graphs, pack, deliver -- instead of in lock step: what the timeline of one rank of an
    N-GPU run looks like (its batch of condensed graphs has the rest of the round to finish beside the next grow / analysis /
    solve), on one device.  Same data flow as the lock-step order: the round-t message of robot r is ingested by robot q in
    its round t + 1 (``RobotGraph.deliver`` keeps two receive buffers in turn).  Device transport only."""

    def __init__(self, rounds):
        self.rounds = rounds
        self.t = 0

    def robot_round(self, rr):
        """One robot's whole round; returns (edges accepted per sender or None, condensed graphs built)."""
        rr.grow()
        rr.optimize()
        n_in = rr.g.ingest_delivered() if self.t > 0 else None
        built = rr.condense()
        rr.g.pack(0)
        for other in self.rounds:
            if other is not rr:
                rr.g.deliver(other.g)
        return n_in, built

    def round(self):
        out = [self.robot_round(rr) for rr in self.rounds]
        self.t += 1
        return out


def run_rounds_loopback(rounds, n_rounds, device: bool = False):
    """Drive several ``RobotRounds`` (one per robot, same process) through ``n_rounds`` rounds with a loopback exchange,
    in the order a real run has: everybody grows and solves, then ingests the previous round, condenses, exchanges."""
    ex = LoopbackExchange([r.g for r in rounds], device=device)
    log = []
    for _ in range(n_rounds):
        for r in rounds:
            r.grow()
            r.optimize()
        n_in = ex.finish_all()
        built = [r.condense() for r in rounds]
        ex.start_all()
        log.append((n_in, built, [float(r.last_chi2[-1]) for r in rounds]))
    ex.finish_all()
    return log
