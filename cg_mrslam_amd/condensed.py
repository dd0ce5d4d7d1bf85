"""Multi-robot path: one robot's pose graph resident in HBM, its condensed graphs and the inter-robot exchange.

Thin host-side mirror of the reference's
  * ``CondensedGraphBuffer``  (src/mrslam/condensed_graph/condensed_graph_buffer.{h,cpp}): in-/out-closures,
    ``computeCondensedGraph``, ``insertEdgesFromRobot`` (replace-on-receive), ``getMyEdges``, ``selectGaugeCentroid``;
  * ``MRGraphSLAM::addInterRobotData``  (src/mrslam/mr_graph_slam.cpp:331-395);
  * the wire structs of ``msg_factory.h`` (44 bytes per edge, float32 on the wire, src/mrslam/msg_factory.h:78-112,200-218);
  * ``GraphComm``'s pairwise UDP send / receive (src/mrslam/graph_comm.cpp:103-208), replaced by ONE all-gather per round
    of a fixed-capacity buffer per rank over RCCL / xGMI
over the C ABI (include/cgmr.h "robot graph" and "exchange"; csrc/mrslam_api.cpp, mrslam_kernels.hip, rccl_comm.cpp).
Everything here is a ctypes call; there is no numeric or bookkeeping logic in Python and no CPU fallback for the numeric
entry points (a graph created without a context only keeps the books, for CPU tests of the protocol).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import CGMR_E_CHOLESKY_BASE, CgmrError, Context, load_library

WIRE_EDGE_DTYPE = np.dtype([("from", "<i4"), ("to", "<i4"), ("est", "<f4", (3,)), ("info", "<f4", (6,))])
assert WIRE_EDGE_DTYPE.itemsize == 44            # EdgeArrayMessage::ESE2Data on the wire


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None and a.size else C.c_void_p(0)


class RobotGraph:
    """``cgmr_graph``: the g2o optimiser of one robot plus its ``CondensedGraphBuffer``."""

    #: edges (and closure ids) per peer a reference message holds: MAX_LENGTH_MSG = 100000 bytes of 44-byte edges (msg_factory.h:115)
    REFERENCE_CAP_EDGES = 2270

    def __init__(self, ctx: Context | None, robot: int, n_robots: int, base_id: int = 10000, cap_edges: int = 128,
                 async_condense: bool = False):
        self.lib = ctx.lib if ctx is not None else load_library()
        self.lib.cgmr_graph_last_error.restype = C.c_char_p
        self.lib.cgmr_graph_wire_bytes.restype = C.c_int64
        self.lib.cgmr_graph_skipped_messages.restype = C.c_int64
        self.lib.cgmr_graph_failed_batches.restype = C.c_int64
        self.lib.cgmr_graph_send_buffer.restype = C.c_void_p
        self.lib.cgmr_graph_recv_buffer.restype = C.c_void_p
        self.lib.cgmr_graph_destroy.restype = None
        self.ctx, self.robot, self.n_robots, self.base_id, self.cap = ctx, robot, n_robots, base_id, cap_edges
        h = C.c_void_p()
        rc = self.lib.cgmr_graph_create(ctx.h if ctx is not None else C.c_void_p(0), C.c_int(robot), C.c_int(n_robots),
                                        C.c_int(base_id), C.c_int(cap_edges), C.byref(h))
        if rc != 0:
            raise CgmrError(rc, "cgmr_graph_create failed")
        self.h = h
        # computeCondensedGraph queues its passes on the context's side stream and returns (cgmr_graph_compute_condensed_async):
        # they run beside the next round's grow / analysis / solve; pack, the all-gather and deliver() follow them on the device
        self.async_condense = bool(async_condense) and ctx is not None
        if self.async_condense:
            self._check(self.lib.cgmr_graph_set_async(self.h, C.c_int(1)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.cgmr_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, allow_cholesky=False):
        if rc >= 0 or (allow_cholesky and rc <= CGMR_E_CHOLESKY_BASE):
            return rc
        raise CgmrError(rc, self.lib.cgmr_graph_last_error(self.h).decode() or
                        (self.ctx.lib.cgmr_last_error(self.ctx.h).decode() if self.ctx is not None else ""))

    # ------------------------------------------------------------------ graph
    def add_vertices(self, ids, poses, fixed=None):
        ids, poses = _i32(ids), _f64(poses).reshape(-1, 3)
        fx = None if fixed is None else np.ascontiguousarray(fixed, dtype=np.uint8)
        self._check(self.lib.cgmr_graph_add_vertices(self.h, C.c_int(len(ids)), _p(ids), _p(poses), _p(fx) if fx is not None else C.c_void_p(0)))

    def add_edges(self, from_ids, to_ids, meas, info):
        f, t = _i32(from_ids), _i32(to_ids)
        self._check(self.lib.cgmr_graph_add_edges(self.h, C.c_int(len(f)), _p(f), _p(t), _p(_f64(meas)), _p(_f64(info))))

    def counts(self):
        out = np.zeros(4, dtype=np.int32)
        self._check(self.lib.cgmr_graph_counts(self.h, _p(out)))
        return dict(zip(["vertices", "own_edges", "received_edges", "peers_with_requests"], out.tolist()))

    def optimize(self, iters: int):
        """``GraphSLAM::optimize(iters)``: returns (status, chi2[iters+1]); a Cholesky failure is a status, not an exception."""
        chi = np.zeros(iters + 1)
        rc = self.lib.cgmr_graph_optimize(self.h, C.c_int(iters), _p(chi))
        self._check(rc, allow_cholesky=True)
        return rc, chi

    def poses(self, first=0, n=None):
        n = self.counts()["vertices"] - first if n is None else n
        out = np.zeros((n, 3))
        self._check(self.lib.cgmr_graph_get_poses(self.h, C.c_int(first), C.c_int(n), _p(out)))
        return out

    def set_poses(self, first, poses):
        poses = _f64(poses).reshape(-1, 3)
        self._check(self.lib.cgmr_graph_set_poses(self.h, C.c_int(first), C.c_int(len(poses)), _p(poses)))

    # ------------------------------------------------------------------ closures / condensed graphs
    def insertInClosure(self, peer, ids):   # noqa: N802
        ids = _i32(ids)
        self._check(self.lib.cgmr_graph_insert_in_closure(self.h, C.c_int(peer), C.c_int(len(ids)), _p(ids)))

    def insertOutClosure(self, peer, ids):   # noqa: N802
        ids = _i32(ids)
        self._check(self.lib.cgmr_graph_insert_out_closure(self.h, C.c_int(peer), C.c_int(len(ids)), _p(ids)))

    def closures(self, peer, which="out"):
        out = np.zeros(4096, dtype=np.int32)
        n = self._check(self.lib.cgmr_graph_closures(self.h, C.c_int(peer), C.c_int(0 if which == "out" else 1), C.c_int(len(out)), _p(out)))
        return out[:n].copy()

    def set_optimal_gauge(self, optimal: bool):
        """``computeCondensedGraph(robot, optimal)``: selectOptimalGauge instead of selectGaugeCentroid (reference default: off)."""
        self._check(self.lib.cgmr_graph_set_optimal_gauge(self.h, C.c_int(1 if optimal else 0)))

    def computeCondensedGraph(self, peer: int = -1):   # noqa: N802
        """``computeCondensedGraph`` for one peer or (``peer < 0``) for every peer that has asked; returns the number built."""
        if self.async_condense:
            return self._check(self.lib.cgmr_graph_compute_condensed_async(self.h, C.c_int(peer)))
        return self._check(self.lib.cgmr_graph_compute_condensed(self.h, C.c_int(peer)))

    def condensed_wait(self):
        """Wait for the passes ``computeCondensedGraph`` queued (``async_condense``); raises if one of them failed."""
        self._check(self.lib.cgmr_graph_condensed_wait(self.h))

    def deliver(self, dst: "RobotGraph"):
        """My packed message (``pack()`` first) into ``dst``'s receive buffer, on the device (robots sharing a GPU)."""
        self._check(self.lib.cgmr_graph_deliver(self.h, dst.h))

    def condensed(self, peer):
        """(gauge id, to ids[n], est[n,3], info_upper[n,6]) of the condensed graph built for ``peer``, double precision."""
        cap = self.cap
        gid = C.c_int32(-1)
        to, est, iu = np.zeros(cap, dtype=np.int32), np.zeros((cap, 3)), np.zeros((cap, 6))
        n = self._check(self.lib.cgmr_graph_get_condensed(self.h, C.c_int(peer), C.c_int(cap), C.byref(gid), _p(to), _p(est), _p(iu)))
        return (int(gid.value) if n else None), to[:n].astype(np.int64), est[:n].copy(), iu[:n].copy()

    def set_condensed(self, peer, from_id, to_ids, est, info):
        to = _i32(to_ids)
        e = np.ascontiguousarray(est, dtype=np.float32).reshape(-1, 3)
        i = np.ascontiguousarray(info, dtype=np.float32).reshape(-1, 6)
        self._check(self.lib.cgmr_graph_set_condensed(self.h, C.c_int(peer), C.c_int(len(to)), C.c_int32(int(from_id)), _p(to), _p(e), _p(i)))

    # ------------------------------------------------------------------ wire
    def skipped_messages(self) -> int:
        """Messages left out, not built or dropped because they exceed ``cap_edges`` (the reference skips a send whose
        ``toCharArray`` does not fit ``MAX_LENGTH_MSG``, graph_comm.cpp:112-122)."""
        return int(self.lib.cgmr_graph_skipped_messages(self.h))

    def failed_batches(self) -> int:
        """Asynchronous batches of condensed graphs that failed (Cholesky / a bounded device-side wait): their peers got no edges
        in that round's message; the later rounds are built regardless."""
        return int(self.lib.cgmr_graph_failed_batches(self.h))

    def wire_bytes(self) -> int:
        return int(self.lib.cgmr_graph_wire_bytes(self.h))

    def send_buffer(self) -> int:
        return int(self.lib.cgmr_graph_send_buffer(self.h) or 0)

    def recv_buffer(self) -> int:
        return int(self.lib.cgmr_graph_recv_buffer(self.h) or 0)

    def pack(self, d_send: int = 0):
        self._check(self.lib.cgmr_graph_pack(self.h, C.c_void_p(d_send)))

    def ingest(self, d_recv: int = 0):
        n = np.zeros(self.n_robots, dtype=np.int32)
        self._check(self.lib.cgmr_graph_ingest(self.h, C.c_void_p(d_recv), _p(n)))
        return n

    def ingest_delivered(self):
        """The ingest that goes with ``deliver``: the k-th call digests the k-th message of every peer."""
        n = np.zeros(self.n_robots, dtype=np.int32)
        self._check(self.lib.cgmr_graph_ingest_delivered(self.h, _p(n)))
        return n

    def pack_host(self) -> np.ndarray:
        buf = np.zeros(self.wire_bytes(), dtype=np.uint8)
        self._check(self.lib.cgmr_graph_pack_host(self.h, _p(buf)))
        return buf

    def ingest_host(self, recv) -> np.ndarray:
        recv = np.ascontiguousarray(recv, dtype=np.uint8).reshape(-1)
        if recv.size != self.n_robots * self.wire_bytes():
            raise ValueError("receive buffer must hold n_robots wire buffers")
        n = np.zeros(self.n_robots, dtype=np.int32)
        self._check(self.lib.cgmr_graph_ingest_host(self.h, _p(recv), _p(n)))
        return n

    def message_for(self, peer):
        """``constructCondensedGraphMessage(peer)``: a ``messages.CondensedGraphMessage`` or None (nothing to send)."""
        from .messages import EDGE_DTYPE, CondensedGraphMessage
        cap = self.cap
        edges, clos = np.empty(cap, dtype=EDGE_DTYPE), np.empty(cap, dtype=np.int32)
        ne, nc = C.c_int32(0), C.c_int32(0)
        rc = self._check(self.lib.cgmr_graph_message_for(self.h, C.c_int(peer), C.c_int(cap), _p(edges), C.byref(ne), C.c_int(cap),
                                                         _p(clos), C.byref(nc)))
        return CondensedGraphMessage(self.robot, edges[:ne.value].copy(), clos[:nc.value].copy()) if rc == 1 else None

    def message_from(self, msg) -> int:
        """``addInterRobotData(CondensedGraphMessage*)``: returns the number of edges now held from the sender because of
        this message (0 = the previous set stays)."""
        e = np.ascontiguousarray(msg.edges)
        c = _i32(msg.closures)
        n = C.c_int32(0)
        self._check(self.lib.cgmr_graph_message_from(self.h, C.c_int(msg.robotId), C.c_int(len(e)), _p(e), C.c_int(len(c)), _p(c), C.byref(n)))
        return int(n.value)

    def received_edges(self, peer):
        n = self.counts_received(peer)
        if n == 0:
            return (np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64), np.zeros((0, 3)), np.zeros((0, 6)))
        f, t, m, i = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32), np.empty((n, 3)), np.empty((n, 6))
        n = self._check(self.lib.cgmr_graph_received_edges(self.h, C.c_int(peer), C.c_int(n), _p(f), _p(t), _p(m), _p(i)))
        return f[:n].astype(np.int64), t[:n].astype(np.int64), m[:n], i[:n]

    def counts_received(self, peer):
        """Edges currently held from ``peer`` (no data moved)."""
        return int(self._check(self.lib.cgmr_graph_received_edges(self.h, C.c_int(peer), C.c_int(0), None, None, None, None)))

    def last_seconds(self):
        out = np.zeros(2)
        self._check(self.lib.cgmr_graph_last_seconds(self.h, _p(out)))
        return {"optimize": float(out[0]), "condense": float(out[1])}


def unpack_wire(buf: np.ndarray, n_robots: int, cap: int):
    """Decode one rank's wire buffer (tests, debugging): (robot, n_edges[R], n_closures[R], edges[R][cap], closures[R][cap])."""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    hdr = buf[:4 * (2 + 2 * n_robots)].view(np.int32)
    o = 4 * (2 + 2 * n_robots)
    edges = buf[o:o + n_robots * cap * 44].view(WIRE_EDGE_DTYPE).reshape(n_robots, cap)
    o += n_robots * cap * 44
    clos = buf[o:o + n_robots * cap * 4].view(np.int32).reshape(n_robots, cap)
    return int(hdr[0]), hdr[2:2 + n_robots].copy(), hdr[2 + n_robots:2 + 2 * n_robots].copy(), edges, clos


class Exchange:
    """One all-gather per round of every rank's wire buffer.

    transport "rccl":  ``cgmr_allgather_condensed`` on device buffers, a native RCCL communicator on a side stream (the
                       unique id travels through ``torch.distributed``); overlaps with whatever the context's stream does next
    transport "torch": ``torch.distributed.all_gather_into_tensor`` on the same device buffers (backend nccl = RCCL)
    transport "host":  host staging (backend gloo: CPU tests, single-GPU dry runs)
    ``start()`` issues the collective, ``finish()`` ingests what arrived; in between the caller runs the next solve."""

    def __init__(self, graph: RobotGraph, transport: str = "auto", group=None):
        import torch
        import torch.distributed as dist
        self.g, self.group, self.torch, self.dist = graph, group, torch, dist
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if self.world != graph.n_robots:
            raise ValueError("one rank per robot: world size must equal n_robots")
        self.wb = graph.wire_bytes()
        backend = dist.get_backend(group) if dist.is_initialized() else "none"
        if transport == "auto":
            transport = "rccl" if (backend == "nccl" and graph.ctx is not None) else ("host" if backend in ("gloo", "none") else "torch")
        self.comm = None
        self.pending = None
        self.fallback_reason = None
        if transport == "rccl":
            try:
                self._init_rccl()
            except Exception as e:          # noqa: BLE001 -- any failure: fall back to torch's own RCCL communicator
                self.fallback_reason = f"{type(e).__name__}: {e}"
                transport = "torch"
            # all ranks must agree on the transport
            flag = torch.tensor([1 if transport == "rccl" else 0], device=torch.device("cuda", graph.ctx.device))
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if int(flag.item()) == 0 and transport == "rccl":
                self._destroy_comm()
                transport = "torch"
                self.fallback_reason = "a peer could not create the RCCL communicator"
        self.transport = transport
        if transport == "torch":
            dev = torch.device("cuda", graph.ctx.device)
            graph.lib.cgmr_ctx_stream.restype = C.c_void_p
            # the context's stream as a torch stream: the two are ordered with events, the host never waits
            self.ctx_stream = torch.cuda.ExternalStream(int(graph.lib.cgmr_ctx_stream(graph.ctx.h) or 0), device=dev)
            self.t_send = torch.zeros(self.wb, dtype=torch.uint8, device=dev)
            self.t_recv = torch.zeros(self.world * self.wb, dtype=torch.uint8, device=dev)

    def _init_rccl(self):
        """Native RCCL communicator next to torch's.  Two guards against a hang (a rank that fails BEFORE the collective
        ``ncclCommInitRank`` would leave the others inside it): every rank first probes librccl locally and the ranks agree
        (MIN all-reduce) that all of them can go on; the collective initialisation itself runs under a timeout
        (``CGMR_RCCL_INIT_TIMEOUT`` seconds, default 60) and a rank that runs out of time falls back to the torch
        transport -- its peers, stuck in the same collective, time out the same way and the agreement in ``__init__``
        settles on "torch" for everybody."""
        import os
        import threading
        torch, dist, g = self.torch, self.dist, self.g
        lib = g.lib
        dev = torch.device("cuda", g.ctx.device)
        h = np.zeros(128, dtype=np.uint8)
        # rank 0 asks librccl for the id; the others only check that the library resolves (ncclGetUniqueId starts a bootstrap
        # listener: one per rank that nothing ever uses otherwise)
        ok_local = lib.cgmr_comm_unique_id(_p(h)) if dist.get_rank(self.group) == 0 else lib.cgmr_comm_probe()
        flag = torch.tensor([1 if ok_local == 0 else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if int(flag.item()) == 0:
            raise CgmrError(ok_local or -1, "librccl is not loadable on every rank")
        uid = torch.from_numpy(h).to(dev)
        dist.broadcast(uid, src=0, group=self.group)
        h = np.ascontiguousarray(uid.cpu().numpy())
        comm = C.c_void_p()
        res = {}

        def create():
            res["rc"] = lib.cgmr_comm_create(g.ctx.h, C.c_int(self.world), C.c_int(dist.get_rank(self.group)), _p(h), C.byref(comm))

        t = threading.Thread(target=create, daemon=True)
        t.start()
        t.join(float(os.environ.get("CGMR_RCCL_INIT_TIMEOUT", "60")))
        if t.is_alive():
            # the thread may still come back with a communicator: close() then takes care of it; until it has, the context is
            # shared with a thread inside ncclCommInitRank, which touches the device but none of the context's buffers
            self._late_init = (t, comm, res)
            raise TimeoutError("ncclCommInitRank did not return (native RCCL communicator); using torch.distributed instead")
        if res.get("rc", -1) != 0:
            raise CgmrError(res.get("rc", -1), g.ctx.lib.cgmr_last_error(g.ctx.h).decode())
        self.comm = comm

    def diagnostics(self) -> dict:
        """What this rank's exchange runs on, for a multi-GPU run's own report: the transport, why it is not the native one (if so),
        and how many ranks the native communicator itself reports (``ncclCommCount``)."""
        d = {"transport": self.transport, "transport_fallback_reason": self.fallback_reason, "world": self.world,
             "native_comm_ranks": None, "native_comm_rank": None}
        if self.transport == "rccl" and self.comm is not None:
            out = np.zeros(3, dtype=np.int32)
            if self.g.lib.cgmr_comm_info(self.comm, _p(out)) == 0:
                d["native_comm_ranks"], d["native_comm_rank"] = int(out[0]), int(out[1])
                d["native_comm_answered_by_librccl"] = bool(out[2])
        return d

    def _destroy_comm(self):
        if self.comm is not None:
            self.g.lib.cgmr_comm_destroy.restype = None
            self.g.lib.cgmr_comm_destroy(self.comm)
            self.comm = None

    def close(self):
        late = getattr(self, "_late_init", None)
        if late is not None:                                  # a collective initialisation that outlived its time-out
            t, comm, res = late
            t.join(5.0)
            if not t.is_alive() and res.get("rc", -1) == 0 and comm:
                self.g.lib.cgmr_comm_destroy.restype = None
                self.g.lib.cgmr_comm_destroy(comm)
            self._late_init = None
        self._destroy_comm()

    def start(self):
        """Serialise my message and issue the all-gather; returns immediately for the device transports."""
        g = self.g
        if self.world == 1:
            self.pending = None
            return
        if self.transport == "rccl":
            g.pack(0)
            g.ctx._check(g.lib.cgmr_allgather_condensed(g.ctx.h, self.comm, C.c_void_p(g.send_buffer()), C.c_size_t(self.wb),
                                                        C.c_void_p(g.recv_buffer())))
            self.pending = "rccl"
        elif self.transport == "torch":
            g.pack(self.t_send.data_ptr())
            g.ctx._check(g.lib.cgmr_ctx_join_side(g.ctx.h))     # (a batch queued on the context's side stream: the packed message sits behind it)
            # the context's stream is not torch's: torch's stream waits (on the device) for the packed message, the
            # collective then overlaps with whatever the context's stream does next -- as on the native transport
            self.torch.cuda.current_stream(self.t_send.device).wait_stream(self.ctx_stream)
            self.pending = self.dist.all_gather_into_tensor(self.t_recv, self.t_send, group=self.group, async_op=True)
        else:
            send = self.torch.from_numpy(g.pack_host())
            recv = self.torch.empty(self.world * self.wb, dtype=self.torch.uint8)
            self.pending = (self.dist.all_gather_into_tensor(recv, send, group=self.group, async_op=True), recv, send)

    def finish(self):
        """Wait for the all-gather issued by ``start()`` and ingest it.  Returns accepted edges per sender (or None)."""
        g = self.g
        if self.pending is None:
            return None
        if self.transport == "rccl":
            g.ctx._check(g.lib.cgmr_comm_wait(g.ctx.h, self.comm))
            n = g.ingest(0)
        elif self.transport == "torch":
            self.pending.wait()                                                   # torch's stream waits for the collective ...
            self.ctx_stream.wait_stream(self.torch.cuda.current_stream(self.t_recv.device))   # ... and the context's stream for torch's
            n = g.ingest(self.t_recv.data_ptr())
        else:
            work, recv, _send = self.pending
            work.wait()
            n = g.ingest_host(recv.numpy())
        self.pending = None
        return n

    def last_collective_seconds(self):
        if self.transport != "rccl" or self.comm is None:
            return None
        s = C.c_double()
        if self.g.lib.cgmr_comm_last_seconds(self.comm, C.byref(s)) != 0:
            return None
        return s.value
