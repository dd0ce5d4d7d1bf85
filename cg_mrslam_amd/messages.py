"""The reference's inter-robot messages, byte for byte (src/mrslam/msg_factory.{h,cpp}).

``GraphComm`` ships these over UDP (src/mrslam/graph_comm.cpp:103-124, port 42001, at most ``MAX_LENGTH_MSG`` bytes);
a robot running this package can therefore talk to a robot running the reference.  Layout (x86-64, little endian;
``_toCharArray`` copies the object representation, msg_factory.h:43-58, except that every ``double`` is narrowed to
``float`` on the wire, :78-112; ``size_t`` counters are 8 bytes):

    header                int32 type, int32 robotId                                   msg_factory.cpp:31-38
    VertexArrayMessage    u64 n, n x {int32 id, float estimate[3]}                    :53-76      (type 1)
    RobotLaserMessage     int32 nodeId, u64 n, n x float, float minangle,
                          angleincrement, maxrange, accuracy                          :95-124     (type 2)
    ComboMessage          header, vertex array body, robot laser body                 :146-152    (type 4)
    EdgeArrayMessage      u64 n, n x {int32 idfrom, idto, float estimate[3],
                          float information[6]}  = 44 bytes per edge                  :163-199    (type 5)
    ClosuresMessage       u64 n, n x int32                                            :225-241    (type 6)
    CondensedGraphMessage header, edge array body, closures body                      :256-262    (type 7)

Only the two messages ``GraphComm::sendToThrd`` actually sends are implemented (graph_comm.cpp:126-155): ComboMessage
and CondensedGraphMessage.  Host bookkeeping only -- no numerics.
"""
from __future__ import annotations

import struct

import numpy as np

MAX_LENGTH_MSG = 100000          # msg_factory.h:115
TYPE_VERTEX_ARRAY, TYPE_ROBOT_LASER, TYPE_COMBO, TYPE_EDGE_ARRAY, TYPE_CLOSURES, TYPE_CONDENSED_GRAPH = 1, 2, 4, 5, 6, 7

VERTEX_DTYPE = np.dtype([("id", "<i4"), ("estimate", "<f4", (3,))])                       # VSE2Data on the wire: 16 B
EDGE_DTYPE = np.dtype([("from", "<i4"), ("to", "<i4"), ("est", "<f4", (3,)), ("info", "<f4", (6,))])   # ESE2Data: 44 B
assert VERTEX_DTYPE.itemsize == 16 and EDGE_DTYPE.itemsize == 44


def _counted(arr: np.ndarray) -> bytes:
    return struct.pack("<Q", len(arr)) + arr.tobytes()


def _read_counted(buf: bytes, o: int, dtype) -> tuple[np.ndarray, int]:
    (n,) = struct.unpack_from("<Q", buf, o)
    o += 8
    size = n * np.dtype(dtype).itemsize
    if o + size > len(buf):
        raise ValueError("message truncated")
    return np.frombuffer(buf, dtype=dtype, count=n, offset=o).copy(), o + size


class ComboMessage:
    """The sender's newest vertex with its scan, and the estimates of its last vertices (mr_graph_slam.cpp:564-605)."""

    type = TYPE_COMBO

    def __init__(self, robotId=-1, vertex_ids=(), estimates=(), nodeId=0, readings=(), minangle=0.0,   # noqa: N803
                 angleincrement=0.0, maxrange=0.0, accuracy=0.0):
        self.robotId = int(robotId)
        self.vertices = np.zeros(len(vertex_ids), dtype=VERTEX_DTYPE)
        self.vertices["id"] = np.asarray(vertex_ids, dtype=np.int32)
        if len(vertex_ids):
            self.vertices["estimate"] = np.asarray(estimates, dtype=np.float64).reshape(-1, 3)      # double -> float
        self.nodeId = int(nodeId)
        self.readings = np.asarray(readings, dtype=np.float32)
        self.minangle, self.angleincrement = np.float32(minangle), np.float32(angleincrement)
        self.maxrange, self.accuracy = np.float32(maxrange), np.float32(accuracy)

    def to_bytes(self) -> bytes | None:
        b = struct.pack("<ii", self.type, self.robotId) + _counted(self.vertices)
        b += struct.pack("<i", self.nodeId) + _counted(self.readings)
        b += struct.pack("<ffff", self.minangle, self.angleincrement, self.maxrange, self.accuracy)
        return b if len(b) <= MAX_LENGTH_MSG else None          # toCharArray returns 0: nothing is sent

    @classmethod
    def from_bytes(cls, buf: bytes) -> "ComboMessage":
        t, rid = struct.unpack_from("<ii", buf, 0)
        if t != cls.type:
            raise ValueError(f"type mismatch: {t}")
        m = cls(rid)
        m.vertices, o = _read_counted(buf, 8, VERTEX_DTYPE)
        (m.nodeId,) = struct.unpack_from("<i", buf, o)
        m.readings, o = _read_counted(buf, o + 4, "<f4")
        m.minangle, m.angleincrement, m.maxrange, m.accuracy = (np.float32(v) for v in struct.unpack_from("<ffff", buf, o))
        if o + 16 != len(buf):
            raise ValueError("trailing bytes")                   # MessageFactory::fromCharArray asserts the size
        return m


class CondensedGraphMessage:
    """Condensed edges built for the receiver plus the receiver's vertices the sender wants condensed in return
    (mr_graph_slam.cpp:607-670)."""

    type = TYPE_CONDENSED_GRAPH

    def __init__(self, robotId=-1, edges=None, closures=()):   # noqa: N803
        self.robotId = int(robotId)
        self.edges = np.zeros(0, dtype=EDGE_DTYPE) if edges is None else np.ascontiguousarray(edges, dtype=EDGE_DTYPE)
        self.closures = np.asarray(closures, dtype=np.int32)

    @classmethod
    def from_arrays(cls, robotId, from_ids, to_ids, est, info, closures):   # noqa: N803
        e = np.zeros(len(to_ids), dtype=EDGE_DTYPE)
        if len(to_ids):
            e["from"], e["to"] = np.asarray(from_ids, dtype=np.int32), np.asarray(to_ids, dtype=np.int32)
            e["est"] = np.asarray(est, dtype=np.float64).reshape(-1, 3)                             # double -> float
            e["info"] = np.asarray(info, dtype=np.float64).reshape(-1, 6)
        return cls(robotId, e, closures)

    def to_bytes(self) -> bytes | None:
        b = struct.pack("<ii", self.type, self.robotId) + _counted(self.edges) + _counted(self.closures)
        return b if len(b) <= MAX_LENGTH_MSG else None

    @classmethod
    def from_bytes(cls, buf: bytes) -> "CondensedGraphMessage":
        t, rid = struct.unpack_from("<ii", buf, 0)
        if t != cls.type:
            raise ValueError(f"type mismatch: {t}")
        edges, o = _read_counted(buf, 8, EDGE_DTYPE)
        clos, o = _read_counted(buf, o, "<i4")
        if o != len(buf):
            raise ValueError("trailing bytes")
        return cls(rid, edges, clos)


def from_bytes(buf: bytes):
    """``MessageFactory::fromCharArray`` (msg_factory.cpp:301-313) for the registered types that are sent."""
    (t,) = struct.unpack_from("<i", buf, 0)
    if t == TYPE_COMBO:
        return ComboMessage.from_bytes(buf)
    if t == TYPE_CONDENSED_GRAPH:
        return CondensedGraphMessage.from_bytes(buf)
    raise ValueError(f"message type {t} is not one GraphComm sends")


__all__ = ["ComboMessage", "CondensedGraphMessage", "from_bytes", "MAX_LENGTH_MSG", "EDGE_DTYPE", "VERTEX_DTYPE"]
