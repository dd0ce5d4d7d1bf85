"""CPU tests of the condensed-graph bookkeeping and the inter-robot exchange (gloo, world_size 2)."""
import os
import socket

import numpy as np
import pytest

from cg_mrslam_amd import synth
from cg_mrslam_amd.condensed import EDGE_DTYPE, CondensedGraphBuffer, select_gauge_centroid
from cg_mrslam_amd.graph import PoseGraph


def _robot_graph(g):
    return PoseGraph(g["ids"], g["poses_all"], g["fixed_all"], g["ef_all"], g["et_all"], g["meas_all"], g["info_all"])


def _fake_edges(ids_from, ids_to, seed):
    rng = np.random.default_rng(seed)
    e = np.zeros(len(ids_to), dtype=EDGE_DTYPE)
    e["from"] = ids_from
    e["to"] = ids_to
    e["est"] = rng.normal(size=(len(ids_to), 3)).astype(np.float32)
    e["info"] = np.tile(np.array([100, 0, 0, 100, 0, 1000], dtype=np.float32), (len(ids_to), 1))
    return e


def test_wire_format_is_44_bytes_and_round_trips():
    R = synth.make_multi_robot(3, 300, 800, seed=5)
    b = [CondensedGraphBuffer(_robot_graph(R[r]), r, 3, cap_edges=64) for r in range(3)]
    for q, ids in R[0]["in_closures"].items():
        b[0].insertInClosure(q, ids)
    b[0].out_condensed[1] = _fake_edges(5, np.arange(6, 16), 1)
    b[0].out_condensed[2] = _fake_edges(7, np.arange(20, 23), 2)
    buf = b[0].pack()
    assert buf.dtype == np.uint8 and len(buf) == b[0].wire_bytes()
    sender, edges, clos = b[1].unpack(buf)
    assert sender == 0 and len(edges) == 10
    assert np.array_equal(edges, b[0].out_condensed[1])
    assert np.array_equal(clos, R[0]["in_closures"][1][:64])
    sender, edges, _ = b[2].unpack(buf)
    assert len(edges) == 3 and np.array_equal(edges["to"], [20, 21, 22])


def test_insert_edges_replaces_previous_set_and_skips_unknown_vertices():
    R = synth.make_multi_robot(2, 300, 800, seed=6)
    g = _robot_graph(R[0])
    buf = CondensedGraphBuffer(g, 0, 2)
    n0 = g.n_edges
    foreign = R[0]["in_closures"][1]
    assert len(foreign) >= 3
    e1 = _fake_edges(foreign[0], foreign[1:3], 3)
    e1 = np.concatenate([e1, _fake_edges(foreign[0], np.array([19999]), 4)])     # unknown end point: skipped
    assert buf.insertEdgesFromRobot(1, e1) == 2
    assert g.n_edges == n0 + 2 and (buf.in_edge_src >= 0).sum() == 2
    assert buf.my_edge_mask().sum() == n0                                         # getMyEdges excludes received edges
    e2 = _fake_edges(foreign[1], foreign[2:3], 5)
    assert buf.insertEdgesFromRobot(1, e2) == 1                                   # replaces, does not accumulate
    assert g.n_edges == n0 + 1
    assert g.meas.dtype == np.float64 and np.allclose(g.meas[-1], e2["est"][0])


def test_select_gauge_centroid():
    xy = np.array([[0.0, 0], [10, 0], [4, 1], [5, 5]])
    assert select_gauge_centroid(xy) == 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R = synth.make_multi_robot(world, 300, 800, seed=7)
    g = _robot_graph(R[rank])
    buf = CondensedGraphBuffer(g, rank, world, cap_edges=64)
    for q, ids in R[rank]["in_closures"].items():
        buf.insertInClosure(q, ids)
    n0 = g.n_edges
    # round 1: only requests travel (nobody knows yet what the peers want)
    buf.exchange()
    peer = 1 - rank
    want = buf.out_closures[peer]                       # what the peer asked me for: ids of MY vertices
    assert np.array_equal(want, R[peer]["in_closures"][rank])
    # build a (fake, CPU) condensed star over the requested vertices and send it in round 2
    buf.out_condensed[peer] = _fake_edges(want[0], want[1:], 10 + rank)
    buf.exchange()
    got = (buf.in_edge_src == peer).sum()
    out.put((rank, int(n0), int(g.n_edges), int(got), len(R[rank]["in_closures"][peer])))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_two_rank_exchange():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, n0, n1, got, n_in in res:
        # I receive a star over the vertices *I* asked for: n_in - 1 edges, all end points known to me
        assert got == n_in - 1 and n1 == n0 + got
