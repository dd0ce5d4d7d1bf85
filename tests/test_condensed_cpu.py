"""CPU tests of the multi-robot bookkeeping behind the C ABI (cgmr_graph_* on a graph without a device: closures, wire
format, replace-on-receive) against the plain-numpy restatement (tests/ref_condensed.py), and of the inter-robot
exchange over gloo with world_size 2."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from cg_mrslam_amd import synth
from cg_mrslam_amd.condensed import WIRE_EDGE_DTYPE, RobotGraph, unpack_wire
from ref_condensed import EDGE_DTYPE, CondensedGraphBuffer, RefRobotGraph, select_gauge_centroid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fill(graph, g):
    """Whole robot graph of synth.make_multi_robot into a RobotGraph-like object."""
    graph.add_vertices(g["ids"], g["poses_all"], g["fixed_all"])
    graph.add_edges(g["ids"][g["ef_all"]], g["ids"][g["et_all"]], g["meas_all"], g["info_all"])
    for q, ids in g["in_closures"].items():
        graph.insertInClosure(q, ids)


def _fake(n, seed):
    rng = np.random.default_rng(seed)
    return rng.normal(size=(n, 3)).astype(np.float32), np.tile(np.array([100, 0, 0, 100, 0, 1000], dtype=np.float32), (n, 1))


def test_wire_edge_is_44_bytes():
    assert WIRE_EDGE_DTYPE.itemsize == 44 == EDGE_DTYPE.itemsize


def test_wire_format_matches_numpy_restatement_and_round_trips():
    R = synth.make_multi_robot(3, 300, 800, seed=5)
    c = [RobotGraph(None, r, 3, cap_edges=64) for r in range(3)]
    ref = [RefRobotGraph(None, r, 3, cap_edges=64) for r in range(3)]
    for r in range(3):
        _fill(c[r], R[r]); _fill(ref[r], R[r])
    own = R[0]["ids"][:300]
    e1, i1 = _fake(10, 1)
    e2, i2 = _fake(3, 2)
    c[0].set_condensed(1, own[5], own[6:16], e1, i1)
    c[0].set_condensed(2, own[7], own[20:23], e2, i2)
    for peer, frm, to, e, i in ((1, own[5], own[6:16], e1, i1), (2, own[7], own[20:23], e2, i2)):
        w = np.zeros(len(to), dtype=EDGE_DTYPE)
        w["from"], w["to"], w["est"], w["info"] = frm, to, e, i
        ref[0].buf.out_condensed[peer] = w
    buf = c[0].pack_host()
    assert buf.dtype == np.uint8 and len(buf) == c[0].wire_bytes() == ref[0].buf.wire_bytes()
    assert np.array_equal(buf, ref[0].pack_host())                       # byte-identical to the restatement
    robot, n_e, n_c, edges, clos = unpack_wire(buf, 3, 64)
    assert robot == 0 and list(n_e) == [0, 10, 3]
    assert np.array_equal(edges[1, :10]["to"], own[6:16]) and np.array_equal(edges[1, :10]["est"], e1)
    assert np.array_equal(clos[1, :n_c[1]], np.sort(R[0]["in_closures"][1]))


def test_ingest_replaces_previous_set_skips_unknown_and_keeps_on_empty():
    R = synth.make_multi_robot(2, 300, 800, seed=6)
    a, b = RobotGraph(None, 0, 2), RobotGraph(None, 1, 2)
    ra, rb = RefRobotGraph(None, 0, 2), RefRobotGraph(None, 1, 2)
    for g, src in ((a, R[0]), (b, R[1]), (ra, R[0]), (rb, R[1])):
        _fill(g, src)
    n0 = a.counts()["own_edges"]
    foreign = np.sort(R[0]["in_closures"][1])           # robot 1's vertices robot 0 knows
    assert len(foreign) >= 3

    def send(frm, to, seed):
        e, i = _fake(len(to), seed)
        b.set_condensed(0, frm, to, e, i)
        w = np.zeros(len(to), dtype=EDGE_DTYPE)
        w["from"], w["to"], w["est"], w["info"] = frm, to, e, i
        rb.buf.out_condensed[0] = w
        wire = np.concatenate([a.pack_host(), b.pack_host()])
        assert np.array_equal(wire, np.concatenate([ra.pack_host(), rb.pack_host()]))
        return a.ingest_host(wire), ra.ingest_host(wire), e

    # 3 edges, one with an end point robot 0 has never seen: skipped (mr_graph_slam.cpp:360-363)
    n, n_ref, e = send(foreign[0], np.array([foreign[1], foreign[2], 19999]), 3)
    assert list(n) == [0, 2] == list(n_ref)
    assert a.counts() == {"vertices": a.counts()["vertices"], "own_edges": n0, "received_edges": 2, "peers_with_requests": 1}
    f, t, m, i = a.received_edges(1)
    fr, tr, mr, ir = ra.received_edges(1)
    assert np.array_equal(f, fr) and np.array_equal(t, tr) and np.array_equal(m, mr) and np.array_equal(i, ir)
    assert m.dtype == np.float64 and np.array_equal(m, e[:2].astype(np.float64))      # float32 widened to double
    # the newest set replaces the previous one (condensed_graph_buffer.cpp:487-510)
    n, n_ref, e = send(foreign[1], foreign[2:3], 5)
    assert list(n) == [0, 1] == list(n_ref) and a.counts()["received_edges"] == 1
    # a message whose edges are all unknown to me (or that has none) leaves the previous set in place (:393-394)
    n, n_ref, _ = send(foreign[1], np.array([19998]), 6)
    assert list(n) == [0, 0] == list(n_ref) and a.counts()["received_edges"] == 1
    n, n_ref, _ = send(foreign[1], np.zeros(0, dtype=np.int64), 7)
    assert list(n) == [0, 0] == list(n_ref) and a.counts()["received_edges"] == 1
    assert np.array_equal(a.received_edges(1)[2], ra.received_edges(1)[2])
    # the closure requests of robot 1 became out-closures of robot 0: exactly the ids robot 1 asks for
    assert np.array_equal(a.closures(1, "out"), np.sort(R[1]["in_closures"][0]))
    assert np.array_equal(a.closures(1, "out"), np.sort(ra.buf.out_closures[1]))


def test_one_peer_messages_equal_the_all_gather_path():
    """``cgmr_graph_message_for`` / ``cgmr_graph_message_from`` (constructCondensedGraphMessage / addInterRobotData for
    ONE CondensedGraphMessage, mr_graph_slam.cpp:607-670, 331-395) move exactly what the round buffer moves, as the
    reference's own byte string."""
    from cg_mrslam_amd.messages import CondensedGraphMessage, from_bytes
    R = synth.make_multi_robot(2, 300, 800, seed=6)
    a, b = RobotGraph(None, 0, 2), RobotGraph(None, 1, 2)
    ra, rb = RefRobotGraph(None, 0, 2), RefRobotGraph(None, 1, 2)
    for g, src in ((a, R[0]), (b, R[1]), (ra, R[0]), (rb, R[1])):
        _fill(g, src)
    foreign = np.sort(R[0]["in_closures"][1])
    e, i = _fake(2, 11)
    b.set_condensed(0, foreign[0], foreign[1:3], e, i)
    w = np.zeros(2, dtype=EDGE_DTYPE)
    w["from"], w["to"], w["est"], w["info"] = foreign[0], foreign[1:3], e, i
    rb.buf.out_condensed[0] = w
    msg, msg_ref = b.message_for(0), rb.message_for(0)
    assert isinstance(msg, CondensedGraphMessage) and msg.robotId == 1
    assert msg.to_bytes() == msg_ref.to_bytes()                  # edges for robot 0 + the ids robot 1 asks robot 0 for
    assert np.array_equal(msg.closures, np.sort(R[1]["in_closures"][0]))
    got = a.message_from(from_bytes(msg.to_bytes()))
    want = ra.buf.ingest_from(1, msg_ref.edges, msg_ref.closures)      # (the restatement's compute step needs a solver)
    assert got == want == 2
    f, t, m, ii = a.received_edges(1)
    fr, tr, mr, ir = ra.received_edges(1)
    assert np.array_equal(f, fr) and np.array_equal(t, tr) and np.array_equal(m, mr) and np.array_equal(ii, ir)
    assert np.array_equal(a.closures(1, "out"), np.sort(ra.buf.out_closures[1]))
    # nothing for a robot that neither asked nor is asked
    c = RobotGraph(None, 0, 3)
    c.add_vertices([0, 1], np.zeros((2, 3)), [1, 0])
    assert c.message_for(2) is None
    # a message from oneself or from an unknown robot is refused; one over the capacity is dropped (a reference node's
    # receive buffer would not have held it either), not an error
    for bad in (CondensedGraphMessage(0), CondensedGraphMessage(5)):
        with pytest.raises(Exception):
            a.message_from(bad)
    assert a.message_from(CondensedGraphMessage(1, np.zeros(200, dtype=EDGE_DTYPE))) == 0 and a.skipped_messages() == 1


def test_capacity_overflow_skips_the_message_like_the_reference():
    """A message beyond the wire capacity is left out whole -- ``toCharArray`` returns 0 beyond MAX_LENGTH_MSG and
    ``GraphComm::send`` skips the send (graph_comm.cpp:112-122) -- never truncated, never an error that would leave the
    other ranks waiting in the all-gather; a condensed graph that cannot fit is refused where it is set."""
    g = RobotGraph(None, 0, 3, cap_edges=4)
    g.add_vertices(np.arange(10), np.zeros((10, 3)))
    from cg_mrslam_amd._lib import CgmrError
    with pytest.raises(CgmrError):
        g.set_condensed(1, 0, np.arange(1, 7), *_fake(6, 1))             # 6 edges > cap 4: no such message can exist
    g.insertInClosure(1, 10000 + np.arange(5))                           # 5 requests > cap 4: the message to robot 1 is skipped
    g.insertInClosure(2, 20000 + np.arange(3))                           # the message to robot 2 fits
    r = RefRobotGraph(None, 0, 3, cap_edges=4)
    r.insertInClosure(1, 10000 + np.arange(5))
    r.insertInClosure(2, 20000 + np.arange(3))
    buf, ref = g.pack_host(), r.pack_host()
    assert np.array_equal(buf, ref)
    robot, n_e, n_c, _edges, clos = unpack_wire(buf, 3, 4)
    assert list(n_c) == [0, 0, 3] and list(clos[2, :3]) == [20000, 20001, 20002]
    assert g.skipped_messages() == 1 and r.buf.skipped == 1
    # the receiving side drops a message beyond its capacity instead of failing
    from cg_mrslam_amd.messages import CondensedGraphMessage
    big = CondensedGraphMessage(1, np.zeros(6, dtype=EDGE_DTYPE), np.zeros(0, dtype=np.int32))
    assert g.message_from(big) == 0 and g.skipped_messages() == 2


def test_select_gauge_centroid():
    xy = np.array([[0.0, 0], [10, 0], [4, 1], [5, 5]])
    assert select_gauge_centroid(xy) == 2


def test_numeric_entry_points_need_a_device():
    from cg_mrslam_amd._lib import CgmrError
    g = RobotGraph(None, 0, 2)
    g.add_vertices([0, 1], np.zeros((2, 3)), [1, 0])
    g.add_edges([0], [1], [[1.0, 0, 0]], [[100, 0, 0, 100, 0, 1000]])
    with pytest.raises(CgmrError):
        g.optimize(1)
    with pytest.raises(CgmrError):
        g.computeCondensedGraph(-1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import torch.distributed as dist
    from cg_mrslam_amd.condensed import Exchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R = synth.make_multi_robot(world, 300, 800, seed=7)
    g = RobotGraph(None, rank, world, cap_edges=64)
    _fill(g, R[rank])
    ex = Exchange(g)
    assert ex.transport == "host"
    n0 = g.counts()["own_edges"]
    # round 1: only requests travel (nobody knows yet what the peers want)
    ex.start(); ex.finish()
    peer = 1 - rank
    want = g.closures(peer, "out")                      # what the peer asked me for: ids of MY vertices
    assert np.array_equal(want, np.sort(R[peer]["in_closures"][rank]))
    # a (fake, CPU) condensed star over the requested vertices travels in round 2
    e, i = _fake(len(want) - 1, 10 + rank)
    g.set_condensed(peer, want[0], want[1:], e, i)
    ex.start()
    n = ex.finish()
    c = g.counts()
    out.put((rank, int(n0), c["own_edges"], c["received_edges"], int(n[peer]), len(R[rank]["in_closures"][peer])))
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
    for rank, n0, own, recv, got, n_in in res:
        # I receive a star over the vertices *I* asked for: n_in - 1 edges, all end points known to me
        assert got == recv == n_in - 1 and own == n0


def test_bench_gpus_2_spawns_two_ranks_dry_run():
    """``bench.py --gpus 2`` without RANK in the environment must start two ranks itself.  Without a GPU the ranks run the
    protocol dry (CGMR_BENCH_DRY=1: rendezvous over gloo, the C5 round protocol on graphs without a device with fake
    condensed edges) -- the launcher, the rank plumbing and the exchange leg are what is tested here."""
    env = dict(os.environ, CGMR_BENCH_DRY="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    import json
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["dry_run"] is True
    ex = out["exchange"]
    assert ex is not None and ex["robots"] == 2 and ex["rounds"] >= 2 and ex["condensed_edges_received_total"] > 0
    # a rank count that disagrees with the environment is refused, not ignored
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env2, capture_output=True, text=True, timeout=120)
    assert p2.returncode != 0 and "WORLD_SIZE" in (p2.stderr + p2.stdout)
