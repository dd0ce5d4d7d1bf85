"""Multi-robot key-frame driver (cg_mrslam, SIM modality; SURVEY.md 8f row 1, BASELINE config C4) on the CPU:
the message formats byte for byte, the closure buffers, and the whole loop -- two robots in one process and one robot
per gloo rank -- with the CPU oracle as the numeric backend (test infrastructure; the product path runs the same driver on
``condensed.RobotGraph`` + the GPU matchers, tests/test_mr_graph_slam_gpu.py)."""
import os
import socket
import struct

import numpy as np
import pytest

from cg_mrslam_amd import synth
from cg_mrslam_amd.messages import ComboMessage, CondensedGraphMessage, EDGE_DTYPE, from_bytes
from cg_mrslam_amd.mr_graph_slam import (GraphCommRanks, GraphCommSim, MRClosureBuffer, MRGraphSLAMDriver, pack_outbox,
                                         run_cg_mrslam, run_cg_mrslam_rank, unpack_outbox)
from cg_mrslam_amd.slam import ClosureBuffer

import oracle_backend as OB
from ref_condensed import RefRobotGraph


# ------------------------------------------------------------------------------------------------ messages
def test_combo_message_bytes_follow_msg_factory():
    """Laid out by hand from msg_factory.cpp: header (type 4, robot), size_t count + {int id, 3 floats} per vertex,
    int nodeId, size_t count + floats, four float laser parameters -- doubles narrowed to float (msg_factory.h:96-112)."""
    m = ComboMessage(2, [20007, 20008], [[1.5, -2.25, 0.5], [1.75, -2.0, 0.625]], nodeId=20008, readings=[1.0, 2.5, 30.0],
                     minangle=-2.35619449, angleincrement=0.00436332313, maxrange=30.0, accuracy=0.1)
    want = struct.pack("<ii", 4, 2)
    want += struct.pack("<Q", 2) + struct.pack("<ifff", 20007, 1.5, -2.25, 0.5) + struct.pack("<ifff", 20008, 1.75, -2.0, 0.625)
    want += struct.pack("<i", 20008) + struct.pack("<Q", 3) + struct.pack("<fff", 1.0, 2.5, 30.0)
    want += struct.pack("<ffff", -2.35619449, 0.00436332313, 30.0, 0.1)
    b = m.to_bytes()
    assert b == want and len(b) == 8 + 8 + 2 * 16 + 4 + 8 + 3 * 4 + 16
    r = from_bytes(b)
    assert isinstance(r, ComboMessage) and r.robotId == 2 and r.nodeId == 20008
    np.testing.assert_array_equal(r.vertices["id"], [20007, 20008])
    np.testing.assert_array_equal(r.vertices["estimate"], np.array([[1.5, -2.25, 0.5], [1.75, -2.0, 0.625]], dtype=np.float32))
    np.testing.assert_array_equal(r.readings, np.array([1.0, 2.5, 30.0], dtype=np.float32))
    assert r.minangle == np.float32(-2.35619449) and r.accuracy == np.float32(0.1)
    with pytest.raises(ValueError):
        from_bytes(b + b"\0")                    # MessageFactory::fromCharArray asserts that the whole buffer is consumed


def test_condensed_graph_message_bytes_and_narrowing():
    info = [1000.0000001, 0.5, -0.25, 999.0, 0.125, 1e4]
    m = CondensedGraphMessage.from_arrays(1, [10004], [10009], [[0.1, 0.2, 0.3]], [info], [7, 9, 12])
    want = struct.pack("<ii", 7, 1) + struct.pack("<Q", 1) + struct.pack("<ii", 10004, 10009)
    want += struct.pack("<fff", 0.1, 0.2, 0.3) + struct.pack("<ffffff", *info)
    want += struct.pack("<Q", 3) + struct.pack("<iii", 7, 9, 12)
    b = m.to_bytes()
    assert b == want and len(b) == 8 + 8 + 44 + 8 + 12
    r = from_bytes(b)
    assert r.robotId == 1 and r.closures.tolist() == [7, 9, 12] and r.edges.dtype == EDGE_DTYPE
    assert r.edges["info"][0, 0] == np.float32(1000.0000001) == np.float32(1000.0)      # float32 on the wire
    # an empty message is still a message: header + two zero counters
    assert len(CondensedGraphMessage(3).to_bytes()) == 8 + 8 + 8
    # 100000 bytes is the limit of one UDP datagram of the reference (msg_factory.h:115): nothing is sent beyond it
    big = CondensedGraphMessage(0, np.zeros(2300, dtype=EDGE_DTYPE))
    assert big.to_bytes() is None


def test_outbox_round_trip():
    out = [(1, b"abc"), (3, b""), (0, bytes(range(200)))]
    assert unpack_outbox(pack_outbox(out, 1024)) == out
    with pytest.raises(ValueError):
        pack_outbox([(0, bytes(100))], 64)


# ------------------------------------------------------------------------------------------------ closure buffers
def test_mr_closure_buffer_windows():
    """mr_closure_buffer.cpp + closure_buffer.cpp:88-107: ages count key frames; a vertex leaves (with its edges) at
    ``window``, a robot's buffer disappears when it is empty, checkList fires at age window-1."""
    mb = MRClosureBuffer()
    e1 = {"from": 5, "to": 10003, "meas": np.zeros(3), "added": False}
    c = ClosureBuffer()
    c.addVertex(10003)
    c.addEdge(e1)
    mb.insert(c, 1)
    assert mb.size() == 1 and mb.findClosuresRobot(1).findVertex(10003) and mb.findClosuresRobot(0) is None
    mb.update(3)
    c2 = ClosureBuffer()
    c2.addVertex(10004)
    c2.addEdge({"from": 6, "to": 10004, "meas": np.zeros(3), "added": False})
    mb.insert(c2, 1)                                   # joins the existing buffer of robot 1
    cb = mb.findClosuresRobot(1)
    assert sorted(cb.vertex_ids()) == [10003, 10004] and len(cb.edges) == 2
    mb.update(3)
    assert cb.checkList(3) and not cb.checkList(4)     # 10003 has age 2 = window - 1, nothing has age 3
    mb.update(3)                                       # 10003 reaches age 3 and leaves with its edge
    assert cb.vertex_ids() == [10004] and len(cb.edges) == 1 and cb.edges[0]["to"] == 10004
    mb.remove(c2, 1)
    assert mb.size() == 0


# ------------------------------------------------------------------------------------------------ the whole loop
def _team(n_robots, n_steps=110, laps=0.26):
    team = synth.make_robot_team(n_robots, n_steps=n_steps, laps=laps, gap=3.0)
    la = (team[0]["n_beams"], team[0]["angle_min"], team[0]["angle_inc"], team[0]["max_range"])
    return team, la


def _oracle_slam(r, n_robots, la):
    octx = OB.OracleContext()
    s = MRGraphSLAMDriver(octx, OB.close_matcher(la), OB.lc_matcher(la), RefRobotGraph(octx, r, n_robots), r, n_robots,
                          windowLoopClosure=5, minInliers=4)
    s.setInterRobotClosureParams(0.15, 3, 5)
    return s


def _summary(s):
    return dict(ids=s.g.ids.copy(), ef=s.g.edge_from.copy(), et=s.g.edge_to.copy(), kind=list(s.edge_kind), meas=s.g.meas.copy(),
                poses=s.g.poses.copy(), log=list(s.log))


@pytest.fixture(scope="module")
def two_robot_run(oracle):
    team, la = _team(2)
    slams = [_oracle_slam(r, 2, la) for r in range(2)]
    comm = GraphCommSim(slams)
    loops = run_cg_mrslam(slams, team, comm=comm, linearUpdate=0.5)
    return team, slams, comm, loops


def test_two_robots_close_inter_robot_loops_and_exchange_condensed_graphs(two_robot_run):
    team, slams, comm, loops = two_robot_run
    assert all(lp.key_frames >= 15 for lp in loops) and comm.delivered > 50
    for s in slams:
        me, peer = s.idRobot, 1 - s.idRobot
        kinds = {k: s.edge_kind.count(k) for k in set(s.edge_kind)}
        assert kinds.get("mr", 0) >= 3 and kinds.get("cond", 0) >= 2, kinds
        # every accepted inter-robot edge goes from one of my vertices to a copy of a peer vertex, information
        # diag(100, 100, 1000) (mr_graph_slam.cpp:234-236), and its far end is in my in-closures for that peer
        want = s.rg.closures(peer, "in")
        for k, kd in enumerate(s.edge_kind):
            if kd == "mr":
                a, b = int(s.g.ids[s.g.edge_from[k]]), int(s.g.ids[s.g.edge_to[k]])
                assert a // 10000 == me and b // 10000 == peer and b in want
                np.testing.assert_array_equal(s.g.info[k], [100, 0, 0, 100, 0, 1000])
            if kd == "cond":                                     # a star among the peer vertices I hold
                a, b = int(s.g.ids[s.g.edge_from[k]]), int(s.g.ids[s.g.edge_to[k]])
                assert a // 10000 == peer and b // 10000 == peer and a in want and b in want
        # what I ask for is what the peer condenses for me
        np.testing.assert_array_equal(np.sort(want), np.sort(slams[peer].rg.closures(me, "out")))
        # the received star is the newest one the peer built (float32 on the wire)
        gid, to, est, iu = slams[peer].rg.condensed(me)
        f, t, m, i = s.rg.received_edges(peer)
        # (the peer may have rebuilt its star after its last message was delivered; the previous one then still stands)
        assert len(f) >= 1 and set(np.concatenate([f, t]).tolist()) <= set(want.tolist())
        # own trajectory stays on the true path
        tp = team[me]["truth"]
        own = [q for q in range(s.g.n_vertices) if s.isMyVertex(q)]
        err = max(np.min(np.hypot(tp[:, 0] - p[0], tp[:, 1] - p[1])) for p in s.g.poses[own])
        assert err < 0.3, err
    # the two maps agree: a vertex of robot 1 as robot 0 estimates it lies where robot 1 itself puts it
    a, b = slams
    shared = [int(v) for v in a.g.ids if v // 10000 == 1]
    assert len(shared) >= 3
    d = [np.hypot(*(a.g.poses[a._index_of_id(v)][:2] - b.g.poses[b._index_of_id(v)][:2])) for v in shared]
    assert max(d) < 0.3, max(d)


def test_detect_robot_in_range_gates_the_closures(oracle):
    """``setDetectRobotInRange(true)`` (mr_graph_slam.h:57): a global match only becomes a closure candidate if
    verifyMatching finds the other robot's body -- points of my local map that the peer's map does not explain, at the
    place the match puts the peer (mean kernel value of the +-0.3 m window <= 40, scan_matcher.cpp:430-505).  The robots of
    this team see each other as 0.5 m boxes; robot 1 drives 3 m ahead of robot 0, inside robot 0's 270 degree field of
    view, while robot 0 is in robot 1's blind rear sector: only robot 0 accepts closures."""
    team = synth.make_robot_team(2, n_steps=70, laps=0.17, gap=3.0, body=0.5)
    la = (team[0]["n_beams"], team[0]["angle_min"], team[0]["angle_inc"], team[0]["max_range"])
    slams = [_oracle_slam(r, 2, la) for r in range(2)]
    for s in slams:
        s.setDetectRobotInRange(True)
    run_cg_mrslam(slams, team, linearUpdate=0.5)
    for s in slams:
        ver = [l for l in s.log if l[0] == "verify"]
        assert len(ver) > 0 and all((l[4] <= 40.0) == l[3] for l in ver)
        passed = {l[2] for l in ver if l[3]}
        for k, kd in enumerate(s.edge_kind):
            if kd == "mr":
                assert int(s.g.ids[s.g.edge_to[k]]) in passed
    assert slams[0].edge_kind.count("mr") >= 3 and slams[0].edge_kind.count("cond") >= 2
    assert slams[1].edge_kind.count("mr") == 0 and not any(l[3] for l in slams[1].log if l[0] == "verify")
    assert len(slams[1].rg.closures(0, "out")) >= 3          # robot 1 still condenses its graph for robot 0


def test_three_robots_only_neighbours_talk(oracle):
    """Robots 3 m apart in a row: 0-1 and 1-2 are within SIM_COMM_RANGE (5 m), 0-2 (6 m) are not
    (graph_comm.cpp:62-64, 76-84): no message, no closure, no condensed graph between them."""
    team, la = _team(3, n_steps=70, laps=0.17)
    slams = [_oracle_slam(r, 3, la) for r in range(3)]
    run_cg_mrslam(slams, team, linearUpdate=0.5)
    assert len(slams[0].rg.closures(2, "in")) == 0 and len(slams[2].rg.closures(0, "in")) == 0
    assert not any(l[0] == "combo" and l[1] == 2 for l in slams[0].log)
    assert any(l[0] == "combo" and l[1] == 1 for l in slams[0].log) and any(l[0] == "combo" and l[1] == 2 for l in slams[1].log)


def test_robots_drifting_out_of_range_stop_talking_and_windows_expire(oracle):
    """Robot 1 starts 1 m ahead of robot 0 and drives 2.4 times as fast: while they are within SIM_COMM_RANGE every new
    vertex travels as a ComboMessage, afterwards nothing does (graph_comm.cpp:126-155); vertices that never matched age
    out of the unmatched-vertex window (mr_closure_buffer.cpp:93-118) and their scans are dropped."""
    n = 120
    team = [synth.make_trajectory(n, seed=31, laps=0.15, start=0.0), synth.make_trajectory(n, seed=132, laps=0.36, start=1.0)]
    la = (team[0]["n_beams"], team[0]["angle_min"], team[0]["angle_inc"], team[0]["max_range"])
    slams = [_oracle_slam(r, 2, la) for r in range(2)]
    comm = GraphCommSim(slams)
    gaps = [float(np.hypot(*(team[0]["truth"][k][:2] - team[1]["truth"][k][:2]))) for k in range(n)]
    last_in_range = max(k for k in range(n) if gaps[k] < 5.0)
    assert 10 < last_in_range < n - 30                      # the scenario really separates the robots well before the end
    delivered = []
    from cg_mrslam_amd.mr_graph_slam import RobotLoop
    loops = [RobotLoop(s, tr["odom"], tr["scans"], tr["truth"][0], 0.5) for s, tr in zip(slams, team)]
    for k in range(1, n):
        for lp in loops:
            lp.tick(k)
        comm.cycle([tr["truth"][k] for tr in team])
        delivered.append(comm.delivered)
    assert delivered[last_in_range - 1] > 5                  # they talked ...
    assert delivered[-1] == delivered[last_in_range]         # ... and not a single message after the last tick in range
    for s in slams:
        # ten key frames after the separation every unmatched vertex has left its window, and with it its scan
        assert s.interRobotVertices.size() == 0
        assert all(s._index_of_id(v) is not None or s._buffered(v) for v in s.peer)
        # what was accepted before stays in the graph and keeps being optimised
        assert s.last_status == 0 and np.isfinite(s.g.poses).all()


# ------------------------------------------------------------------------------------------------ one rank per robot
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    O.lib()
    team, la = _team(world)
    s = _oracle_slam(rank, world, la)
    run_cg_mrslam_rank(s, team, linearUpdate=0.5)
    out.put((rank, _summary(s)))
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_per_robot_reproduces_the_single_process_run(two_robot_run):
    """The all-gather transport (``GraphCommRanks``, gloo here, RCCL on GPUs) delivers the same messages in the same
    order as the in-process simulation: identical graphs, bit for bit."""
    import torch.multiprocessing as mp
    _, slams, _, _ = two_robot_run
    mpc = mp.get_context("spawn")
    out = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_rank_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(out.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        a, b = _summary(slams[r]), res[r]
        assert a["kind"] == b["kind"] and a["log"] == b["log"]
        for k in ("ids", "ef", "et", "meas", "poses"):
            np.testing.assert_array_equal(a[k], b[k])
    assert GraphCommRanks is not None
