"""GPU parity tests of the multi-robot path (rows a7, a8, e of SURVEY.md section 8): the device-resident robot graph,
its condensed graphs and the exchange, through the C ABI, against the CPU oracle driven through the same rounds by the
plain-numpy restatement of the bookkeeping (tests/ref_condensed.py).
Tolerances: condensed measurement <= 1e-6 (m, rad), information <= 1e-4 relative (it is the inverse of a 3x3 covariance
whose entries agree to ~1e-9; the inversion amplifies by its condition number), poses <= 1e-6; the wire is float32."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from cg_mrslam_amd import synth
from cg_mrslam_amd.condensed import RobotGraph, unpack_wire
from cg_mrslam_amd.mrslam import RobotRounds, RobotWorld, run_rounds_loopback

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rounds(graph_factory, robots, chunk, n_robots):
    return [RobotRounds(graph_factory(r), RobotWorld(robots, r, chunk=chunk)) for r in range(n_robots)]


def _meeting_world(n_robots, n_v, n_e, min_shared=8):
    for seed in range(46, 120):                            # first seed whose walks actually meet
        R = synth.make_multi_robot(n_robots, n_v, n_e, seed=seed)
        if all(len(R[r]["in_closures"].get(q, [])) >= min_shared for r in range(n_robots) for q in range(n_robots) if q != r):
            return R
    raise AssertionError("no seed makes the robots meet")


def test_two_robot_rounds_match_oracle_backend_edge_by_edge(ctx, oracle):
    """Two robots, 8 rounds of 150 vertices each on one GPU (loopback exchange), the same rounds on the oracle backend:
    every condensed edge of every round's final graphs, every received edge and the final poses agree; the received
    condensed edges pull the foreign vertices towards their truth."""
    import oracle_backend as ob
    from ref_condensed import RefRobotGraph
    R = _meeting_world(2, 1200, 4000)
    n_rounds, chunk = 8, 150
    gpu = _rounds(lambda r: RobotGraph(ctx, r, 2), R, chunk, 2)
    ref = _rounds(lambda r: RefRobotGraph(ob.OracleContext(), r, 2), R, chunk, 2)
    log_g = run_rounds_loopback(gpu, n_rounds)
    log_o = run_rounds_loopback(ref, n_rounds)
    exchanged = 0
    for t in range(n_rounds):
        (n_in_g, built_g, chi_g), (n_in_o, built_o, chi_o) = log_g[t], log_o[t]
        assert built_g == built_o
        if n_in_g is not None:
            assert [list(a) for a in n_in_g] == [list(a) for a in n_in_o]
            exchanged += int(np.sum(n_in_g))
        assert np.allclose(chi_g, chi_o, rtol=1e-6)
    assert exchanged > 20                                   # condensed edges really travelled, in several rounds
    for r in range(2):
        a, b = gpu[r].g, ref[r].g
        assert a.counts() == b.counts()
        gid_a, to_a, est_a, iu_a = a.condensed(1 - r)
        gid_b, to_b, est_b, iu_b = b.condensed(1 - r)
        assert gid_a == gid_b and len(to_a) >= 7 and np.array_equal(to_a, to_b)
        assert np.abs(est_a - est_b).max() < 1e-6
        assert np.abs(iu_a - iu_b).max() <= 1e-4 * np.abs(iu_b).max()
        fa, ta, ma, ia = a.received_edges(1 - r)
        fb, tb, mb, ib = b.received_edges(1 - r)
        assert len(fa) >= 7 and np.array_equal(fa, fb) and np.array_equal(ta, tb)
        assert np.abs(ma - mb).max() < 1e-5 and np.abs(ia - ib).max() <= 2e-4 * np.abs(ib).max()    # float32 wire
        pa, pb = a.poses(), b.poses()
        assert np.abs(pa[:, :2] - pb[:, :2]).max() < 1e-6 and np.abs(synth.normalize_theta(pa[:, 2] - pb[:, 2])).max() < 1e-6
        # the wire message is what the restatement would send (ids exactly, float32 payload to rounding)
        robot, n_e, n_c, edges, clos = unpack_wire(a.pack_host(), 2, a.cap)
        robot_b, n_e_b, n_c_b, edges_b, clos_b = unpack_wire(b.pack_host(), 2, b.cap)
        assert robot == robot_b == r and np.array_equal(n_e, n_e_b) and np.array_equal(n_c, n_c_b) and np.array_equal(clos, clos_b)
        k = n_e[1 - r]
        assert np.array_equal(edges[1 - r, :k]["to"], edges_b[1 - r, :k]["to"]) and np.array_equal(edges[1 - r, :k]["from"], edges_b[1 - r, :k]["from"])
        assert np.abs(edges[1 - r, :k]["est"] - edges_b[1 - r, :k]["est"]).max() < 1e-5


def test_three_robot_rounds_match_oracle_backend(ctx, oracle):
    """R > 2: every robot addresses two peers, the wire buffers carry three slices each.  Three robots x 6 rounds on one
    GPU against the same rounds on the oracle backend: what every robot built for every peer, what it holds from every
    peer and its final poses."""
    import oracle_backend as ob
    from ref_condensed import RefRobotGraph
    R = _meeting_world(3, 900, 3000, min_shared=4)
    n_rounds, chunk, nr = 6, 150, 3
    gpu = _rounds(lambda r: RobotGraph(ctx, r, nr), R, chunk, nr)
    ref = _rounds(lambda r: RefRobotGraph(ob.OracleContext(), r, nr), R, chunk, nr)
    log_g = run_rounds_loopback(gpu, n_rounds)
    log_o = run_rounds_loopback(ref, n_rounds)
    for t in range(n_rounds):
        (n_in_g, built_g, chi_g), (n_in_o, built_o, chi_o) = log_g[t], log_o[t]
        assert built_g == built_o
        if n_in_g is not None:
            assert [list(a) for a in n_in_g] == [list(a) for a in n_in_o]
        assert np.allclose(chi_g, chi_o, rtol=1e-6)
    pairs = 0
    for r in range(nr):
        a, b = gpu[r].g, ref[r].g
        assert a.counts() == b.counts()
        for q in range(nr):
            if q == r:
                continue
            gid_a, to_a, est_a, iu_a = a.condensed(q)
            gid_b, to_b, est_b, iu_b = b.condensed(q)
            assert gid_a == gid_b and np.array_equal(to_a, to_b)
            if len(to_a):
                pairs += 1
                assert np.abs(est_a - est_b).max() < 1e-6 and np.abs(iu_a - iu_b).max() <= 1e-4 * np.abs(iu_b).max()
            fa, ta, ma, ia = a.received_edges(q)
            fb, tb, mb, ib = b.received_edges(q)
            assert np.array_equal(fa, fb) and np.array_equal(ta, tb)
            if len(fa):
                assert np.abs(ma - mb).max() < 1e-5 and np.abs(ia - ib).max() <= 2e-4 * np.abs(ib).max()
        pa, pb = a.poses(), b.poses()
        assert np.abs(pa[:, :2] - pb[:, :2]).max() < 1e-6 and np.abs(synth.normalize_theta(pa[:, 2] - pb[:, 2])).max() < 1e-6
        robot, n_e, n_c, edges, clos = unpack_wire(a.pack_host(), nr, a.cap)
        robot_b, n_e_b, n_c_b, edges_b, clos_b = unpack_wire(b.pack_host(), nr, b.cap)
        assert robot == robot_b == r and np.array_equal(n_e, n_e_b) and np.array_equal(n_c, n_c_b) and np.array_equal(clos, clos_b)
    assert pairs >= 4                                       # most ordered pairs exchanged condensed graphs


_BATCH_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, 'tests')
from cg_mrslam_amd import synth, Context
from cg_mrslam_amd.condensed import RobotGraph
from cg_mrslam_amd.mrslam import RobotRounds, RobotWorld, run_rounds_loopback
import test_multirobot_gpu as T
nr = 4
R = T._meeting_world(nr, 900, 3000, min_shared=4)
ctx = Context(0)
rounds = T._rounds(lambda r: RobotGraph(ctx, r, nr), R, 150, nr)
log = run_rounds_loopback(rounds, 6)
out = {}
for r in range(nr):
    out['wire%d' % r] = np.frombuffer(rounds[r].g.pack_host(), dtype=np.uint8).copy()
    out['poses%d' % r] = rounds[r].g.poses()
    for q in range(nr):
        if q != r:
            gid, to, est, iu = rounds[r].g.condensed(q)
            out['to%d_%d' % (r, q)] = np.asarray(to); out['est%d_%d' % (r, q)] = np.asarray(est); out['iu%d_%d' % (r, q)] = np.asarray(iu)
out['built'] = np.array([b for (_, b, _) in log], dtype=np.int64)
np.savez(sys.argv[1], **out)
print('rounds ok')
"""


def test_condensed_graphs_as_one_batch_equal_the_passes_on_streams(tmp_path):
    """The condensed graphs of a round run as ONE batch of launches with a job dimension (mrslam_api.cpp run_cond_jobs); with
    CGMR_COND_BATCH=0 every pass gets its stream of launches as in round 2.  Four robots x 6 rounds each way (the switch is
    read once per process: two children): same graphs built, same requested vertices, condensed edges and poses equal to
    rounding (the batch splits the backward solve's chained launch at another level: other summation order in the border
    reductions), the float32 wire records equal to a float32 ulp."""
    res = {}
    for mode in ("1", "0"):
        path = str(tmp_path / ("batch%s.npz" % mode))
        env = dict(os.environ, CGMR_COND_BATCH=mode, PYTHONPATH=ROOT)
        r = subprocess.run([sys.executable, "-c", _BATCH_CHILD, path], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "rounds ok" in r.stdout, r.stderr[-2000:]
        res[mode] = dict(np.load(path))
    a, b = res["1"], res["0"]
    assert sorted(a) == sorted(b) and np.array_equal(a["built"], b["built"]) and a["built"].sum() > 10
    n_edges = 0
    for k in a:
        if k.startswith("to"):
            assert np.array_equal(a[k], b[k])
            n_edges += len(a[k])
        elif k.startswith("wire"):
            assert a[k].shape == b[k].shape
            fa, fb = a[k][: len(a[k]) // 4 * 4].view(np.float32), b[k][: len(b[k]) // 4 * 4].view(np.float32)
            ia, ib = a[k][: len(a[k]) // 4 * 4].view(np.int32), b[k][: len(b[k]) // 4 * 4].view(np.int32)
            same = ia == ib
            ok = np.isfinite(fa) & np.isfinite(fb)
            assert np.all(same | (ok & (np.abs(fa - fb) <= 2e-6 * np.maximum(1e-3, np.abs(fb)))))   # ints equal; floats to an ulp or two
        elif len(a[k]):
            assert np.abs(a[k] - b[k]).max() <= 1e-9 * max(1.0, np.abs(b[k]).max()), k
    assert n_edges > 30


def test_c5_shape_eight_robots_loopback_properties(ctx):
    """BASELINE.json C5 at its real shape on one GPU: 8 robots x 5000 vertices, a round every 50 vertices, R = 8 wire slices
    of 128 edges (49 KB per robot), 24 rounds in loopback.  No oracle at this size: properties -- every robot with requests
    builds its condensed graphs, every slice addressed to a robot is ingested by it in the next round, no message is ever
    left out for capacity, chi2 stays finite and each robot's solve does not diverge, the received edges are in the graphs."""
    nr, n_rounds = 8, 24
    R = synth.make_multi_robot(nr, 5000, 20000, seed=777)
    rounds = _rounds(lambda r: RobotGraph(ctx, r, nr, cap_edges=128), R, 50, nr)
    assert rounds[0].g.wire_bytes() == 4 * (2 + 2 * nr) + nr * 128 * 44 + nr * 128 * 4          # 49 224 bytes
    from cg_mrslam_amd.mrslam import LoopbackExchange
    ex = LoopbackExchange([r.g for r in rounds])
    sent_prev = None
    built_total, ingested_total = 0, 0
    for t in range(n_rounds):
        for r in rounds:
            r.grow()
            assert r.optimize() == 0
            assert np.all(np.isfinite(r.last_chi2)) and r.last_chi2[-1] <= r.last_chi2[0] * (1 + 1e-9)
        n_in = ex.finish_all()
        if sent_prev is not None:
            # slice p of robot q's buffer was addressed to robot p: p accepted exactly the edges whose end points it knows,
            # and never more than were sent; a non-empty slice whose end points are all known is ingested whole
            for p in range(nr):
                for q in range(nr):
                    if q != p:
                        assert 0 <= n_in[p][q] <= sent_prev[q][p]
            ingested_total += int(sum(int(np.sum(x)) for x in n_in))
        built = [r.condense() for r in rounds]
        for r, b in zip(rounds, built):
            asked = sum(1 for q in range(nr) if q != r.g.robot and len(r.g.closures(q, "out")) >= 2)
            assert b == asked                                # a condensed graph for every peer that asked for >= 2 vertices
        built_total += sum(built)
        ex.start_all()
        sent_prev = []
        for r in rounds:
            robot, n_e, n_c, edges, clos = unpack_wire(r.g.pack_host(), nr, 128)
            assert robot == r.g.robot and n_e[robot] == 0
            sent_prev.append(n_e.copy())
    ex.finish_all()
    assert built_total > 50 and ingested_total > 100
    assert all(r.g.skipped_messages() == 0 for r in rounds)
    assert sum(r.g.counts()["received_edges"] for r in rounds) > 50
    assert all(r.g.counts()["vertices"] >= n_rounds * 50 for r in rounds)


def test_c5_rounds_with_a_context_per_robot_as_one_rank_per_robot_has():
    """The C5 rounds of two robots, each with a context (and so an analysis cache) of its own -- what one rank per robot
    gives; the tests above share one context, where every solve sees another robot's graph and the cached ordering is
    never extended.  Round 3's incremental ordering took graphs whose new edges joined two old vertices of different
    dissection subtrees and handed a wrong structure to the factorisation: a Cholesky failure in robot 0's condensed graph
    in round 40.  Sixty rounds here: every solve and every condensed graph succeeds, edges arrive."""
    from cg_mrslam_amd import Context
    from cg_mrslam_amd.mrslam import LoopbackExchange
    nr, n_rounds = 2, 60
    ctxs = [Context(0) for _ in range(nr)]
    R = synth.make_multi_robot(nr, 5000, 20000, seed=777)
    rounds = [RobotRounds(RobotGraph(ctxs[r], r, nr, cap_edges=128), RobotWorld(R, r, chunk=50)) for r in range(nr)]
    ex = LoopbackExchange([r.g for r in rounds])
    built = 0
    for t in range(n_rounds):
        for r in rounds:
            r.grow()
            assert r.optimize() == 0, (t, r.g.robot)
            assert np.all(np.isfinite(r.last_chi2)) and r.last_chi2[-1] <= r.last_chi2[0] * (1 + 1e-9)
        ex.finish_all()
        built += sum(r.condense() for r in rounds)            # raises on a failed factorisation
        ex.start_all()
    ex.finish_all()
    assert built > 40 and sum(r.g.counts()["received_edges"] for r in rounds) > 20
    assert any(c.symbolic_cache_stats()["extended"] > 0 for c in ctxs)


def test_condense_on_c5_sized_subgraph_with_100_requested_vertices(ctx, oracle):
    """Row a7 at C5 size: a 5000-vertex / 20000-edge sub-graph, ~100 requested vertices (K ~ 100), against the oracle;
    and the same through the flat-array entry point cgmr_condense."""
    g = synth.make_pose_graph(5000, 20000, seed=321, id_base=30000)
    rg = RobotGraph(ctx, 3, 4)
    rg.add_vertices(g["ids"], g["poses"], g["fixed"])
    rg.add_edges(g["ids"][g["edge_from"]], g["ids"][g["edge_to"]], g["meas"], g["info"])
    rc, chi = rg.optimize(5)
    assert rc == 0
    want = g["ids"][np.sort(np.random.default_rng(5).choice(5000, size=101, replace=False))]
    rg.insertOutClosure(1, want)
    assert rg.computeCondensedGraph(1) == 1
    gid, to, est, iu = rg.condensed(1)
    p = rg.poses()
    idx = (want - 30000).astype(np.int32)
    c = p[idx, :2].mean(axis=0)
    gauge = int(idx[np.argmin(np.sqrt(((p[idx, :2] - c) ** 2).sum(axis=1)))])
    assert gid == 30000 + gauge and len(to) == 100
    n, to_o, est_o, iu_o, _ = oracle.condense(p, g["edge_from"], g["edge_to"], g["meas"], g["info"], gauge, idx)
    assert n == 100 and np.array_equal(to - 30000, to_o)
    assert np.abs(est - est_o).max() < 1e-6 and np.abs(iu - iu_o).max() <= 1e-4 * np.abs(iu_o).max()
    to_f, est_f, iu_f, _ = ctx.condense(p, g["edge_from"], g["edge_to"], g["meas"], g["info"], gauge, idx)
    # (the robot graph's spanning-tree guess carries the orientations along by angle addition, the flat-array entry point
    # asks libm per vertex: the guesses differ by rounding, and the one GN iteration from a spanning-tree guess amplifies
    # that by cond(H) -- 1e-8 m between the two entry points, against a parity bar of 1e-6)
    assert np.array_equal(to_f, to_o) and np.abs(est_f - est).max() < 1e-7 and np.abs(iu_f - iu).max() <= 1e-6 * np.abs(iu).max()
    s = rg.last_seconds()
    print(f"C5-sized condense, K = 100: {1e3 * s['condense']:.2f} ms; optimize(5) {1e3 * s['optimize']:.2f} ms")


def test_native_rccl_allgather_single_rank(ctx):
    """cgmr_comm_* / cgmr_allgather_condensed with a one-rank RCCL communicator: librccl resolves, the collective runs
    on the side stream behind the context's stream, cgmr_comm_wait orders the ingest after it."""
    lib = ctx.lib
    uid = np.zeros(128, dtype=np.uint8)
    assert lib.cgmr_comm_unique_id(C.c_void_p(uid.ctypes.data)) == 0
    comm = C.c_void_p()
    assert lib.cgmr_comm_create(ctx.h, C.c_int(1), C.c_int(0), C.c_void_p(uid.ctypes.data), C.byref(comm)) == 0, \
        lib.cgmr_last_error(ctx.h)
    info = np.zeros(3, dtype=np.int32)
    assert lib.cgmr_comm_info(comm, C.c_void_p(info.ctypes.data)) == 0 and list(info[:2]) == [1, 0]     # what librccl itself reports
    g = RobotGraph(ctx, 0, 1)
    g.add_vertices([0, 1, 2], np.zeros((3, 3)), [1, 0, 0])
    g.insertInClosure(0, [1, 2])
    g.pack(0)
    rc = lib.cgmr_allgather_condensed(ctx.h, comm, C.c_void_p(g.send_buffer()), C.c_size_t(g.wire_bytes()), C.c_void_p(g.recv_buffer()))
    assert rc == 0, lib.cgmr_last_error(ctx.h)
    assert lib.cgmr_comm_wait(ctx.h, comm) == 0
    n = g.ingest(0)
    assert list(n) == [0]
    ctx.synchronize()
    sent = g.pack_host()
    s = C.c_double()
    assert lib.cgmr_comm_last_seconds(comm, C.byref(s)) == 0 and s.value >= 0
    robot, n_e, n_c, edges, clos = unpack_wire(sent, 1, g.cap)
    assert robot == 0 and list(n_c) == [2] and list(clos[0, :2]) == [1, 2]
    # ... and behind a batch of condensed graphs that was NOT waited for (the context's side stream): the all-gather must
    # carry the star the batch writes, not what the send buffer held when the call was made
    import torch
    gd = synth.make_pose_graph(600, 1800, seed=3)
    g2 = RobotGraph(ctx, 0, 2, async_condense=True)
    g2.add_vertices(gd["ids"], gd["poses"], gd["fixed"])
    g2.add_edges(gd["ids"][gd["edge_from"]], gd["ids"][gd["edge_to"]], gd["meas"], gd["info"])
    assert g2.optimize(3)[0] == 0
    want = gd["ids"][[4, 90, 222, 410, 577]]
    g2.insertOutClosure(1, want)
    assert g2.computeCondensedGraph(1) == 1                        # queued on the side stream
    g2.pack(0)
    wb = g2.wire_bytes()
    recv = torch.zeros(wb, dtype=torch.uint8, device="cuda")      # a one-rank gather: my buffer comes back
    rc = lib.cgmr_allgather_condensed(ctx.h, comm, C.c_void_p(g2.send_buffer()), C.c_size_t(wb), C.c_void_p(recv.data_ptr()))
    assert rc == 0, lib.cgmr_last_error(ctx.h)
    assert lib.cgmr_comm_last_seconds(comm, C.byref(s)) == 0      # (waits for the collective)
    got = recv.cpu().numpy()
    robot, n_e, n_c, edges, clos = unpack_wire(got, 2, g2.cap)
    assert robot == 0 and list(n_e) == [0, len(want) - 1]
    gid, to, est, iu = g2.condensed(1)                             # (waits for the batch)
    assert np.array_equal(edges[1, :n_e[1]]["to"], to) and np.all(edges[1, :n_e[1]]["from"] == gid)
    assert np.abs(edges[1, :n_e[1]]["est"] - est.astype(np.float32)).max() == 0
    lib.cgmr_comm_destroy.restype = None
    lib.cgmr_comm_destroy(comm)


def test_bench_two_ranks_on_one_gpu():
    """``bench.py --gpus 2`` spawns its two ranks itself; on this one-GPU box they share cuda:0 and the collectives run
    over gloo (CGMR_BENCH_SINGLE_DEVICE / CGMR_BENCH_BACKEND).  The C5 exchange leg must really exchange."""
    env = dict(os.environ, CGMR_BENCH_SINGLE_DEVICE="1", CGMR_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--vertices", "2000",
           "--edges", "7000", "--match-pairs", "0", "--c5-vertices", "800", "--c5-edges", "2800", "--c5-chunk", "100"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["warm"]["bit_identical_to_cold"]
    ex = out["exchange"]
    assert ex["robots"] == 2 and ex["rounds"] == 8 and ex["transport"] == "host"
    assert ex["condensed_graphs_built_total"] > 0 and ex["condensed_edges_received_total"] > 0 and ex["status_rank0"] == 0


def test_bench_eight_ranks_on_one_gpu():
    """``bench.py --gpus 8`` on ONE GPU (all ranks share cuda:0, collectives over gloo): everything of the 8-rank run except
    xGMI -- eight Python ranks, eight helper pools placed by LOCAL_RANK on different cache groups (where the host has them),
    R = 8 addressing through the real rank path (8 wire slices per buffer), the solo-vs-real accounting of the exchange
    leg.  Reduced sizes; the full-size line is kept under profiles/ (tools/bench_8ranks_one_gpu.sh)."""
    env = dict(os.environ, CGMR_BENCH_SINGLE_DEVICE="1", CGMR_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--vertices", "2000",
           "--edges", "7000", "--match-pairs", "0", "--c5-vertices", "600", "--c5-edges", "2100", "--c5-chunk", "50", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 8 and out["value"] > 0 and out["warm"]["bit_identical_to_cold"]
    ex = out["exchange"]
    assert ex["robots"] == 8 and ex["rounds"] == 12 and ex["transport"] == "host"
    assert ex["bytes_gathered_per_rank_per_round"] == 8 * (4 * (2 + 16) + 8 * 128 * 44 + 8 * 128 * 4)
    assert ex["condensed_graphs_built_total"] > 0 and ex["condensed_edges_received_total"] > 0 and ex["status_rank0"] == 0
    assert ex["messages_skipped_over_capacity_total"] == 0 and 0 < ex["weak_scaling_efficiency_vs_solo"] < 2
    # every rank accounts for itself (round 5: the first multi-GPU run has to be self-diagnosing): transport and why, the ranks its
    # native communicator reports, its sampled all-gather time, its own weak-scaling figure
    ranks = ex["ranks"]
    assert len(ranks) == 8 and sorted(d["rank"] for d in ranks) == list(range(8))
    for d in ranks:
        assert d["transport"] == "host" and d["transport_fallback_reason"] is None and d["native_comm_ranks"] is None
        assert d["world"] == 8 and d["round_ms_mean"] > 0 and d["solo_round_ms_mean"] > 0 and 0 < d["weak_scaling_efficiency_vs_solo"] < 2
        assert d["status"] == 0 and d["failed_condensed_batches"] == 0
    assert ex["transport_is_native_rccl_on_every_rank"] is False and "WARNING" not in ex      # (gloo on purpose: nothing to warn about)
    pools = out["host_pool_of_every_rank"]
    assert len(pools) == 8 and sorted(q["rank"] for q in pools) == list(range(8))
    pinned = [q for q in pools if q["pinned"]]
    if len(pinned) == 8 and pools[0]["cpus_allowed"] >= 128:       # a host with >= 8 cache groups: no two pools around the same core
        assert len({q["home_cpu"] for q in pinned}) == 8, pools


def test_select_optimal_gauge_matches_oracle_backend(ctx, oracle):
    """computeCondensedGraph(robot, optimal = true): selectOptimalGauge (condensed_graph_buffer.cpp:252-288) -- every
    requested vertex in turn as the gauge, the star with the smallest sum of det(information^-1) wins -- against the same
    search on the oracle backend: same gauge (the runner-up is clearly worse), same edges."""
    import oracle_backend as ob
    from ref_condensed import RefRobotGraph
    g = synth.make_pose_graph(1500, 5000, seed=77, id_base=20000)
    want = g["ids"][np.sort(np.random.default_rng(9).choice(1500, size=12, replace=False))]
    graphs = [RobotGraph(ctx, 2, 4), RefRobotGraph(ob.OracleContext(), 2, 4)]
    for rg in graphs:
        rg.add_vertices(g["ids"], g["poses"], g["fixed"])
        rg.add_edges(g["ids"][g["edge_from"]], g["ids"][g["edge_to"]], g["meas"], g["info"])
        assert rg.optimize(6)[0] == 0
        rg.insertOutClosure(0, want)
    graphs[0].set_optimal_gauge(True)
    graphs[1].optimal_gauge = True
    assert graphs[0].computeCondensedGraph(0) == 1 and graphs[1].computeCondensedGraph(0) == 1
    gid_a, to_a, est_a, iu_a = graphs[0].condensed(0)
    gid_b, to_b, est_b, iu_b = graphs[1].condensed(0)
    u = sorted(graphs[1].uncertainties.values())
    assert u[1] > u[0] * (1 + 1e-6)                          # the choice is not a coin flip
    assert gid_a == gid_b and np.array_equal(to_a, to_b) and len(to_a) == 11
    assert np.abs(est_a - est_b).max() < 1e-6 and np.abs(iu_a - iu_b).max() <= 1e-4 * np.abs(iu_b).max()
    # and it differs from the centroid choice on this graph, i.e. the search did something
    graphs[0].set_optimal_gauge(False)
    graphs[0].computeCondensedGraph(0)
    assert graphs[0].condensed(0)[0] != gid_a or len(set(graphs[1].uncertainties.values())) == 1


def test_async_condensed_graphs_on_the_side_stream_equal_the_synchronous_ones():
    """cgmr_graph_compute_condensed_async: the round's condensed graphs queued on the context's side stream and not waited for,
    the message packed and delivered behind them on the device (RobotGraph.deliver), the next round's grow / analysis / solve
    running meanwhile.  Three robots with a context each, 10 rounds: every round's condensed graphs, every ingest and the
    final poses equal those of the synchronous rounds with the host loopback (same kernels on the same snapshot; 1e-9: the
    two runs may split the chained backward solve at another level)."""
    from cg_mrslam_amd import Context
    nr, n_rounds, chunk = 3, 10, 120
    R = _meeting_world(nr, 1200, 4000, min_shared=6)

    def run(async_on):
        ctxs = [Context(0) for _ in range(nr)]
        rounds = [RobotRounds(RobotGraph(ctxs[r], r, nr, cap_edges=128, async_condense=async_on), RobotWorld(R, r, chunk=chunk))
                  for r in range(nr)]
        from cg_mrslam_amd.mrslam import LoopbackExchange
        ex = LoopbackExchange([r.g for r in rounds], device=async_on)
        log = []
        for t in range(n_rounds):
            for r in rounds:
                r.grow()
                assert r.optimize() == 0
            n_in = ex.finish_all()
            built = [r.condense() for r in rounds]
            ex.start_all()
            # (reading the condensed graphs waits for the batch: done AFTER the exchange was queued, so the asynchronous run
            # has packed and delivered behind a batch that was still in flight)
            cond = [[r.g.condensed(p) for p in range(nr) if p != r.g.robot] for r in rounds]
            log.append((n_in, built, cond, [r.last_chi2.copy() for r in rounds]))
        ex.finish_all()
        for r in rounds:
            r.g.condensed_wait()
        return log, [r.g.poses() for r in rounds], [r.g.counts() for r in rounds]

    log_s, poses_s, counts_s = run(False)
    log_a, poses_a, counts_a = run(True)
    assert counts_s == counts_a
    edges = 0
    for (n_in_s, built_s, cond_s, chi_s), (n_in_a, built_a, cond_a, chi_a) in zip(log_s, log_a):
        assert built_s == built_a
        assert (n_in_s is None) == (n_in_a is None)
        if n_in_s is not None:
            assert [list(x) for x in n_in_s] == [list(x) for x in n_in_a]
        for cs, ca in zip(cond_s, cond_a):
            for (gid_s, to_s, est_s, iu_s), (gid_a, to_a, est_a, iu_a) in zip(cs, ca):
                assert gid_s == gid_a and np.array_equal(to_s, to_a)
                edges += len(to_s)
                if len(to_s):
                    assert np.abs(est_s - est_a).max() <= 1e-9 and np.abs(iu_s - iu_a).max() <= 1e-9 * np.abs(iu_s).max()
        for a, b in zip(chi_s, chi_a):
            np.testing.assert_allclose(a, b, rtol=1e-9)
    assert edges > 30
    for a, b in zip(poses_s, poses_a):
        assert np.abs(a - b).max() <= 1e-9


def test_robots_taking_turns_with_whole_rounds_equal_the_lock_step_rounds():
    """``TakeTurns``: robot after robot does its whole round (grow, solve, ingest, condensed graphs, pack, deliver) instead of
    everybody solving, then everybody ingesting, ... -- a rank's timeline on one device.  Two receive buffers in turn keep the
    data flow that of the lock-step order (round t's message is ingested in round t + 1): same condensed graphs, same ingests,
    same final poses."""
    from cg_mrslam_amd import Context
    from cg_mrslam_amd.mrslam import LoopbackExchange, TakeTurns
    nr, n_rounds, chunk = 3, 8, 120
    R = _meeting_world(nr, 1200, 4000, min_shared=6)

    def make():
        ctxs = [Context(0) for _ in range(nr)]
        return [RobotRounds(RobotGraph(ctxs[r], r, nr, cap_edges=128, async_condense=True), RobotWorld(R, r, chunk=chunk)) for r in range(nr)]
    a = make()
    ex = LoopbackExchange([r.g for r in a], device=True)
    log_a = []
    for t in range(n_rounds):
        for r in a:
            r.grow(); assert r.optimize() == 0
        n_in = ex.finish_all()
        built = [r.condense() for r in a]
        ex.start_all()
        log_a.append((n_in, built))
    b = make()
    tt = TakeTurns(b)
    log_b = [tt.round() for _ in range(n_rounds)]
    got = 0
    for t in range(n_rounds):
        n_in_a, built_a = log_a[t]
        for r in range(nr):
            n_in_b, built_b = log_b[t][r]
            assert built_b == built_a[r]
            assert (n_in_b is None) == (n_in_a is None)
            if n_in_b is not None:
                assert list(n_in_b) == list(n_in_a[r])
                got += int(np.sum(n_in_b))
    assert got > 20
    for ra, rb in zip(a, b):
        ra.g.condensed_wait(); rb.g.condensed_wait()
        for p in range(nr):
            if p != ra.g.robot:
                ga, gb = ra.g.condensed(p), rb.g.condensed(p)
                assert ga[0] == gb[0] and np.array_equal(ga[1], gb[1])
                if len(ga[1]):
                    assert np.abs(ga[2] - gb[2]).max() <= 1e-9 and np.abs(ga[3] - gb[3]).max() <= 1e-9 * np.abs(ga[3]).max()
        assert np.abs(ra.g.poses() - rb.g.poses()).max() <= 1e-9


def test_failed_asynchronous_batch_sends_nothing_and_reports_itself():
    """A pass of a batch that was not waited for fails (an edge with a negative-definite information matrix: a negative pivot
    for certain).  The message packed BEHIND the batch, before anybody knew, must carry no edges for the batch's peers -- the
    counts are taken back on the device (k_wire_fix_counts) -- and the failure is reported by cgmr_graph_condensed_wait.  The
    control run without the bad edge delivers the star."""
    from cg_mrslam_amd import Context
    from cg_mrslam_amd._lib import CgmrError
    g = synth.make_pose_graph(400, 1200, seed=11, id_base=0)
    want = g["ids"][[5, 60, 150, 260, 399]]

    def run(bad):
        c0, c1 = Context(0), Context(0)
        g0 = RobotGraph(c0, 0, 2, async_condense=True)
        g1 = RobotGraph(c1, 1, 2, async_condense=True)
        g0.add_vertices(g["ids"], g["poses"], g["fixed"])
        ef, et, meas, info = g["ids"][g["edge_from"]], g["ids"][g["edge_to"]], g["meas"], g["info"].copy()
        if bad:
            info[7] = [-1e6, 0, 0, -1e6, 0, -1e6]
        g0.add_edges(ef, et, meas, info)
        # robot 1 holds copies of the requested vertices (it would accept the star) and one vertex of its own
        g1.add_vertices(np.concatenate([[10000], want]), np.zeros((1 + len(want), 3)), None)
        g0.insertOutClosure(1, want)
        assert g0.computeCondensedGraph(1) == 1                    # queued, not waited for
        g0.pack(0)
        g0.deliver(g1)
        g1.pack(0)
        g1.deliver(g0)
        n_in = g1.ingest_delivered()
        return g0, int(n_in[0])

    g0, n_ok = run(False)
    g0.condensed_wait()
    assert n_ok == len(want) - 1
    g0, n_bad = run(True)
    assert n_bad == 0                                              # nothing of the failed batch travelled
    with pytest.raises(CgmrError) as e:
        g0.condensed_wait()
    assert e.value.code <= -100
    assert g0.condensed(1)[1].size == 0
    g0.close()                                                     # (the exception info above would keep it alive until interpreter exit)


def test_a_failed_batch_does_not_stop_the_next_round_and_a_skipped_delivery_is_no_message():
    """Round 5 (advisor findings of round 4).  (1) The failure of a batch nobody waited for used to come back from the NEXT
    cgmr_graph_compute_condensed call, which then built nothing: now the next round is built regardless and the failure is on
    record (failed_batches).  (2) cgmr_graph_deliver / cgmr_graph_ingest_delivered pair up by call count: a robot that skips a
    round must not have its previous message ingested again two rounds later -- a buffer that does not hold the expected
    delivery is ingested as "no message"; a robot cannot deliver to itself or before it has packed anything."""
    from cg_mrslam_amd import Context
    from cg_mrslam_amd._lib import CgmrError
    g = synth.make_pose_graph(400, 1200, seed=11, id_base=0)
    want = g["ids"][[5, 60, 150, 260, 399]]
    c0, c1 = Context(0), Context(0)
    g0 = RobotGraph(c0, 0, 2, async_condense=True)
    g1 = RobotGraph(c1, 1, 2, async_condense=True)
    with pytest.raises(CgmrError):
        g0.deliver(g1)                                             # nothing packed yet
    with pytest.raises(CgmrError):
        g0.deliver(g0)
    g0.add_vertices(g["ids"], g["poses"], g["fixed"])
    info = g["info"].copy()
    info[7] = [-1e6, 0, 0, -1e6, 0, -1e6]                          # a negative pivot for certain
    g0.add_edges(g["ids"][g["edge_from"]], g["ids"][g["edge_to"]], g["meas"], info)
    g1.add_vertices(np.concatenate([[10000], want]), np.zeros((1 + len(want), 3)), None)
    g0.insertOutClosure(1, want)
    assert g0.computeCondensedGraph(1) == 1                        # round 1: queued, fails on the device
    g0.pack(0); g0.deliver(g1)
    assert int(g1.ingest_delivered()[0]) == 0                      # nothing of the failed batch travelled
    # round 2: the bad edge is still there, so this batch fails as well -- but it IS built (round 4 returned round 1's error here)
    assert g0.computeCondensedGraph(1) == 1
    assert g0.failed_batches() == 1
    with pytest.raises(CgmrError):
        g0.condensed_wait()
    assert g0.failed_batches() == 2
    # (2) pairing: g1 has ingested once, g0 has delivered once.  g0 now SKIPS a round (no deliver); g1's second ingest finds the
    # other buffer without a second delivery -> no message, not a stale one
    g1.pack(0)
    assert int(g1.ingest_delivered()[0]) == 0
    g0.close(); g1.close()


def test_a_sender_two_rounds_ahead_loses_one_message_not_two():
    """Round 6 (advisor finding of round 5).  g0 delivers rounds 0, 1 and 2 before g1 ingests anything: delivery 2 lands in
    receive buffer 0 over delivery 0, which is lost.  g1's ingest of round 0 must then see "no message" from g0 WITHOUT clearing
    the slice -- it holds round 2's message, which the third ingest digests (round 5 cleared it: rounds 0 and 2 both gone)."""
    from cg_mrslam_amd import Context
    g = synth.make_pose_graph(400, 1200, seed=11, id_base=0)
    want = g["ids"][[5, 60, 150, 260, 399]]
    c0, c1 = Context(0), Context(0)
    g0 = RobotGraph(c0, 0, 2)
    g1 = RobotGraph(c1, 1, 2)
    g0.add_vertices(g["ids"], g["poses"], g["fixed"])
    g0.add_edges(g["ids"][g["edge_from"]], g["ids"][g["edge_to"]], g["meas"], g["info"])
    g1.add_vertices(np.concatenate([[10000], want]), np.zeros((1 + len(want), 3)), None)
    g0.insertOutClosure(1, want)
    assert g0.computeCondensedGraph(1) == 1
    for _ in range(3):                                             # rounds 0, 1, 2 of g0, nothing ingested in between
        g0.pack(0); g0.deliver(g1)
    assert int(g1.ingest_delivered()[0]) == 0                      # round 0: overwritten by round 2 -> no message, slice untouched
    assert int(g1.ingest_delivered()[0]) == len(want) - 1          # round 1
    assert int(g1.ingest_delivered()[0]) == len(want) - 1          # round 2 is still there
    assert int(g1.ingest_delivered()[0]) == 0                      # round 3: never delivered
    g0.close(); g1.close()
