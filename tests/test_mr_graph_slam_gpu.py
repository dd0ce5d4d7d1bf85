"""cg_mrslam in SIM modality (BASELINE config C4) end to end on the GPU: every robot runs ``MRGraphSLAMDriver`` on a
device-resident ``RobotGraph`` with the GPU matchers, and the run is compared with the same driver on the oracle backend
(numpy condensed-graph buffer + CPU oracle solver / matcher): same key frames, same messages, same inter-robot closures,
matcher output bit-identical, estimates and condensed edges within the Gauss-Newton tolerance."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from cg_mrslam_amd import synth
from cg_mrslam_amd.condensed import RobotGraph
from cg_mrslam_amd.matcher import LCScanMatcher, ScanMatcher
from cg_mrslam_amd.mr_graph_slam import GraphCommSim, MRGraphSLAMDriver, run_cg_mrslam

import oracle_backend as OB
from ref_condensed import RefRobotGraph

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def _make(ctx, r, n_robots, la, gpu):
    if gpu:
        s = MRGraphSLAMDriver(ctx, ScanMatcher(ctx, *la), LCScanMatcher(ctx, *la), RobotGraph(ctx, r, n_robots), r, n_robots,
                              windowLoopClosure=5, minInliers=4)
    else:
        octx = OB.OracleContext()
        s = MRGraphSLAMDriver(octx, OB.close_matcher(la), OB.lc_matcher(la), RefRobotGraph(octx, r, n_robots), r, n_robots,
                              windowLoopClosure=5, minInliers=4)
    s.setInterRobotClosureParams(0.15, 3, 5)
    return s


def _run(ctx, team, gpu, detect=False):
    la = (team[0]["n_beams"], team[0]["angle_min"], team[0]["angle_inc"], team[0]["max_range"])
    n = len(team)
    slams = [_make(ctx, r, n, la, gpu) for r in range(n)]
    for s in slams:
        s.setDetectRobotInRange(detect)
    comm = GraphCommSim(slams)
    t0 = time.time()
    run_cg_mrslam(slams, team, comm=comm, linearUpdate=0.5)
    return slams, comm, time.time() - t0


def _strip(log):
    # chi2 of the closure checker is a float that may differ in the last digits between the two solvers
    cut = {"lcc": 2, "mr_lcc": 3, "verify": 4}
    return [l[:cut[l[0]]] if l[0] in cut else l for l in log]


def _compare(a_slams, b_slams):
    for a, b in zip(a_slams, b_slams):
        np.testing.assert_array_equal(a.g.ids, b.g.ids)
        assert a.edge_kind == b.edge_kind
        np.testing.assert_array_equal(a.g.edge_from, b.g.edge_from)
        np.testing.assert_array_equal(a.g.edge_to, b.g.edge_to)
        own = np.array([k != "cond" for k in a.edge_kind])
        np.testing.assert_array_equal(a.g.meas[own], b.g.meas[own])          # odometry + matcher output: bit-identical
        np.testing.assert_allclose(a.g.meas[~own], b.g.meas[~own], rtol=0, atol=2e-5)      # condensed edges, float32 wire
        np.testing.assert_allclose(a.g.info[~own], b.g.info[~own], rtol=2e-4, atol=1e-2)
        assert np.abs(a.g.poses - b.g.poses).max() < 1e-6
        assert _strip(a.log) == _strip(b.log)
        for p in range(a.nRobots):
            if p == a.idRobot:
                continue
            np.testing.assert_array_equal(a.rg.closures(p, "in"), b.rg.closures(p, "in"))
            np.testing.assert_array_equal(a.rg.closures(p, "out"), b.rg.closures(p, "out"))
            ga, ta, ea, ia = a.rg.condensed(p)
            gb, tb, eb, ib = b.rg.condensed(p)
            assert ga == gb
            np.testing.assert_array_equal(ta, tb)
            np.testing.assert_allclose(ea, eb, rtol=0, atol=1e-7)
            np.testing.assert_allclose(ia, ib, rtol=1e-6, atol=1e-6)


def test_two_robots_match_oracle_backend(ctx, oracle):
    team = synth.make_robot_team(2, n_steps=110, laps=0.26, gap=3.0)
    a, comm_a, t_gpu = _run(ctx, team, True)
    b, comm_b, t_cpu = _run(ctx, team, False)
    assert comm_a.delivered == comm_b.delivered > 50
    assert [s.bytes_sent for s in comm_a.senders] == [s.bytes_sent for s in comm_b.senders]
    _compare(a, b)
    for s in a:
        kinds = {k: s.edge_kind.count(k) for k in set(s.edge_kind)}
        assert kinds.get("mr", 0) >= 3 and kinds.get("cond", 0) >= 2, kinds
    print(f"2 robots: GPU run {t_gpu:.1f} s, oracle-backed run {t_cpu:.1f} s, {comm_a.delivered} messages, "
          f"{sum(s.bytes_sent for s in comm_a.senders)} bytes")


def test_four_robots_c4_sim_modality_matches_oracle_backend(ctx, oracle):
    """BASELINE config C4 in one process on one GPU: four robots, 3 m apart, so only neighbours are within the 5 m
    communication range; the robots see each other (0.5 m boxes) and verifyMatching gates the closures
    (setDetectRobotInRange): a robot detects the neighbour ahead of it, not the one in its blind rear sector."""
    team = synth.make_robot_team(4, n_steps=90, laps=0.21, gap=3.0, body=0.5)
    a, comm_a, t_gpu = _run(ctx, team, True, detect=True)
    b, comm_b, _ = _run(ctx, team, False, detect=True)
    assert comm_a.delivered == comm_b.delivered
    _compare(a, b)
    for s in a:
        for p in range(4):
            if abs(p - s.idRobot) > 1:
                assert len(s.rg.closures(p, "in")) == 0 and len(s.rg.closures(p, "out")) == 0
    for r in range(3):                                       # r asks the robot ahead, which condenses its graph for r
        assert len(a[r].rg.closures(r + 1, "in")) > 0 and len(a[r + 1].rg.closures(r, "out")) > 0
        assert a[r].edge_kind.count("cond") >= 1
    tp = [team[s.idRobot]["truth"] for s in a]
    for s, t in zip(a, tp):
        own = [q for q in range(s.g.n_vertices) if s.isMyVertex(q)]
        err = max(np.min(np.hypot(t[:, 0] - p[0], t[:, 1] - p[1])) for p in s.g.poses[own])
        assert err < 0.3, err
    print(f"4 robots: GPU run {t_gpu:.1f} s, {comm_a.delivered} messages")


def test_four_robots_a_thread_and_a_context_each_equal_the_sequential_run(ctx):
    """run_cg_mrslam(concurrent=True): between the communication cycles every robot ticks on a thread of its own (the C-ABI
    calls release the interpreter lock), on a context of its own -- what one process per robot does.  Same key frames,
    messages, closures and logs as the robots one after the other on one context; estimates to the Gauss-Newton tolerance
    (a context of its own extends its cached ordering as the graph grows: another elimination order)."""
    from cg_mrslam_amd import Context
    team = synth.make_robot_team(4, n_steps=90, laps=0.21, gap=3.0, body=0.5)
    la = (team[0]["n_beams"], team[0]["angle_min"], team[0]["angle_inc"], team[0]["max_range"])
    a, comm_a, _ = _run(ctx, team, True, detect=True)
    b = [_make(Context(0), r, 4, la, True) for r in range(4)]
    for s in b:
        s.setDetectRobotInRange(True)
    comm_b = GraphCommSim(b)
    run_cg_mrslam(b, team, comm=comm_b, linearUpdate=0.5, concurrent=True)
    assert comm_a.delivered == comm_b.delivered > 100
    assert [s.bytes_sent for s in comm_a.senders] == [s.bytes_sent for s in comm_b.senders]
    _compare(a, b)


def test_cg_mrslam_cli_one_rank_per_robot_equals_one_process(tmp_path):
    """``python -m cg_mrslam_amd.cg_mrslam`` (the cg_mrslam node, sim modality): two robots in one process, then one
    rank per robot (both ranks share this box's only GPU; the all-gather runs over gloo here, RCCL on a multi-GPU node).
    The saved graphs (robot-<id>-<o>, cg_mrslam.cpp:199-202) must agree: same vertices and edges, numbers to rounding -- the two
    runs reach the solver with different call histories (one context shared by both robots never sees a graph grow, a rank's own
    context does and extends its cached ordering: another elimination order, results equal to the last few ulps)."""
    base = [sys.executable, "-m", "cg_mrslam_amd.cg_mrslam", "-nRobots", "2", "-steps", "90", "-laps", "0.21", "-linearUpdate", "0.5",
            "-windowLoopClosure", "5", "-minInliers", "4", "-minInliersMR", "3", "-windowMRLoopClosure", "5"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    p = subprocess.run(base + ["-o", str(tmp_path / "one.g2o")], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    one = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])["robots"]
    assert all(r["edges"].get("mr", 0) >= 1 and r["edges"].get("cond", 0) >= 1 for r in one), one
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", _free_port(), "-m", "cg_mrslam_amd.cg_mrslam"]
    p = subprocess.run(launch + base[3:] + ["-o", str(tmp_path / "ranks.g2o"), "-device", "0", "-backend", "gloo"], env=env,
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    ranks = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])["robots"]
    for a, b in zip(one, sorted(ranks, key=lambda r: r["robot"])):
        assert a["vertices"] == b["vertices"] and a["edges"] == b["edges"] and abs(a["chi2"] - b["chi2"]) <= 1e-9 * abs(a["chi2"])
    for r in range(2):
        la = (tmp_path / f"robot-{r}-one.g2o").read_text().splitlines()
        lb = (tmp_path / f"robot-{r}-ranks.g2o").read_text().splitlines()
        assert len(la) == len(lb)
        for x, y in zip(la, lb):
            tx, ty = x.split(), y.split()
            assert len(tx) == len(ty) and tx[0] == ty[0]
            for u, v in zip(tx[1:], ty[1:]):
                assert u == v or abs(float(u) - float(v)) <= 1e-7 * max(1.0, abs(float(u))), (x, y)


def test_cg_mrslam_cli_rank_mode_over_rccl_single_rank(tmp_path):
    """The rank mode with the default backend (nccl = RCCL): outboxes live in HBM, the all-gather is an RCCL collective.
    One GPU here, so one rank = one robot; the multi-rank path is the gloo test above."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), "-m", "cg_mrslam_amd.cg_mrslam", "-nRobots", "1", "-steps", "60", "-laps", "0.14",
           "-linearUpdate", "0.5", "-o", str(tmp_path / "solo.g2o")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    rep = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])["robots"]
    assert len(rep) == 1 and rep[0]["transport"] == "all-gather/nccl" and rep[0]["own_vertices"] >= 8
    assert rep[0]["max_distance_to_true_path_m"] < 0.2 and (tmp_path / "robot-0-solo.g2o").exists()
