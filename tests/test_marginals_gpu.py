"""GPU parity tests of the marginal-covariance / condensed-measurement path against the CPU oracle.
Tolerances ("hard parts"): condensed edge mean <= 1e-6, information <= 1e-4 relative; marginal
blocks <= 1e-6 relative to the largest entry."""
import numpy as np
import pytest

from cg_mrslam_amd import synth
from cg_mrslam_amd.condensed import RobotGraph

pytestmark = pytest.mark.gpu


def _opt(ctx, g, iters=8):
    a = (g["edge_from"], g["edge_to"], g["meas"], g["info"])
    rc, p, _ = ctx.gn_optimize(g["poses"], g["fixed"], *a, iters)
    assert rc == 0
    return p, a


@pytest.mark.parametrize("V,E,nq", [(60, 110, 4), (400, 1300, 17), (3000, 11000, 60)])
def test_marginals_match_oracle(ctx, oracle, V, E, nq):
    g = synth.make_pose_graph(V, E, seed=40 + nq)
    p, a = _opt(ctx, g)
    query = np.unique(np.linspace(0, V - 1, nq).astype(np.int32))       # includes the fixed vertex 0 -> zeros
    cov = ctx.marginals(p, g["fixed"], *a, query)
    st, want = oracle.marginals(p, g["fixed"], *a, query)
    assert st == 0
    assert np.abs(cov - want).max() <= 1e-6 * np.abs(want).max()
    assert np.all(cov[0] == 0)
    for k in range(1, len(query)):                                      # symmetric positive definite blocks
        assert np.allclose(cov[k], cov[k].T, rtol=1e-9, atol=1e-15)
        assert np.all(np.linalg.eigvalsh(cov[k]) > 0)


@pytest.mark.parametrize("K,Lc,H,nq", [(40, 30, 1, 30), (100, 12, 2, 90)])
def test_marginals_on_hub_graphs(ctx, oracle, K, Lc, H, nq):
    """Fronts with dozens of children (100 chains on 2 hubs: more children than one 64-lane scan of the multi-RHS forward
    solve's live-children list takes in), queries spread over many chains: most (front, group) pairs carry nothing, the live
    ones collect from one child among many."""
    from cg_mrslam_amd._lib import gn_symbolic_info
    g = synth.make_hub_graph(K, Lc, H)
    assert gn_symbolic_info(len(g["poses"]), g["fixed"], g["edge_from"], g["edge_to"])["max_children"] > (64 if K >= 100 else 8)
    p, a = _opt(ctx, g, iters=5)
    V = len(g["poses"])
    query = np.unique(np.linspace(0, V - 2, nq).astype(np.int32))
    cov = ctx.marginals(p, g["fixed"], *a, query)
    st, want = oracle.marginals(p, g["fixed"], *a, query)
    assert st == 0
    assert np.abs(cov - want).max() <= 1e-6 * np.abs(want).max()


def test_covariance_estimate_matches_oracle(ctx, oracle):
    g = synth.make_pose_graph(800, 2600, seed=44)
    p, a = _opt(ctx, g)
    query = np.arange(700, 790, 9).astype(np.int32)
    cov = ctx.covariance_estimate(p, *a, 799, query)                     # gauge = last vertex (graph_slam.cpp:316)
    st, want = oracle.covariance_estimate(p, *a, 799, query)
    assert st == 0
    assert np.abs(cov - want).max() <= 1e-6 * np.abs(want).max()


def test_condense_matches_oracle_and_chain_closed_form(ctx, oracle):
    g = synth.make_pose_graph(1500, 5000, seed=45)
    p, a = _opt(ctx, g)
    query = np.array([100, 400, 420, 777, 1200, 1499], dtype=np.int32)
    to, est, iu, cov = ctx.condense(p, *a, 420, query)
    n, to2, est2, iu2, cov2 = oracle.condense(p, *a, 420, query)
    assert n == len(to) == 5 and np.array_equal(to, to2)
    assert np.abs(est - est2).max() <= 1e-6
    assert np.abs(iu - iu2).max() <= 1e-4 * np.abs(iu2).max()
    assert np.abs(cov - cov2).max() <= 1e-6 * np.abs(cov2).max()
    # closed form on a straight chain: variances add up
    V = 5
    poses = np.array([[float(k), 0, 0] for k in range(V)])
    ef = np.arange(V - 1, dtype=np.int32)
    meas = np.tile([1.0, 0, 0], (V - 1, 1))
    info = np.tile([100.0, 0, 0, 100, 0, 1000], (V - 1, 1))
    to, est, iu, cov = ctx.condense(poses, ef, ef + 1, meas, info, 0, np.arange(V, dtype=np.int32))
    for k in range(V - 1):
        assert abs(cov[k][0, 0] - (k + 1) / 100.0) < 1e-12 and abs(cov[k][2, 2] - (k + 1) / 1000.0) < 1e-12
        np.testing.assert_allclose(est[k], [k + 1.0, 0, 0], atol=1e-9)


def test_two_robot_round_reduces_error(ctx):
    """One full multi-robot round on one GPU with the whole graphs loaded at once (robots run one after the other, the
    'wire' is pack_host / ingest_host): optimise, build the condensed graph each peer asked for, exchange, re-optimise.
    The received condensed edges must tie the foreign vertices together consistently: chi2 stays ~dof and the foreign
    poses move towards their truth.  (The oracle-compared multi-round version is tests/test_multirobot_gpu.py.)"""
    for seed in range(46, 80):                            # first seed whose two walks actually meet
        R = synth.make_multi_robot(2, 1200, 4000, seed=seed)
        if all(len(R[r]["in_closures"].get(1 - r, [])) >= 8 for r in range(2)):
            break
    graphs = []
    for r in range(2):
        gr = R[r]
        g = RobotGraph(ctx, r, 2)
        g.add_vertices(gr["ids"], gr["poses_all"], gr["fixed_all"])
        g.add_edges(gr["ids"][gr["ef_all"]], gr["ids"][gr["et_all"]], gr["meas_all"], gr["info_all"])
        for q, ids in gr["in_closures"].items():
            g.insertInClosure(q, ids)
        graphs.append(g)
    for g in graphs:
        assert g.optimize(6)[0] == 0
    wire = np.concatenate([g.pack_host() for g in graphs])          # round 1: requests only
    for g in graphs:
        assert list(g.ingest_host(wire)) == [0, 0]
    for r, g in enumerate(graphs):
        assert g.computeCondensedGraph(1 - r) == 1
        gid, to, est, iu = g.condensed(1 - r)
        assert len(to) == len(g.closures(1 - r, "out")) - 1 > 3
        assert np.all(np.isfinite(est)) and np.all(np.isfinite(iu))
    wire = np.concatenate([g.pack_host() for g in graphs])          # round 2: condensed edges travel
    before = []
    for r, g in enumerate(graphs):
        n_own = R[r]["n_own"]
        before.append(np.abs(g.poses()[n_own:, :2] - R[r]["truth_all"][n_own:, :2]).mean())
        n = g.ingest_host(wire)
        assert n[1 - r] == len(graphs[1 - r].condensed(r)[1])
    for r, g in enumerate(graphs):
        rc, chi = g.optimize(6)
        assert rc == 0
        c = g.counts()
        dof = 3 * (c["own_edges"] + c["received_edges"]) - 3 * (c["vertices"] - 1)
        assert chi[-1] < 1.5 * dof
        n_own = R[r]["n_own"]
        after = np.abs(g.poses()[n_own:, :2] - R[r]["truth_all"][n_own:, :2]).mean()
        assert after < before[r] * 1.05                  # never worse; usually clearly better
