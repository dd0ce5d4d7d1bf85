"""CPU tests of the GN oracle (oracle/gn_oracle.c): golden fixtures from the independent
numpy/SciPy implementation plus the known-answer tests of SURVEY.md section 8(c).
g2o is not available, so these pin the restated algorithm ("parity unpinned" vs g2o)."""
import glob
import os

import numpy as np
import pytest

import ref_numpy as R
from cg_mrslam_amd import synth

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "gn_*.npz")))


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_oracle_matches_golden(oracle, path):
    d = np.load(path)
    st, poses, chi2, _ = oracle.gn_optimize(d["poses0"], d["fixed"], d["edge_from"], d["edge_to"], d["meas"],
                                            d["info"], int(d["iters"]))
    assert st == 0
    np.testing.assert_allclose(chi2, d["chi2"], rtol=1e-6)
    assert np.abs(poses - d["poses"]).max() < 1e-6
    st, cov = oracle.marginals(d["poses"], d["fixed"], d["edge_from"], d["edge_to"], d["meas"], d["info"], d["query"])
    assert st == 0
    np.testing.assert_allclose(cov, d["cov"], rtol=1e-6, atol=1e-12)


def test_jacobians_vs_central_differences(oracle):
    rng = np.random.default_rng(1)
    for _ in range(20):
        xi = rng.normal(size=3) * [3, 3, 1]
        xj = rng.normal(size=3) * [3, 3, 1]
        z = rng.normal(size=3) * [1, 1, 0.5]
        e, Ji, Jj = oracle.edge_terms(xi, xj, z)
        h = 1e-6
        for J, which in ((Ji, 0), (Jj, 1)):
            for k in range(3):
                d = np.zeros(3)
                d[k] = h
                a = [xi.copy(), xj.copy()]
                b = [xi.copy(), xj.copy()]
                a[which] = a[which] + d          # oplus: translation added in the global frame, angle added
                b[which] = b[which] - d
                ep, _, _ = oracle.edge_terms(a[0], a[1], z)
                em, _, _ = oracle.edge_terms(b[0], b[1], z)
                np.testing.assert_allclose((ep - em) / (2 * h), J[:, k], atol=5e-8)


def test_two_vertices_one_edge_reaches_zero(oracle):
    poses = np.array([[0.0, 0, 0], [0.3, -0.2, 0.1]])
    fixed = np.array([1, 0], dtype=np.uint8)
    meas = np.array([[1.0, 0.5, 0.3]])
    info = np.array([[100.0, 0, 0, 100, 0, 1000]])
    st, p, chi2, _ = oracle.gn_optimize(poses, fixed, [0], [1], meas, info, 3)
    assert st == 0
    assert chi2[-1] < 1e-20
    np.testing.assert_allclose(p[1], [1.0, 0.5, 0.3], atol=1e-10)


def test_consistent_square_loop(oracle):
    truth = np.array([[0.0, 0, 0], [1, 0, np.pi / 2], [1, 1, np.pi], [0, 1, -np.pi / 2]])
    ef = np.array([0, 1, 2, 3], dtype=np.int32)
    et = np.array([1, 2, 3, 0], dtype=np.int32)
    meas = synth.se2_compose(synth.se2_inverse(truth[ef]), truth[et])
    info = np.tile([100.0, 0, 0, 100, 0, 1000], (4, 1))
    rng = np.random.default_rng(3)
    p0 = truth + rng.normal(scale=0.05, size=truth.shape)
    p0[0] = truth[0]
    st, p, chi2, _ = oracle.gn_optimize(p0, [1, 0, 0, 0], ef, et, meas, info, 8)
    assert st == 0 and chi2[-1] < 1e-18
    assert np.abs(p[:, :2] - truth[:, :2]).max() < 1e-9
    assert np.abs(synth.normalize_theta(p[:, 2] - truth[:, 2])).max() < 1e-9


def test_fixed_point_two_initial_guesses(oracle):
    g = synth.make_pose_graph(400, 1200, seed=21)
    a = (g["fixed"], g["edge_from"], g["edge_to"], g["meas"], g["info"])
    _, p1, c1, _ = oracle.gn_optimize(g["poses"], *a, 12)
    rng = np.random.default_rng(5)
    start2 = g["truth"] + rng.normal(scale=0.02, size=g["truth"].shape)
    start2[0] = g["truth"][0]
    _, p2, c2, _ = oracle.gn_optimize(start2, *a, 12)
    assert abs(c1[-1] - c2[-1]) / c1[-1] < 1e-9
    assert np.abs(p1 - p2).max() < 1e-6
    dof = 3 * len(g["edge_from"]) - 3 * (400 - 1)
    assert 0.8 * dof < c1[-1] < 1.2 * dof            # chi2 ~ dof on Gaussian noise


def test_singular_system_reports_failure(oracle):
    # no fixed vertex: gauge freedom makes H singular -> Cholesky must fail, poses untouched
    poses = np.array([[0.0, 0, 0], [1, 0, 0]])
    st, p, chi2, _ = oracle.gn_optimize(poses, [0, 0], [0], [1], [[1.0, 0, 0]], [[1.0, 0, 0, 1, 0, 1]], 2)
    assert st < 0
    np.testing.assert_array_equal(p, poses)


def test_initial_guess_chain(oracle):
    g = synth.make_pose_graph(50, 49, seed=2)          # pure odometry chain
    scrambled = g["poses"] + 5.0
    scrambled[0] = g["poses"][0]
    p = oracle.initial_guess(scrambled, g["fixed"], g["edge_from"], g["edge_to"], g["meas"])
    np.testing.assert_allclose(p, g["poses"], atol=1e-9)


def test_unscented_label_small_covariance_is_first_order(oracle):
    # tiny Sigma: information ~ (J Sigma J^T)^-1 with J = d e / d x_v
    xg = np.array([0.3, -0.2, 0.4])
    xv = np.array([2.0, 1.0, -0.7])
    A = np.array([[2.0, 0.3, 0.1], [0.3, 1.5, -0.2], [0.1, -0.2, 0.8]]) * 1e-8
    st, m, iu = oracle.label_edge(xg, xv, A)
    assert st == 0
    np.testing.assert_allclose(m, synth.se2_compose(synth.se2_inverse(xg), xv), atol=1e-12)
    _, _, Jj = oracle.edge_terms(xg, xv, m)
    want = np.linalg.inv(Jj @ A @ Jj.T)
    got = np.array([[iu[0], iu[1], iu[2]], [iu[1], iu[3], iu[4]], [iu[2], iu[4], iu[5]]])
    np.testing.assert_allclose(got, want, rtol=1e-4)


def test_condensed_chain_covariance_closed_form(oracle):
    # chain graph 0-1-2-3 with the gauge at 0: Sigma of vertex k = composed odometry covariance.
    # With zero rotation and measurement (1,0,0) the translation x-variance simply adds up.
    V = 4
    poses = np.array([[float(k), 0, 0] for k in range(V)])
    ef = np.arange(V - 1, dtype=np.int32)
    et = ef + 1
    meas = np.tile([1.0, 0, 0], (V - 1, 1))
    info = np.tile([100.0, 0, 0, 100, 0, 1000], (V - 1, 1))
    n, to, est, iu, cov = oracle.condense(poses, ef, et, meas, info, 0, [0, 1, 2, 3])
    assert n == 3 and list(to) == [1, 2, 3]
    for k in range(3):
        assert abs(cov[k][0, 0] - (k + 1) / 100.0) < 1e-12      # x variance accumulates
        assert abs(cov[k][2, 2] - (k + 1) / 1000.0) < 1e-12     # theta variance accumulates
        np.testing.assert_allclose(est[k], [k + 1.0, 0, 0], atol=1e-9)


def test_condense_matches_numpy_marginals(oracle):
    g = synth.make_pose_graph(200, 600, seed=9)
    _, p, _, _ = oracle.gn_optimize(g["poses"], g["fixed"], g["edge_from"], g["edge_to"], g["meas"], g["info"], 8)
    query = np.array([20, 60, 100, 140], dtype=np.int32)
    gauge = 60
    n, to, est, iu, cov = oracle.condense(p, g["edge_from"], g["edge_to"], g["meas"], g["info"], gauge, query)
    assert n == 3
    fixed = np.zeros(200, dtype=np.uint8)
    fixed[gauge] = 1
    p_init = oracle.initial_guess(p, fixed, g["edge_from"], g["edge_to"], g["meas"])
    want = R.marginals_dense(p_init, fixed, g["edge_from"], g["edge_to"], g["meas"], g["info"], to)
    np.testing.assert_allclose(cov, want, rtol=1e-6, atol=1e-14)


def test_hub_graph_oracle_vs_numpy(oracle):
    """The elimination-tree stress graph of the GPU tests (fronts with dozens of children): the C oracle and the
    independent numpy / SuperLU restatement agree on it."""
    g = synth.make_hub_graph(24, 20, 3)
    a = (g["poses"], g["fixed"], g["edge_from"], g["edge_to"], g["meas"], g["info"])
    st, p, chi, _ = oracle.gn_optimize(*a, 5)
    p2, chi2 = R.gn_optimize(*a, 5)
    assert st == 0
    np.testing.assert_allclose(chi, chi2, rtol=1e-6)
    np.testing.assert_allclose(chi[-1], chi2[-1], rtol=1e-9)
    assert np.abs(p[:, :2] - p2[:, :2]).max() < 1e-8


def test_hand_worked_gauss_newton_step(oracle):
    """tests/known_answers.py (G): H, b and dx of a three-vertex graph written out from SURVEY.md Appendix A's formulas."""
    import known_answers as K
    st, poses, chi2, _ = oracle.gn_optimize(K.GN_POSES, K.GN_FIXED, K.GN_FROM, K.GN_TO, K.GN_MEAS, K.GN_INFO, 1)
    assert st == 0
    assert abs(chi2[0] - K.GN_CHI2_BEFORE) < 1e-10
    assert np.abs(poses - K.GN_POSES_AFTER).max() < 1e-12
