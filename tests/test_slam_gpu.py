"""The key-frame loop end to end on the GPU, compared with the same driver running on the
oracle backend: same key frames, same edges (kind, end points), scan-match measurements bit-identical while the
two pose trajectories stay within GN tolerance of each other, loop closures found and accepted."""
import time

import numpy as np
import pytest

from cg_mrslam_amd import synth
from cg_mrslam_amd.matcher import LCScanMatcher, ScanMatcher
from cg_mrslam_amd.slam import GraphSLAMDriver, run_srslam

import oracle_backend as OB

pytestmark = pytest.mark.gpu


def _run(ctx, tr, gpu, **kw):
    la = (tr["n_beams"], tr["angle_min"], tr["angle_inc"], tr["max_range"])
    if gpu:
        slam = GraphSLAMDriver(ctx, ScanMatcher(ctx, *la), LCScanMatcher(ctx, *la), **kw)
    else:
        slam = GraphSLAMDriver(OB.OracleContext(), OB.close_matcher(la), OB.lc_matcher(la), **kw)
    run_srslam(slam, tr["odom"], tr["scans"], linearUpdate=0.5)
    return slam


def test_short_run_matches_oracle_backend(ctx, oracle):
    tr = synth.make_trajectory(120, laps=0.3)
    a = _run(ctx, tr, True)
    b = _run(ctx, tr, False)
    assert a.g.n_vertices == b.g.n_vertices and a.edge_kind == b.edge_kind
    np.testing.assert_array_equal(a.g.edge_from, b.g.edge_from)
    np.testing.assert_array_equal(a.g.edge_to, b.g.edge_to)
    np.testing.assert_array_equal(a.g.meas, b.g.meas)            # matcher output: bit-identical
    assert np.abs(a.g.poses - b.g.poses).max() < 1e-6


def test_loop_closure_run_matches_oracle_backend(ctx, oracle):
    """One lap and a bit of a corridor loop: the front end proposes loop closures when the start is revisited, the
    checker accepts a consistent set, and the GPU run reproduces the oracle-backed run."""
    tr = synth.make_trajectory(400, laps=1.12)
    kw = dict(windowLoopClosure=5, minInliers=4)
    t0 = time.time()
    a = _run(ctx, tr, True, **kw)
    t_gpu = time.time() - t0
    b = _run(ctx, tr, False, **kw)
    assert a.edge_kind.count("lc") > 0
    assert a.g.n_vertices == b.g.n_vertices and a.edge_kind == b.edge_kind
    np.testing.assert_array_equal(a.g.edge_from, b.g.edge_from)
    np.testing.assert_array_equal(a.g.edge_to, b.g.edge_to)
    np.testing.assert_array_equal(a.g.meas, b.g.meas)
    assert np.abs(a.g.poses - b.g.poses).max() < 1e-6
    assert [l for l in a.log if l[0] != "lcc"] == [l for l in b.log if l[0] != "lcc"]
    # closing the loop pulls the trajectory back onto the true path
    tp = tr["truth"]
    err = max(np.min(np.hypot(tp[:, 0] - p[0], tp[:, 1] - p[1])) for p in a.g.poses)
    assert err < 0.2
    print(f"GPU run: {a.g.n_vertices} key frames, {a.edge_kind.count('lc')} loop closures, {t_gpu:.1f} s")
