"""GPU parity tests of the scan matcher: the HIP kernel (through the C ABI) must be BIT-IDENTICAL to the CPU
oracle: byte grid, integer sums, float scores, double poses."""
import os

import numpy as np
import pytest

from cg_mrslam_amd import synth
from cg_mrslam_amd.matcher import ScanMatcher

pytestmark = pytest.mark.gpu


def _matcher(ctx, sp, **kw):
    return ScanMatcher(ctx, sp["n_beams"], sp["angle_min"], sp["angle_inc"], sp["max_range"], **kw)


def _oracle(oracle, sp, rr, rq, guess, laser_pose=(0, 0, 0), max_score=0.15):
    return oracle.close_scan_match_batch(rr, rq, sp["angle_min"], sp["angle_inc"], sp["max_range"], laser_pose, guess,
                                         max_score=max_score)


def _assert_same(got, want):
    found, xyt, score = got
    xo, so, fo = want
    assert np.array_equal(found, fo.astype(bool))
    assert np.array_equal(xyt, xo)
    assert np.array_equal(score, so)


def test_golden_fixture(ctx):
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "match_close12.npz"))
    m = ScanMatcher(ctx, d["ranges_ref"].shape[1], float(d["angle_min"]), float(d["angle_inc"]), float(d["max_range"]))
    found, xyt, score = m.closeScanMatching(d["ranges_ref"], d["ranges_qry"], d["guess"])
    assert np.array_equal(found, d["found"].astype(bool))
    assert np.array_equal(xyt, d["xyt"]) and np.array_equal(score, d["score"])


def test_parity_random_pairs(ctx, oracle):
    sp = synth.make_scan_pairs(48, seed=77)
    m = _matcher(ctx, sp)
    got = m.closeScanMatching(sp["ranges_ref"], sp["ranges_qry"], sp["guess"])
    _assert_same(got, _oracle(oracle, sp, sp["ranges_ref"], sp["ranges_qry"], sp["guess"]))
    assert got[0].all()


def test_parity_edge_cases(ctx, oracle):
    sp = synth.make_scan_pairs(8, seed=78)
    rr, rq, g = sp["ranges_ref"].copy(), sp["ranges_qry"].copy(), sp["guess"].copy()
    rr[0] = 100.0                       # reference scan entirely out of range: empty grid, nothing below maxScore
    rq[1] = 100.0                       # query scan entirely out of range: k = 0 -> score = maxScore + 1
    rr[2, ::2] = 100.0                  # ragged: every other beam invalid
    rq[3, 100:900] = 0.0                # r > min_range fails for zero ranges
    g[4] += [0.31, -0.29, 0.21]         # guess off by more than the window: usually no good match
    g[5] = [14.9, -14.95, 3.1]          # window partly outside the grid, angle window across pi
    rr[6] = 29.99                       # a circle of far points: stamps hang over the grid border
    rq[6] = 29.99
    m = _matcher(ctx, sp)
    got = m.closeScanMatching(rr, rq, g)
    _assert_same(got, _oracle(oracle, sp, rr, rq, g))
    assert not got[0][0] and not got[0][1]


def test_parity_scores_and_laser_pose(ctx, oracle):
    sp = synth.make_scan_pairs(6, seed=79)
    for max_score in (0.02, 0.5, 5.0):                     # 5.0 accepts every candidate: every bin is populated
        m = _matcher(ctx, sp)
        got = m.closeScanMatching(sp["ranges_ref"], sp["ranges_qry"], sp["guess"], maxScore=max_score)
        _assert_same(got, _oracle(oracle, sp, sp["ranges_ref"], sp["ranges_qry"], sp["guess"], max_score=max_score))
    lp = (0.12, -0.05, 0.3)                               # laser mounted off-centre and rotated
    m = _matcher(ctx, sp, laser_pose=lp)
    got = m.closeScanMatching(sp["ranges_ref"], sp["ranges_qry"], sp["guess"])
    _assert_same(got, _oracle(oracle, sp, sp["ranges_ref"], sp["ranges_qry"], sp["guess"], laser_pose=lp))


def test_small_scans_and_single_pair(ctx, oracle):
    sp = synth.make_scan_pairs(3, seed=80, n_beams=181)
    sp["angle_inc"] = synth.LASER_ANGLE_INC
    m = _matcher(ctx, sp)
    got = m.closeScanMatching(sp["ranges_ref"], sp["ranges_qry"], sp["guess"])
    _assert_same(got, _oracle(oracle, sp, sp["ranges_ref"], sp["ranges_qry"], sp["guess"]))
    got1 = m.closeScanMatching(sp["ranges_ref"][0], sp["ranges_qry"][0], sp["guess"][0])
    assert np.array_equal(got1[1][0], got[1][0])


def test_batch_recovers_truth_and_is_order_independent(ctx):
    """Size-independent properties on a larger batch: the match recovers the true motion to within a cell /
    angle step for the vast majority of pairs, and a pair's result does not depend on its position in the batch."""
    sp = synth.make_scan_pairs(600, seed=81)
    m = _matcher(ctx, sp)
    found, xyt, score = m.closeScanMatching(sp["ranges_ref"], sp["ranges_qry"], sp["guess"])
    err = np.abs(xyt - sp["true_rel"])
    good = (err[:, 0] < 0.04) & (err[:, 1] < 0.04) & (err[:, 2] < 0.013) & found
    assert good.mean() > 0.9
    perm = np.random.default_rng(1).permutation(600)
    f2, x2, s2 = m.closeScanMatching(sp["ranges_ref"][perm], sp["ranges_qry"][perm], sp["guess"][perm])
    assert np.array_equal(x2, xyt[perm]) and np.array_equal(s2, score[perm]) and np.array_equal(f2, found[perm])


def test_rejects_bad_configuration(ctx):
    from cg_mrslam_amd import CgmrError
    sp = synth.make_scan_pairs(1, seed=82)
    m = _matcher(ctx, sp)
    m.initializeGrid((-100, -100), (100, 100), 0.025)      # 8000x8000 cells: beyond the tile directory
    with pytest.raises(CgmrError):
        m.closeScanMatching(sp["ranges_ref"], sp["ranges_qry"], sp["guess"])
