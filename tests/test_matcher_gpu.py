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


def test_beam_counts_divisible_by_the_stamping_stride(ctx, oracle):
    """The rasteriser visits the reference beams with stride 67 modulo an odd count so that concurrent stamps rarely
    overlap; a count that is a multiple of 67 (1005 beams = 15 x 67, and 1004 -> 1005) must not lose beams: the
    permutation has to stay a bijection."""
    for nb in (1005, 1004, 469):
        sp = synth.make_scan_pairs(6, seed=300 + nb, n_beams=nb)
        m = _matcher(ctx, sp)
        got = m.closeScanMatching(sp["ranges_ref"], sp["ranges_qry"], sp["guess"])
        _assert_same(got, _oracle(oracle, sp, sp["ranges_ref"], sp["ranges_qry"], sp["guess"]))


def test_scattered_scans_take_the_overflow_paths(ctx, oracle):
    """Scans whose points do not lie on walls: (a) a reference scan that touches more tiles than the LDS pool holds
    (tiles spill to HBM, the pair takes the bounds-checked search), (b) a query scan with more subsampled points
    than one point list holds (half the wavefronts search with two lists each), (c) both at once.  Results must
    still be bit-identical to the oracle."""
    sp = synth.make_scan_pairs(4, seed=82)
    rng = np.random.default_rng(5)
    scattered = lambda: rng.uniform(1.0, 14.0, size=sp["n_beams"]).astype(np.float32)   # noqa: E731
    rr, rq, g = sp["ranges_ref"].copy(), sp["ranges_qry"].copy(), sp["guess"].copy()
    rr[0] = scattered()                        # (a)
    rq[1] = scattered()                        # (b)
    rr[2] = scattered(); rq[2] = rr[2] + rng.normal(scale=0.01, size=sp["n_beams"]).astype(np.float32)   # (c), matchable
    g[2] = 0.0
    m = _matcher(ctx, sp)
    got = m.closeScanMatching(rr, rq, g, maxScore=0.3)
    _assert_same(got, _oracle(oracle, sp, rr, rq, g, max_score=0.3))
    assert got[0][2] and np.abs(got[1][2]).max() < 0.05       # the scattered scan matches itself at the origin


def _room_pairs(sp, sizes_and_headings, seed=3):
    """Pairs of scans taken inside an empty rectangular room: (width, height, heading of the first pose)."""
    rng = np.random.default_rng(seed)
    ang0 = sp["angle_min"] + sp["angle_inc"] * np.arange(sp["n_beams"])
    rr, rq, g = [], [], []
    for (w, h, th) in sizes_and_headings:
        boxes = [(-w / 2, -h / 2, w / 2, h / 2)]
        p1 = np.array([0.3, -0.2, th])
        d = np.array([0.12, -0.08, 0.05])
        p2 = synth.se2_compose(p1, d)
        for pose, out in ((p1, rr), (p2, rq)):
            r = synth._raycast_boxes(pose[0], pose[1], pose[2] + ang0, boxes, sp["max_range"])
            out.append(np.clip(r + rng.normal(scale=0.01, size=r.shape), 0.05, None).astype(np.float32))
        g.append(synth.se2_compose(d, np.array([0.03, -0.02, 0.01])))
    return np.stack(rr), np.stack(rq), np.stack(g)


def test_long_walls_borrow_the_point_lists_for_their_tiles(ctx, oracle):
    """A reference scan that sees 60 m of wall at an angle to the grid claims more tiles than the LDS pool's 1248: up to 1424 the
    tiles go on into the first half of the point lists and half the wavefronts search (same fast search, no HBM tiles); the
    results stay bit-identical to the oracle in the pruned and the exhaustive search, in a batch and alone."""
    sp = synth.make_scan_pairs(2, seed=82)
    rooms = [(21.0, 21.0, 0.78), (22.0, 22.0, 0.6), (24.0, 24.0, 0.3), (28.0, 28.0, 0.0), (12.0, 9.0, 0.4), (22.0, 21.0, 1.0)]   # 1260 .. 1350 tiles
    rr, rq, g = _room_pairs(sp, rooms)
    m = _matcher(ctx, sp)
    want = _oracle(oracle, sp, rr, rq, g)
    got = m.closeScanMatching(rr, rq, g)
    st = m.last_stats()
    _assert_same(got, want)
    assert st["slow_pairs"] == 0 and st["borrowed_pool_pairs"] >= 4, st
    f2, x2, s2, nres = m.closeScanMatching(rr, rq, g, want_nresults=True)
    _assert_same((f2, x2, s2), want)
    assert m.last_stats()["slow_pairs"] == 0 and m.last_stats()["borrowed_pool_pairs"] == st["borrowed_pool_pairs"]
    assert got[0].all()
    for i in range(len(rooms)):                                   # single calls: 16 workgroups share the pair
        _assert_same(m.closeScanMatching(rr[i], rq[i], g[i]), tuple(np.asarray(a)[i:i + 1] for a in want))


def test_a_full_batch_sorts_its_pairs_by_search_path_and_answers_them_all(ctx, oracle):
    """More pairs than compute units: the batch runs in the kernel instance built for the common shape, which hands what it cannot take
    to the launches behind it -- pairs whose tiles borrow the point lists stay with it, a reference scan with tiles beyond even that
    and a query scan with more points than one list holds go onto the slow list (16 workgroups each).  Every pair, whatever its
    path, is bit-identical to the oracle, pruned and exhaustive, and the path counters say where the pairs went."""
    sp = synth.make_scan_pairs(200, seed=83)
    rooms = [(21.0, 21.0, 0.78), (22.0, 22.0, 0.6), (24.0, 24.0, 0.3), (28.0, 28.0, 0.0), (22.0, 21.0, 1.0)]
    r1, q1, g1 = _room_pairs(sp, rooms, seed=4)
    rng = np.random.default_rng(6)
    scattered = lambda: rng.uniform(1.0, 14.0, size=sp["n_beams"]).astype(np.float32)   # noqa: E731
    rr = np.concatenate([sp["ranges_ref"], r1, sp["ranges_ref"][:95]])
    rq = np.concatenate([sp["ranges_qry"], q1, sp["ranges_qry"][:95]])
    g = np.concatenate([sp["guess"], g1, sp["guess"][:95]])
    rr[7] = scattered()                                           # tiles beyond the borrowed lists
    rq[11] = scattered()                                          # more subsampled points than one list holds
    rr[13] = scattered(); rq[13] = rr[13] + rng.normal(scale=0.01, size=sp["n_beams"]).astype(np.float32); g[13] = 0.0   # both
    rr[17] = 29.99; rq[17] = 29.99                                # far points: stamps hang over the grid border
    assert len(rr) == 300
    m = _matcher(ctx, sp)
    want = _oracle(oracle, sp, rr, rq, g, max_score=0.3)
    for exhaustive in (False, True):
        out = m.closeScanMatching(rr, rq, g, maxScore=0.3, want_nresults=exhaustive)
        st = m.last_stats()
        _assert_same(out[:3], want)
        assert st["pairs"] == 300 and st["borrowed_pool_pairs"] >= 4, st
        assert st["slow_pairs"] == 2, st                          # pairs 7 and 13
        assert st["redo_by_cause"]["grid"] >= 2 and st["redo_by_cause"]["window_or_points"] >= 1, st
    assert want[2][13] and np.abs(want[0][13]).max() < 0.05


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


def test_pruned_search_returns_the_exhaustive_winner(ctx, oracle):
    """A caller that asks for the number of populated result bins gets the exhaustive search, everybody else the pruned
    one (partial sums as lower bounds, rows of the window that cannot win dropped): same transform, score and flag on
    ordinary pairs, on the edge cases, with loose and tight acceptance thresholds, and for single calls (16 workgroups
    share a pair, each with its own bound)."""
    sp = synth.make_scan_pairs(40, seed=91)
    rr, rq, g = sp["ranges_ref"].copy(), sp["ranges_qry"].copy(), sp["guess"].copy()
    rr[0] = 100.0; rq[1] = 100.0; rr[2, ::2] = 100.0; rq[3, 100:900] = 0.0
    g[4] += [0.31, -0.29, 0.21]; g[5] = [14.9, -14.95, 3.1]; rr[6] = 29.99; rq[6] = 29.99
    m = _matcher(ctx, sp)
    for max_score in (0.02, 0.15, 5.0):
        f1, x1, s1 = m.closeScanMatching(rr, rq, g, maxScore=max_score)
        f2, x2, s2, nres = m.closeScanMatching(rr, rq, g, maxScore=max_score, want_nresults=True)
        assert np.array_equal(f1, f2) and np.array_equal(x1, x2) and np.array_equal(s1, s2)
        assert np.array_equal(nres > 0, f2)
        _assert_same((f1, x1, s1), _oracle(oracle, sp, rr, rq, g, max_score=max_score))
    for i in (7, 8, 9):
        a = m.closeScanMatching(rr[i], rq[i], g[i])
        b = m.closeScanMatching(rr[i], rq[i], g[i], want_nresults=True)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0])


def test_fallback_rasteriser_and_exhaustive_search_match_the_golden_fixture():
    """The library read with CGMR_MATCH_EDT=0 CGMR_MATCH_PRUNE=0 (compare-and-swap stamping instead of the distance
    transform, every candidate evaluated): the paths grids with overflow tiles, unusual kernels and bin counts fall back
    to must stay bit-identical too.  The switches are read once per process, hence the child."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, numpy as np\n"
        "from cg_mrslam_amd import Context\n"
        "from cg_mrslam_amd.matcher import ScanMatcher\n"
        "d = np.load(os.path.join('tests', 'golden', 'match_close12.npz'))\n"
        "m = ScanMatcher(Context(0), d['ranges_ref'].shape[1], float(d['angle_min']), float(d['angle_inc']), float(d['max_range']))\n"
        "f, x, s = m.closeScanMatching(d['ranges_ref'], d['ranges_qry'], d['guess'])\n"
        "assert np.array_equal(f, d['found'].astype(bool)) and np.array_equal(x, d['xyt']) and np.array_equal(s, d['score'])\n"
        "print('fallback ok')\n")
    env = dict(os.environ, CGMR_MATCH_EDT="0", CGMR_MATCH_PRUNE="0", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fallback ok" in r.stdout, r.stderr[-2000:]


def test_rejects_bad_configuration(ctx):
    from cg_mrslam_amd import CgmrError
    sp = synth.make_scan_pairs(1, seed=82)
    m = _matcher(ctx, sp)
    m.initializeGrid((-100, -100), (100, 100), 0.025)      # 8000x8000 cells: beyond the tile directory
    with pytest.raises(CgmrError):
        m.closeScanMatching(sp["ranges_ref"], sp["ranges_qry"], sp["guess"])


# ----------------------------------------------------------------------- generic searches (LC / hierarchical)

def _lc(ctx, sp):
    from cg_mrslam_amd.matcher import LCScanMatcher
    return LCScanMatcher(ctx, sp["n_beams"], sp["angle_min"], sp["angle_inc"], sp["max_range"])


def test_greedy_multi_region_matches_oracle(ctx, oracle):
    """CharGrid::greedySearch with 1..9 regions (more than the 4 per-thread result maps), full sorted result list."""
    sp = synth.make_scan_pairs(2, seed=90)
    m = _lc(ctx, sp)
    for p in range(2):
        ref = m.cartesian(sp["ranges_ref"][p])
        q = m.subsample(m.cartesian(sp["ranges_qry"][p]))
        assert np.array_equal(ref, oracle.cartesian(sp["ranges_ref"][p], sp["angle_min"], sp["angle_inc"], sp["max_range"]))
        assert np.array_equal(q, oracle.subsample(oracle.cartesian(sp["ranges_qry"][p], sp["angle_min"], sp["angle_inc"], sp["max_range"])))
        g = sp["guess"][p]
        base = np.array([-.5 + g[0], -1.5 + g[1], -.8 + g[2], .5 + g[0], 1.5 + g[1], .8 + g[2]])
        rng = np.random.default_rng(p)
        for nreg in (1, 3, 9):
            regs = np.array([base + np.tile(rng.uniform(-0.4, 0.4, 3), 2) for _ in range(nreg)], dtype=np.float32)
            regs[0] = base
            got = m.greedySearch(ref, q, regs, 0.025, 0.3, 0.5, 0.5, 0.2)
            n, want = oracle.greedy_search((-35, -35), (35, 35), 0.1, 0.1, 0.5, ref, q, regs, 0.1, 0.025, 0.3, 0.5, 0.5, 0.2)
            assert len(got) == n > 0
            assert np.array_equal(got, want)
    # degenerate inputs: empty region (upper < lower), no query points
    empty = np.array([[0.5, 0.5, 0.1, 0.2, 0.2, 0.0]], dtype=np.float32)
    assert len(m.greedySearch(ref, q, empty, 0.025, 0.3, 0.5, 0.5, 0.2)) == 0
    assert len(m.greedySearch(ref, np.zeros((0, 2)), regs, 0.025, 0.3, 0.5, 0.5, 0.2)) == 0


def test_greedy_with_more_query_points_than_one_list_holds(ctx, oracle):
    """A current set of several scans: ~2500 subsampled query points, i.e. three chunks of the kept-point list per (region,
    angle) item.  The wavefront pairs of the 512-thread search kernel share a list slot, so a chunk must not be rebuilt
    while the pair's other wavefront still reads the previous one: full result list against the oracle."""
    sp = synth.make_scan_pairs(6, seed=93)
    m = _lc(ctx, sp)
    ref = m.cartesian(sp["ranges_ref"][0])
    rng = np.random.default_rng(2)
    parts = []
    for p in range(6):                                            # six scans of different rooms, shifted: few points share a 0.1 m cell
        pts = m.cartesian(sp["ranges_qry"][p]) + rng.uniform(-0.05, 0.05, size=2)
        parts.append(pts)
    q = m.subsample(np.concatenate(parts))
    assert len(q) > 2200
    g = sp["guess"][0]
    regs = np.array([[-.5 + g[0], -1.0 + g[1], -.4 + g[2], .5 + g[0], 1.0 + g[1], .4 + g[2]]], dtype=np.float32)
    got = m.greedySearch(ref, q, regs, 0.025, 0.6, 0.5, 0.5, 0.2)
    n, want = oracle.greedy_search((-35, -35), (35, 35), 0.1, 0.1, 0.5, ref, q, regs, 0.1, 0.025, 0.6, 0.5, 0.5, 0.2)
    assert len(got) == n > 0 and np.array_equal(got, want)


def test_hierarchical_and_global_matching(ctx, oracle):
    sp = synth.make_scan_pairs(2, seed=91)
    m = _lc(ctx, sp)
    region = np.array([[-10, -5, np.float32(-np.pi), 10, 5, np.float32(np.pi)]], dtype=np.float32)
    for p in range(2):
        ref = m.cartesian(sp["ranges_ref"][p])
        q = m.subsample(m.cartesian(sp["ranges_qry"][p]))
        got = m.hierarchicalSearch(ref, q, region, 0.025, 0.2, 0.5, 0.5, 0.2, 4)
        n, want = oracle.hierarchical_search((-35, -35), (35, 35), 0.1, 0.1, 0.5, ref, q, region, 0.025, 0.2, 0.5, 0.5, 0.2, 4)
        assert len(got) == n and np.array_equal(got, want)
        assert len(m.hierarchicalSearch(ref, q, region, 0.025, 0.2, 0.5, 0.5, 0.2, 1)) == 0      # reference quirk
        ok, trel = m.globalMatching([(sp["ranges_ref"][p], np.zeros(3))], 0, [(sp["ranges_qry"][p], sp["guess"][p])], 0, 0.2)
        assert ok
        err = np.abs(trel - sp["true_rel"][p])
        assert err[0] < 0.15 and err[1] < 0.15 and err[2] < 0.06          # LC grid is 0.1 m, theta step 0.025


def test_level_loop_on_the_device_and_level_by_level(ctx, oracle):
    """hierarchicalSearch's levels run back to back on the device (k_hier_next makes a level's regions from the results of the
    one before); a search with more than 256 results on a level outgrows the device tables and is served level by level with
    the tables made on the host.  Both against the oracle, and the device loop against the host loop of another process."""
    import subprocess, sys, os, json
    sp = synth.make_scan_pairs(2, seed=77)
    m = _lc(ctx, sp)
    ref = m.cartesian(sp["ranges_ref"][0])
    q = m.subsample(m.cartesian(sp["ranges_qry"][0]))
    region = np.array([[-6, -4, np.float32(-np.pi), 6, 4, np.float32(np.pi)]], dtype=np.float32)
    counts = {}
    for levels, max_score in ((2, 0.2), (3, 0.2), (4, 0.25), (3, 0.45)):
        got = m.hierarchicalSearch(ref, q, region, 0.025, max_score, 0.5, 0.5, 0.2, levels)
        n, want = oracle.hierarchical_search((-35, -35), (35, 35), 0.1, 0.1, 0.5, ref, q, region, 0.025, max_score, 0.5, 0.5, 0.2, levels)
        assert len(got) == n and np.array_equal(got, want), (levels, max_score, len(got), n)
        counts[(levels, max_score)] = n
    assert 256 < counts[(3, 0.45)] <= 4096                        # (cannot have stayed in the device tables; the oracle wrapper holds 4096 rows)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, json\nfrom cg_mrslam_amd import Context, synth\nfrom tests.test_matcher_gpu import _lc\n"
            "ctx = Context(0)\nsp = synth.make_scan_pairs(2, seed=77)\nm = _lc(ctx, sp)\n"
            "ref = m.cartesian(sp['ranges_ref'][0]); q = m.subsample(m.cartesian(sp['ranges_qry'][0]))\n"
            "region = np.array([[-6, -4, np.float32(-np.pi), 6, 4, np.float32(np.pi)]], dtype=np.float32)\n"
            "print(json.dumps(np.asarray(m.hierarchicalSearch(ref, q, region, 0.025, 0.25, 0.5, 0.5, 0.2, 4)).tolist()))\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, PYTHONPATH=root, CGMR_HIER_HOST="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    host = np.array(json.loads(r.stdout.strip().splitlines()[-1]))
    dev = np.asarray(m.hierarchicalSearch(ref, q, region, 0.025, 0.25, 0.5, 0.5, 0.2, 4))
    assert host.shape == dev.shape and np.array_equal(host.reshape(dev.shape), dev)


def test_scan_matching_lc_multi_scan_reference_set(ctx, oracle):
    """scanMatchingLC (scan_matcher.cpp:201-294) with a 3-scan reference set: the host-side region/merge logic
    around the GPU search is checked against the same flow driven by the oracle's greedy search."""
    ang = synth.LASER_ANGLE_MIN + synth.LASER_ANGLE_INC * np.arange(1081)
    boxes = [(-6.0, -4.5, 6.0, 4.5), (1.0, 1.0, 2.2, 2.0)]
    poses = [np.array([0.0, 0.0, 0.1]), np.array([0.5, 0.1, 0.2]), np.array([-0.4, 0.3, -0.1])]
    scans = [(synth._raycast_boxes(p[0], p[1], p[2] + ang, boxes, 30.0).astype(np.float32), p) for p in poses]
    true_cur = np.array([0.8, -0.6, 0.35])
    cur = [(synth._raycast_boxes(true_cur[0], true_cur[1], true_cur[2] + ang, boxes, 30.0).astype(np.float32),
            true_cur + [0.2, -0.3, 0.1])]                               # odometry estimate of the current vertex
    sp = dict(n_beams=1081, angle_min=synth.LASER_ANGLE_MIN, angle_inc=synth.LASER_ANGLE_INC, max_range=30.0)
    m = _lc(ctx, sp)
    res = m.scanMatchingLC(scans, 0, cur, 0, 0.15)
    assert 1 <= len(res) <= 2
    rel_true = synth.se2_compose(synth.se2_inverse(poses[0]), true_cur)
    best = min(res, key=lambda r: np.abs(r[:2] - rel_true[:2]).sum())
    assert np.abs(best[:2] - rel_true[:2]).max() < 0.15 and abs(synth.normalize_theta(best[2] - rel_true[2])) < 0.06
    # same flow with the oracle's search in place of the GPU's
    ref_pts = m.transformPointsFromVSet(scans, 0)
    qry = m.subsample(m.transformPointsFromVSet(cur, 0), 0.1)
    from cg_mrslam_amd.matcher import _se2_inv, _se2_mul, normalize_theta
    merged = {}
    for shift in (0.0, 1.0):
        regs = []
        for k, (_, pose) in enumerate(scans):
            rel = np.zeros(3) if k == 0 else _se2_mul(_se2_inv(poses[0]), pose)
            lo = np.array([-.5 + rel[0], -1.5 + rel[1], -0.8 + rel[2]], dtype=np.float32)
            up = np.array([.5 + rel[0], 1.5 + rel[1], 0.8 + rel[2]], dtype=np.float32)
            if shift:
                lo[2] += np.float32(np.pi); up[2] += np.float32(np.pi)
            regs.append(np.concatenate([lo, up]))
        n, r = oracle.greedy_search((-35, -35), (35, 35), 0.1, 0.1, 0.5, ref_pts, qry, np.array(regs), 0.1, 0.025, 0.15, 0.5, 0.5, 0.2)
        if n:
            b = r[0].copy(); b[2] = normalize_theta(b[2])
            key = (int(b[0] / 0.5), int(b[1] / 0.5), int(b[2] / 0.2))
            if key not in merged or merged[key][3] > b[3]:
                merged[key] = b
    want = [merged[k][:3] for k in sorted(merged)]
    assert len(want) == len(res) and all(np.array_equal(a, b) for a, b in zip(res, want))


def test_verify_matching_core_matches_oracle(ctx, oracle):
    """cgmr_match_verify (numeric core of verifyMatching) against the oracle: same non-matched count, same score."""
    sp = synth.make_scan_pairs(3, seed=92)
    m = _lc(ctx, sp)
    import ctypes as C
    for p in range(3):
        pts2 = m.cartesian(sp["ranges_ref"][p])
        pts1 = m.applyTransfToScan(sp["true_rel"][p] + [0.4 * p, -0.2 * p, 0.05 * p], m.cartesian(sp["ranges_qry"][p]))
        for (lo, up) in (((-0.3, -0.3), (0.3, 0.3)), ((1.0, -2.0), (1.6, -1.4)), ((34.8, 34.8), (35.4, 35.4))):
            lo32, up32 = np.array(lo, dtype=np.float32), np.array(up, dtype=np.float32)
            score, nnm = C.c_double(), C.c_int()
            rc = ctx.lib.cgmr_match_verify(ctx.h, C.byref(m.cfg), C.c_int(len(pts2)), C.c_void_p(pts2.ctypes.data), C.c_int(len(pts1)),
                                           C.c_void_p(np.ascontiguousarray(pts1).ctypes.data), C.c_double(0.3),
                                           C.c_void_p(lo32.ctypes.data), C.c_void_p(up32.ctypes.data), C.byref(score), C.byref(nnm))
            assert rc == 0
            n_o, s_o = oracle.verify((-35, -35), (35, 35), 0.1, 0.1, 0.5, pts2, pts1, lo32, up32)
            assert nnm.value == n_o
            assert score.value == s_o or (np.isnan(score.value) and np.isnan(s_o))
    # well aligned scans explain each other: (almost) nothing unexplained near the origin -> the window stays at fill
    ok, sc = m.verifyMatching([(sp["ranges_qry"][0], sp["true_rel"][0])], 0, [(sp["ranges_ref"][0], np.zeros(3))], 0,
                              synth.se2_inverse(sp["true_rel"][0]))
    assert sc > 40.0 and not ok      # reference semantics: a *low* mean (unexplained points nearby) passes the <= 40 test


def test_close_matching_with_multi_scan_reference_set(ctx, oracle):
    """closeScanMatching as the reference calls it (up to 6 reference scans, graph_slam.cpp:230-241) through the generic
    search, against the oracle's greedy search on the same points, and against the batched single-scan kernel."""
    ang = synth.LASER_ANGLE_MIN + synth.LASER_ANGLE_INC * np.arange(1081)
    boxes = [(-5.0, -4.0, 5.0, 4.0)]
    poses = [np.array([0.1 * k, -0.05 * k, 0.03 * k]) for k in range(4)]
    scans = [(synth._raycast_boxes(p[0], p[1], p[2] + ang, boxes, 30.0).astype(np.float32), p) for p in poses]
    cur_true = np.array([0.55, -0.2, 0.18])
    cur_r = synth._raycast_boxes(cur_true[0], cur_true[1], cur_true[2] + ang, boxes, 30.0).astype(np.float32)
    from cg_mrslam_amd.matcher import ScanMatcher
    m = ScanMatcher(ctx, 1081, synth.LASER_ANGLE_MIN, synth.LASER_ANGLE_INC, 30.0)
    found, trel = m.closeScanMatchingVSet(scans, 3, cur_r, cur_true + [0.05, 0.04, -0.02], 0.15)
    assert found
    rel_true = synth.se2_compose(synth.se2_inverse(poses[3]), cur_true)
    assert np.abs(trel[:2] - rel_true[:2]).max() < 0.03 and abs(trel[2] - rel_true[2]) < 0.0126
    # oracle on the same points / window
    ref_pts = m.transformPointsFromVSet(scans, 3)
    qry = m.subsample(m.cartesian(cur_r), 0.1)
    from cg_mrslam_amd.matcher import _se2_inv, _se2_mul
    g = _se2_mul(_se2_inv(poses[3]), cur_true + [0.05, 0.04, -0.02])
    region = np.array([[-.3 + g[0], -.3 + g[1], -0.2 + g[2], .3 + g[0], .3 + g[1], 0.2 + g[2]]], dtype=np.float32)
    n, want = oracle.greedy_search((-15, -15), (15, 15), 0.025, 0.025, 0.2, ref_pts, qry, region, 0.025, 0.0125 * .5, 0.15, 0.5, 0.5, 0.2)
    assert n > 0 and np.array_equal(trel, want[0, :3])
    # single-scan reference set: identical to the batched kernel
    f1, t1 = m.closeScanMatchingVSet([scans[3]], 0, cur_r, cur_true + [0.05, 0.04, -0.02], 0.15)
    fb, tb, sb = m.closeScanMatching(scans[3][0], cur_r, g)
    assert f1 and fb[0] and np.array_equal(t1, tb[0])


def test_c_abi_scan_matcher_members_match_python_restatement_on_oracle(ctx, oracle):
    """Every ScanMatcher member function of the C ABI (csrc/matcher_api.cpp: transformPointsFromVSet,
    closeScanMatching with a 6-scan set, scanMatchingLC, globalMatching, verifyMatching) against the plain-Python
    restatement of the same bookkeeping (tests/ref_scan_matcher.py) running on the CPU oracle: bit-identical."""
    import oracle_backend as ob
    from cg_mrslam_amd.matcher import LCScanMatcher, ScanMatcher
    ang = synth.LASER_ANGLE_MIN + synth.LASER_ANGLE_INC * np.arange(1081)
    boxes = [(-6.0, -4.5, 6.0, 4.5), (1.0, 1.0, 2.2, 2.0), (-3.0, -2.5, -2.2, -1.0)]
    poses = [np.array([0.12 * k, -0.06 * k + 0.01 * k * k, 0.04 * k]) for k in range(6)]
    scans = [(synth._raycast_boxes(p[0], p[1], p[2] + ang, boxes, 30.0).astype(np.float32), p) for p in poses]
    cur_true = np.array([0.85, -0.25, 0.22])
    cur_r = synth._raycast_boxes(cur_true[0], cur_true[1], cur_true[2] + ang, boxes, 30.0).astype(np.float32)
    cur_est = cur_true + [0.06, -0.05, 0.03]
    la = (1081, synth.LASER_ANGLE_MIN, synth.LASER_ANGLE_INC, 30.0)
    lp = (0.1, -0.02, 0.03)                                    # a laser that is not at the robot centre
    mc, mo = ScanMatcher(ctx, *la, laser_pose=lp), ob.OracleMatcher(la[1], la[2], la[3], (-15.0, -15.0), (15.0, 15.0), 0.025, 0.2, lp)
    lc, lo = LCScanMatcher(ctx, *la, laser_pose=lp), ob.OracleMatcher(la[1], la[2], la[3], (-35.0, -35.0), (35.0, 35.0), 0.1, 0.5, lp)
    assert np.array_equal(mc.transformPointsFromVSet(scans, 5), mo.transformPointsFromVSet(scans, 5))
    # closeScanMatching: the reference's call shape (last vertex + 5 predecessors, graph_slam.cpp:230-244)
    f_c, t_c = mc.closeScanMatchingVSet(scans, 5, cur_r, cur_est, 0.15)
    f_o, t_o = mo.closeScanMatchingVSet(scans, 5, cur_r, cur_est, 0.15)
    assert f_c and f_o and np.array_equal(t_c, t_o)
    rel_true = synth.se2_compose(synth.se2_inverse(poses[5]), cur_true)
    assert np.abs(t_c[:2] - rel_true[:2]).max() < 0.03 and abs(t_c[2] - rel_true[2]) < 0.0126
    # scanMatchingLC, 3-scan reference set and 2-scan current set
    est2 = synth.se2_compose(cur_est, synth.se2_compose(synth.se2_inverse(cur_true), poses[5]))   # consistent with cur_est
    cur_set = [(cur_r, cur_est), (scans[5][0], est2)]
    r_c = lc.scanMatchingLC(scans[:3], 1, cur_set, 0, 0.3)
    r_o = lo.scanMatchingLC(scans[:3], 1, cur_set, 0, 0.3)
    assert len(r_c) == len(r_o) >= 1 and all(np.array_equal(a, b) for a, b in zip(r_c, r_o))
    # globalMatching
    g_c = lc.globalMatching(scans[:2], 0, [(cur_r, cur_est)], 0, 0.2)
    g_o = lo.globalMatching(scans[:2], 0, [(cur_r, cur_est)], 0, 0.2)
    assert g_c[0] == g_o[0] and g_c[0] and np.array_equal(g_c[1], g_o[1])
    # scanMatchingLChierarchical (scan_matcher.cpp:296-356; unused by the reference's loop): +-(2, 2, 1) around the relative estimate,
    # three hierarchy levels, the best result -- once with the estimate near the truth, once a metre and half a radian off
    for cs in (cur_set, [(cur_r, synth.se2_compose(cur_est, np.array([0.9, -0.7, 0.45]))), cur_set[1]]):
        h_c = lc.scanMatchingLChierarchical(scans[:3], 1, cs, 0, 0.3)
        h_o = lo.scanMatchingLChierarchical(scans[:3], 1, cs, 0, 0.3)
        assert h_c[0] == h_o[0] and len(h_c[1]) == len(h_o[1]) and all(np.array_equal(a, b) for a, b in zip(h_c[1], h_o[1]))
    assert lc.scanMatchingLChierarchical(scans[:3], 1, cur_set, 0, 0.3)[0]
    # verifyMatching with two-scan sets and a transform that is off by 0.4 m (unexplained points -> low window mean)
    for t12 in (synth.se2_compose(synth.se2_inverse(poses[0]), cur_true), np.array([0.4, 0.3, 0.1])):
        v_c = lc.verifyMatching(scans[:2], 0, cur_set, 0, t12)
        v_o = lo.verifyMatching(scans[:2], 0, cur_set, 0, t12)
        assert v_c[0] == v_o[0] and v_c[1] == v_o[1]


def _keyframe_sets(n_sets, n_scans=6, seed=77):
    """Reference sets the way GraphSLAM::addDataSM builds them: the last vertex + its predecessors along a trajectory
    (graph_slam.cpp:230-244), the next key frame as the current scan, odometry estimates as vertex poses."""
    tr = synth.make_trajectory(40 + 3 * n_sets, seed=seed, laps=0.35)
    ref, rel, cur, guess, sets = [], [], [], [], []
    for k in range(n_sets):
        last = 8 + 3 * k
        idx = [last - 2 * j for j in range(n_scans)][::-1]          # origin (the last vertex) is the final scan of the set
        scans = [(tr["scans"][i], tr["odom"][i]) for i in idx]
        org = tr["odom"][last]
        ref.append(np.stack([tr["scans"][i] for i in idx]))
        rel.append(np.stack([np.zeros(3) if i == last else synth.se2_compose(synth.se2_inverse(org), tr["odom"][i]) for i in idx]))
        cur.append(tr["scans"][last + 2])
        guess.append(synth.se2_compose(synth.se2_inverse(org), tr["odom"][last + 2]))
        sets.append((scans, n_scans - 1, tr["scans"][last + 2], tr["odom"][last + 2]))
    return np.stack(ref), np.stack(rel), np.stack(cur), np.stack(guess), sets, tr


def test_close_matching_batch_with_six_scan_reference_sets(ctx, oracle):
    """The batched close matcher with the reference's real call shape (6-scan reference sets): bit-identical to the
    per-set C entry point, to the generic search path, and to the Python restatement on the oracle; the sets stay on
    the LDS fast path."""
    import oracle_backend as ob
    from cg_mrslam_amd.matcher import ScanMatcher
    ref, rel, cur, guess, sets, tr = _keyframe_sets(24)
    la = (1081, synth.LASER_ANGLE_MIN, synth.LASER_ANGLE_INC, 30.0)
    m = ScanMatcher(ctx, *la)
    found, trel, score = m.closeScanMatchingVSetBatch(ref, rel, cur, guess, 0.15)
    st = m.last_stats()
    assert st["pairs"] == 24 and st["slow_pairs"] == 0
    assert found.sum() >= 22
    mo = ob.OracleMatcher(la[1], la[2], la[3], (-15.0, -15.0), (15.0, 15.0), 0.025, 0.2)
    for k in (0, 5, 11, 23):
        scans, oi, cr, cp = sets[k]
        f1, t1 = m.closeScanMatchingVSet(scans, oi, cr, cp, 0.15)                 # per-set C entry point
        fo, to = mo.closeScanMatchingVSet(scans, oi, cr, cp, 0.15)               # restatement on the CPU oracle
        assert f1 == fo == bool(found[k])
        if fo:
            assert np.array_equal(t1, to) and np.array_equal(trel[k], to)
        # the generic search path on the same points
        ref_pts = m.transformPointsFromVSet(scans, oi)
        qry = m.subsample(m.cartesian(cr), 0.1)
        g = guess[k]
        region = np.array([[-.3 + g[0], -.3 + g[1], -0.2 + g[2], .3 + g[0], .3 + g[1], 0.2 + g[2]]], dtype=np.float32)
        res = m.greedySearch(ref_pts, qry, region, 0.0125 * .5, 0.15, 0.5, 0.5, 0.2)
        assert (len(res) > 0) == fo and (not fo or (np.array_equal(res[0, :3], to) and res[0, 3] == score[k]))
    # the matched increments follow the true motion
    for k in range(24):
        if found[k]:
            last = 8 + 3 * k
            true_rel = synth.se2_compose(synth.se2_inverse(tr["truth"][last]), tr["truth"][last + 2])
            # the estimate is relative to the odometry frame of the set; compare the increment's length
            assert abs(np.hypot(*trel[k][:2]) - np.hypot(*true_rel[:2])) < 0.06
    # fewer scans than the set holds: padded with all-zero scans (no valid beam) == the shorter set
    pad = ref[:4].copy(); pad[:, :3] = 0.0
    f_p, t_p, _ = m.closeScanMatchingVSetBatch(pad, rel[:4], cur[:4], guess[:4], 0.15)
    f_s, t_s, _ = m.closeScanMatchingVSetBatch(ref[:4, 3:], rel[:4, 3:], cur[:4], guess[:4], 0.15)
    assert np.array_equal(f_p, f_s) and np.array_equal(t_p, t_s)


def test_batched_lc_global_verify_equal_single_calls(ctx, oracle):
    """a batch of loop-closure / global / verify jobs (one launch per search level for all of them)
    returns exactly what the single calls return -- which the tests above pin against the oracle."""
    import time
    from cg_mrslam_amd.matcher import LCScanMatcher
    ref, rel, cur, guess, sets, tr = _keyframe_sets(12, n_scans=3, seed=78)
    la = (1081, synth.LASER_ANGLE_MIN, synth.LASER_ANGLE_INC, 30.0)
    lc = LCScanMatcher(ctx, *la)
    jobs = []
    for k, (scans, oi, cr, cp) in enumerate(sets):
        off = np.array([0.3 * np.cos(k), 0.4 * np.sin(2 * k), 0.2 * np.sin(k)])     # a poor prior for the current vertex
        jobs.append((scans, oi, [(cr, cp + off)], 0))
    t0 = time.perf_counter()
    single_lc = [lc.scanMatchingLC(j[0], j[1], j[2], j[3], 0.3) for j in jobs]
    t1 = time.perf_counter()
    batch_lc = lc.scanMatchingLCBatch(jobs, 0.3)
    t2 = time.perf_counter()
    assert sum(len(r) for r in single_lc) >= 10
    for a, b in zip(single_lc, batch_lc):
        assert len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))
    single_g = [lc.globalMatching(j[0], j[1], j[2], j[3], 0.25) for j in jobs[:6]]
    t3 = time.perf_counter()
    batch_g = lc.globalMatchingBatch(jobs[:6], 0.25)
    t4 = time.perf_counter()
    assert sum(1 for f, _ in single_g if f) >= 4
    for (fa, ta), (fb, tb) in zip(single_g, batch_g):
        assert fa == fb and (not fa or np.array_equal(ta, tb))
    single_h = [lc.scanMatchingLChierarchical(j[0], j[1], j[2], j[3], 0.3) for j in jobs[:6]]
    batch_h = lc.scanMatchingLChierarchicalBatch(jobs[:6], 0.3)
    assert sum(1 for f, _ in single_h if f) >= 4
    for (fa, ta), (fb, tb) in zip(single_h, batch_h):
        assert fa == fb and len(ta) == len(tb) and all(np.array_equal(x, y) for x, y in zip(ta, tb))
    vjobs = [(j[0], j[1], j[2], j[3]) for j in jobs]
    t12 = np.array([r[0] if len(r) else np.array([0.4, 0.2, 0.1]) for r in single_lc])
    single_v = [lc.verifyMatching(j[0], j[1], j[2], j[3], t12[k]) for k, j in enumerate(vjobs)]
    batch_v = lc.verifyMatchingBatch(vjobs, t12)
    assert single_v == batch_v
    print(f"12 LC jobs: {1e3 * (t1 - t0):.1f} ms one by one, {1e3 * (t2 - t1):.1f} ms batched; "
          f"6 global jobs: {1e3 * (t3 - t2):.1f} ms one by one, {1e3 * (t4 - t3):.1f} ms batched")


def test_hand_derived_three_point_scan_on_gpu(ctx):
    """The hand-derived known answers of tests/known_answers.py (worked out from chargrid.cpp / gridmap.h, no
    implementation involved) through the C ABI."""
    import known_answers as K
    from cg_mrslam_amd.matcher import ScanMatcher
    m = ScanMatcher(ctx, 1081, synth.LASER_ANGLE_MIN, synth.LASER_ANGLE_INC, 30.0)
    res = m.greedySearch(K.REF, K.REF, K.REGION, 0.00625, 0.5, 0.5, 0.5, 0.2)
    assert len(res) == 1 and tuple(res[0]) == K.EXPECTED_SAME
    res = m.greedySearch(K.REF, K.REF + [0.025, 0.0], K.REGION, 0.00625, 0.5, 0.5, 0.5, 0.2)
    assert len(res) == 1 and tuple(res[0]) == K.EXPECTED_SHIFTED
    res = m.greedySearch(K.REF, K.REF, K.REGION, 0.00625, 0.5, 0.0125, 0.0125, 0.2)
    got = {(round((r[0] + 15) * 40), round((r[1] + 15) * 40)): r[3] for r in res}
    assert len(res) == 16
    for i in range(598, 602):
        for j in range(598, 602):
            assert got[(i, j)] == K.expected_score(i, j, K.QUERY_CELLS), (i, j)


def test_committed_4096_pair_subset_of_c3(ctx):
    """parity on a committed 4096-pair subset of the C3 workload (tests/golden/match_close4096.npz:
    the CPU oracle's results for synth.make_scan_pairs(4096, seed=4242), inputs pinned by their SHA-256): bit-exact."""
    import hashlib
    import os
    from cg_mrslam_amd.matcher import ScanMatcher
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "match_close4096.npz"))
    sp = synth.make_scan_pairs(int(G["n_pairs"]), seed=int(G["seed"]))
    h = hashlib.sha256()
    for k in ("ranges_ref", "ranges_qry", "guess"):
        h.update(np.ascontiguousarray(sp[k]).tobytes())
    assert h.hexdigest() == str(G["inputs_sha256"]), "the regenerated inputs differ from those the fixture was made from"
    m = ScanMatcher(ctx, sp["n_beams"], sp["angle_min"], sp["angle_inc"], sp["max_range"])
    found, xyt, score = m.closeScanMatching(sp["ranges_ref"], sp["ranges_qry"], sp["guess"])
    assert np.array_equal(found, G["found"].astype(bool)) and np.array_equal(xyt, G["xyt"]) and np.array_equal(score, G["score"])
    assert m.last_stats()["slow_pairs"] == 0


def test_hand_derived_subsample_hierarchy_verify_and_twin_regions_on_gpu(ctx):
    """The round-5 cases of tests/known_answers.py (worked out on paper from chargrid.cpp:61-122, 310-455 and
    scan_matcher.cpp:220-294) through the C ABI: cgmr_subsample, cgmr_match_hierarchical, cgmr_match_verify and
    cgmr_scan_matching_lc (whose twin regions hinge on `lower[2] += M_PI` being a double sum narrowed once)."""
    import ctypes as C
    import known_answers as K
    from cg_mrslam_amd.matcher import LCScanMatcher, ScanMatcher
    m = ScanMatcher(ctx, 1081, synth.LASER_ANGLE_MIN, synth.LASER_ANGLE_INC, 30.0)
    # (S)
    assert np.array_equal(m.subsample(K.SUBSAMPLE_IN, 0.1), K.SUBSAMPLE_OUT)
    # (H)
    a = K.HIER_ARGS
    res = m.hierarchicalSearch(K.HIER_REF, K.HIER_REF, K.HIER_REGION, a["theta_res"], a["max_score"], a["dx"], a["dy"], a["dth"], a["n_levels"])
    assert [tuple(r) for r in res] == K.HIER_EXPECTED
    # (V)
    lc = LCScanMatcher(ctx, K.LC_N_BEAMS, K.LC_ANGLE_MIN, K.LC_ANGLE_INC, K.LC_MAX_RANGE)
    score, nnm = C.c_double(), C.c_int()
    p2, p1 = np.ascontiguousarray(K.VERIFY_PTS2), np.ascontiguousarray(K.VERIFY_PTS1)
    rc = ctx.lib.cgmr_match_verify(ctx.h, C.byref(lc.cfg), C.c_int(len(p2)), C.c_void_p(p2.ctypes.data), C.c_int(len(p1)),
                                   C.c_void_p(p1.ctypes.data), C.c_double(0.3), C.c_void_p(K.VERIFY_LOWER.ctypes.data),
                                   C.c_void_p(K.VERIFY_UPPER.ctypes.data), C.byref(score), C.byref(nnm))
    assert rc == 0 and nnm.value == K.VERIFY_NONMATCHED and score.value == K.VERIFY_SCORE
    # (L): the same scan as reference and current set, at unrelated poses (only the sets' own frames enter)
    assert np.array_equal(lc.cartesian(K.LC_RANGES), K.LC_POINTS)
    res = lc.scanMatchingLC([(K.LC_RANGES, np.array([3.0, -2.0, 0.7]))], 0, [(K.LC_RANGES, np.array([-1.0, 4.0, -2.0]))], 0, K.LC_MAX_SCORE)
    assert [tuple(r) for r in res] == K.LC_EXPECTED
