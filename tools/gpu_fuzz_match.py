"""Randomised GPU-vs-oracle check of the close matcher over kernel / grid / laser configurations (argv: configs [seed0]).
Every configuration: 6 pairs, with and without the bin count (exhaustive and pruned search), bit-identical to the oracle."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth, Context
from cg_mrslam_amd.matcher import ScanMatcher
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 500
ctx = Context(0)
rng = np.random.default_rng(seed0)
bad = 0
for k in range(n):
    nb = int(rng.choice([181, 361, 541, 1081]))
    res = float(rng.choice([0.025, 0.05]))
    kr = float(rng.choice([0.1, 0.15, 0.2, 0.3])) * (2 if res == 0.05 else 1)
    kr = min(kr, 8 * res + 1e-9) if rng.random() < 0.8 else kr          # mostly radii the distance transform takes, some beyond
    ms = float(rng.choice([0.03, 0.15, 0.4]))
    lp = (float(rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.5, 0.5))) if k % 3 == 0 else (0.0, 0.0, 0.0)
    sp = synth.make_scan_pairs(6, seed=seed0 + k, n_beams=nb)
    if nb != 1081: sp["angle_inc"] = synth.LASER_ANGLE_INC * 1081 / nb
    try:
        m = ScanMatcher(ctx, nb, sp["angle_min"], sp["angle_inc"], sp["max_range"], laser_pose=lp, resolution=res, kernel_range=kr)
        a = m.closeScanMatching(sp["ranges_ref"], sp["ranges_qry"], sp["guess"], maxScore=ms)
        b = m.closeScanMatching(sp["ranges_ref"], sp["ranges_qry"], sp["guess"], maxScore=ms, want_nresults=True)
    except Exception as e:                                              # a configuration the library rejects (kernel not representable)
        print(f"{k:3d} beams {nb} res {res} range {kr:.3f} maxScore {ms}: rejected ({e})")
        continue
    xo, so, fo = O.close_scan_match_batch(sp["ranges_ref"], sp["ranges_qry"], sp["angle_min"], sp["angle_inc"], sp["max_range"], lp, sp["guess"],
                                          resolution=res, kernel_range=kr, max_score=ms)
    ok = all(np.array_equal(r[0], fo.astype(bool)) and np.array_equal(r[1], xo) and np.array_equal(r[2], so) for r in (a, b))
    bad += 0 if ok else 1
    print(f"{k:3d} beams {nb:4d} res {res} range {kr:.3f} maxScore {ms} laser {lp}: found {int(a[0].sum())}/6 {'ok' if ok else '<-- MISMATCH'}", flush=True)
print("mismatching configurations:", bad)
