"""Randomised GPU-vs-oracle check of the batched close matcher on FULL batches (more pairs than compute units: the kernel
instances built for the common shape, their redo and slow lists, the borrowed tile pool): argv: configs [seed0 [pairs]].
Every configuration: `pairs` pairs (default 288) of rooms scaled by 1 .. 1.5 (long walls: more tiles, more subsampled points),
a few scattered scans, pruned and exhaustive, bit-identical to the oracle (run on all host cores)."""
import sys, os, numpy as np
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth, Context
from cg_mrslam_amd.matcher import ScanMatcher
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 900
P = int(sys.argv[3]) if len(sys.argv) > 3 else 288
ctx = Context(0)
rng = np.random.default_rng(seed0)
bad = 0
for k in range(n):
    nb = int(rng.choice([541, 1081, 1081]))
    res = float(rng.choice([0.025, 0.025, 0.05]))
    kr = 0.2 * (2 if res == 0.05 else 1)
    ms = float(rng.choice([0.15, 0.3]))
    sp = synth.make_scan_pairs(P, seed=seed0 + k, n_beams=nb)
    if nb != 1081: sp["angle_inc"] = synth.LASER_ANGLE_INC * 1081 / nb
    rr, rq, g = sp["ranges_ref"].copy(), sp["ranges_qry"].copy(), sp["guess"].copy()
    scale = rng.uniform(1.0, 1.5, size=P).astype(np.float32)       # the same room, larger (the guess's translation with it)
    rr *= scale[:, None]; rq *= scale[:, None]; g[:, :2] *= scale[:, None]
    for i in rng.choice(P, size=4, replace=False):
        which = int(rng.integers(3))
        scat = rng.uniform(1.0, 14.0, size=nb).astype(np.float32)
        if which != 1: rr[i] = scat
        if which != 0: rq[i] = scat + rng.normal(scale=0.01, size=nb).astype(np.float32) if which == 2 else rng.uniform(1.0, 14.0, size=nb).astype(np.float32)
        if which == 2: g[i] = 0.0
    m = ScanMatcher(ctx, nb, sp["angle_min"], sp["angle_inc"], sp["max_range"], resolution=res, kernel_range=kr)
    a = m.closeScanMatching(rr, rq, g, maxScore=ms); sa = m.last_stats()
    b = m.closeScanMatching(rr, rq, g, maxScore=ms, want_nresults=True)
    nthr = min(os.cpu_count() or 1, 64)
    per = (P + nthr - 1) // nthr
    def run(t):
        lo, hi = t * per, min(P, (t + 1) * per)
        if lo >= hi: return None
        return O.close_scan_match_batch(rr[lo:hi], rq[lo:hi], sp["angle_min"], sp["angle_inc"], sp["max_range"], (0.0, 0.0, 0.0), g[lo:hi],
                                        resolution=res, kernel_range=kr, max_score=ms)
    with ThreadPoolExecutor(max_workers=nthr) as ex: parts = [p for p in ex.map(run, range(nthr)) if p is not None]
    xo = np.concatenate([p[0] for p in parts]); so = np.concatenate([p[1] for p in parts]); fo = np.concatenate([p[2] for p in parts])
    ok = all(np.array_equal(r[0], fo.astype(bool)) and np.array_equal(r[1], xo) and np.array_equal(r[2], so) for r in (a, b))
    bad += 0 if ok else 1
    wrong = [int(i) for i in np.nonzero(~(np.all(a[1] == xo, axis=1) & (a[2] == so) & (a[0] == fo.astype(bool))))[0][:8]]
    print(f"{k:3d} beams {nb:4d} res {res} maxScore {ms}: found {int(a[0].sum())}/{P} paths {sa} {'ok' if ok else '<-- MISMATCH at ' + str(wrong)}", flush=True)
print("mismatching configurations:", bad)
