"""One hierarchical search against the oracle, with the first differing rows (argv: levels max_score)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cg_mrslam_amd import Context, synth
from tests.test_matcher_gpu import _lc
from oracle import oracle
levels, max_score = int(sys.argv[1]), float(sys.argv[2])
ctx = Context(0)
oracle.lib()
sp = synth.make_scan_pairs(2, seed=77)
m = _lc(ctx, sp)
ref = m.cartesian(sp["ranges_ref"][0]); q = m.subsample(m.cartesian(sp["ranges_qry"][0]))
region = np.array([[-6, -4, np.float32(-np.pi), 6, 4, np.float32(np.pi)]], dtype=np.float32)
got = np.asarray(m.hierarchicalSearch(ref, q, region, 0.025, max_score, 0.5, 0.5, 0.2, levels))
n, want = oracle.hierarchical_search((-35, -35), (35, 35), 0.1, 0.1, 0.5, ref, q, region, 0.025, max_score, 0.5, 0.5, 0.2, levels)
print(len(got), n)
if len(got) == n:
    bad = np.nonzero((got != want).any(axis=1))[0]
    print("differing rows", len(bad), bad[:10])
    for i in bad[:5]: print(i, got[i], want[i])
    a = {tuple(r) for r in got.tolist()}; b = {tuple(r) for r in want.tolist()}
    print("as sets: only gpu", len(a - b), "only oracle", len(b - a))
