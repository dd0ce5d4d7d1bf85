import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth, Context
from cg_mrslam_amd.matcher import ScanMatcher
from oracle import oracle as O
ctx = Context(0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sp = synth.make_scan_pairs(P, seed=5)
m = ScanMatcher(ctx, sp["n_beams"], sp["angle_min"], sp["angle_inc"], sp["max_range"])
t = time.time(); found, xyt, score, nres = m.closeScanMatching(sp["ranges_ref"], sp["ranges_qry"], sp["guess"], want_nresults=True); tg = time.time() - t
print("gpu wall", tg, "kernel s", m.last_kernel_seconds(), "pairs/s (kernel)", P / m.last_kernel_seconds())
n = min(P, 64)
t = time.time(); xo, so, fo = O.close_scan_match_batch(sp["ranges_ref"][:n], sp["ranges_qry"][:n], sp["angle_min"], sp["angle_inc"], sp["max_range"], [0, 0, 0], sp["guess"][:n]); tc = time.time() - t
print("cpu s/pair", tc / n)
print("found eq", np.array_equal(found[:n], fo.astype(bool)), "xyt eq", np.array_equal(xyt[:n], xo), "score eq", np.array_equal(score[:n], so))
bad = np.flatnonzero((xyt[:n] != xo).any(1) | (score[:n] != so))
print("mismatches", bad[:10], "nres", nres[:8])
for b in bad[:5]:
    print(b, xyt[b], xo[b], score[b], so[b])
print("err vs truth", np.abs(xyt - sp["true_rel"]).max(0))
