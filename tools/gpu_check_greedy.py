import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth, Context
from cg_mrslam_amd.matcher import LCScanMatcher
from oracle import oracle as O
ctx = Context(0)
sp = synth.make_scan_pairs(3, seed=11)
m = LCScanMatcher(ctx, sp["n_beams"], sp["angle_min"], sp["angle_inc"], sp["max_range"])
for p in range(3):
    ref = m.cartesian(sp["ranges_ref"][p]); q = m.subsample(m.cartesian(sp["ranges_qry"][p]))
    oref = O.cartesian(sp["ranges_ref"][p], sp["angle_min"], sp["angle_inc"], sp["max_range"]); oq = O.subsample(O.cartesian(sp["ranges_qry"][p], sp["angle_min"], sp["angle_inc"], sp["max_range"]))
    assert np.array_equal(ref, oref) and np.array_equal(q, oq)
    g = sp["guess"][p]
    regs = np.array([[-.5 + g[0], -1.5 + g[1], -.8 + g[2], .5 + g[0], 1.5 + g[1], .8 + g[2]],
                     [-.5 + g[0] + 1, -1.5 + g[1], -.8 + g[2] + 0.3, .5 + g[0] + 1, 1.5 + g[1], .8 + g[2] + 0.3],
                     [-.5, -1.5, -.8, .5, 1.5, .8], [0, 0, 0, .3, .3, .1], [-.2, .1, -.3, .4, 1.0, .3]], dtype=np.float32)
    t = time.time(); got = m.greedySearch(ref, q, regs, 0.025, 0.3, 0.5, 0.5, 0.2); tg = time.time() - t
    t = time.time(); n, want = O.greedy_search((-35, -35), (35, 35), 0.1, 0.1, 0.5, ref, q, regs, 0.1, 0.025, 0.3, 0.5, 0.5, 0.2); tc = time.time() - t
    print(p, "greedy gpu %.4f cpu %.4f" % (tg, tc), len(got), n, "equal", np.array_equal(got, want[:len(got)]), flush=True)
    t = time.time(); goth = m.hierarchicalSearch(ref, q, np.array([[-10, -5, np.float32(-np.pi), 10, 5, np.float32(np.pi)]], dtype=np.float32), 0.025, 0.2, 0.5, 0.5, 0.2, 4); tg = time.time() - t
    t = time.time(); n, wanth = O.hierarchical_search((-35, -35), (35, 35), 0.1, 0.1, 0.5, ref, q, np.array([[-10, -5, np.float32(-np.pi), 10, 5, np.float32(np.pi)]], dtype=np.float32), 0.025, 0.2, 0.5, 0.5, 0.2, 4); tc = time.time() - t
    print(p, "hier gpu %.4f cpu %.4f" % (tg, tc), len(goth), n, "equal", np.array_equal(goth, wanth[:len(goth)]), goth[:1], sp["true_rel"][p])
