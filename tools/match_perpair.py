"""Per-pair kernel time of the close matcher (single-pair launches) to expose outliers."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cg_mrslam_amd import synth, Context
from cg_mrslam_amd.matcher import ScanMatcher
ctx = Context(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sp = synth.make_scan_pairs(N, seed=5)
m = ScanMatcher(ctx, sp["n_beams"], sp["angle_min"], sp["angle_inc"], sp["max_range"])
m.closeScanMatching(sp["ranges_ref"][:8], sp["ranges_qry"][:8], sp["guess"][:8])
t = np.zeros(N)
for i in range(N):
    m.closeScanMatching(sp["ranges_ref"][i:i+1], sp["ranges_qry"][i:i+1], sp["guess"][i:i+1])
    t[i] = m.last_kernel_seconds() * 1e3
print("ms per pair: min %.3f median %.3f mean %.3f p90 %.3f p99 %.3f max %.3f" % (t.min(), np.median(t), t.mean(), np.percentile(t, 90), np.percentile(t, 99), t.max()))
worst = np.argsort(t)[-8:]
nvalid_q = ((sp["ranges_qry"] < sp["max_range"]) & (sp["ranges_qry"] > 0)).sum(1)
nvalid_r = ((sp["ranges_ref"] < sp["max_range"]) & (sp["ranges_ref"] > 0)).sum(1)
for i in worst: print(i, "%.3f ms" % t[i], "valid beams q/r", nvalid_q[i], nvalid_r[i], "mean range q", sp["ranges_qry"][i].mean())
print("corr(time, mean ref range)", np.corrcoef(t, sp["ranges_ref"].mean(1))[0, 1])
m.closeScanMatching(sp["ranges_ref"], sp["ranges_qry"], sp["guess"])
print("batch of", N, "kernel ms", m.last_kernel_seconds() * 1e3, "sum of singles / 256 =", t.sum() / 256)
