"""Run the batched close matcher on N synthetic pairs (used under rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cg_mrslam_amd import synth, Context
from cg_mrslam_amd.matcher import ScanMatcher
ctx = Context(0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sp = synth.make_scan_pairs(256, seed=5)
rr = np.tile(sp["ranges_ref"], (P // 256, 1)); rq = np.tile(sp["ranges_qry"], (P // 256, 1)); g = np.tile(sp["guess"], (P // 256, 1))
m = ScanMatcher(ctx, sp["n_beams"], sp["angle_min"], sp["angle_inc"], sp["max_range"])
for r in range(2):
    found, xyt, score = m.closeScanMatching(rr, rq, g)
print("kernel s", m.last_kernel_seconds(), "pairs/s", P / m.last_kernel_seconds())
