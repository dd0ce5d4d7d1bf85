"""Kernel time of the batched close matcher vs. batch size (same 32 synthetic pairs tiled)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cg_mrslam_amd import synth, Context
from cg_mrslam_amd.matcher import ScanMatcher
ctx = Context(0)
sp = synth.make_scan_pairs(32, seed=5)
m = ScanMatcher(ctx, sp["n_beams"], sp["angle_min"], sp["angle_inc"], sp["max_range"])
for P in [32, 64, 128, 256, 512, 1024, 4096, 16384]:
    rr = np.tile(sp["ranges_ref"], (P // 32, 1)); rq = np.tile(sp["ranges_qry"], (P // 32, 1)); g = np.tile(sp["guess"], (P // 32, 1))
    ts = []
    for r in range(3):
        m.closeScanMatching(rr, rq, g)
        ts.append(m.last_kernel_seconds())
    t = min(ts)
    print(f"P={P:6d} kernel {t*1e3:8.3f} ms  pairs/s {P/t:10.0f}  ms per round of 256: {t*1e3/max(1,P/256):7.3f}")
