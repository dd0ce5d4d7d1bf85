"""PCIe-inclusive rate of the batched close matcher: host buffers in, host results out (cgmr_match_close_batch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cg_mrslam_amd import Context, synth
from cg_mrslam_amd.matcher import ScanMatcher
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sp = synth.make_scan_pairs(256, seed=4242)
rep = (n + 255) // 256
rr, rq, g = (np.tile(sp[k], (rep, 1))[:n] for k in ("ranges_ref", "ranges_qry", "guess"))
m = ScanMatcher(Context(0), sp["n_beams"], sp["angle_min"], sp["angle_inc"], sp["max_range"])
m.closeScanMatching(rr[:256], rq[:256], g[:256])
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    m.closeScanMatching(rr, rq, g)
    ts.append(time.perf_counter() - t0)
w = float(np.median(ts))
print(f"{n} pairs from host buffers: wall {1e3 * w:.2f} ms = {n / w:.0f} pairs/s, of which kernel {1e3 * m.last_kernel_seconds():.2f} ms")
