"""Latency of ONE closeScanMatching call (the reference's call shape: once per key frame): wall per call and kernel time,
for the workgroups-per-pair settings given (CGMR_MATCH_SPLIT is read once per process: one subprocess each)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
from cg_mrslam_amd import synth, Context
from cg_mrslam_amd.matcher import ScanMatcher
ctx = Context(0)
sp = synth.make_scan_pairs(64, seed=5)
m = ScanMatcher(ctx, sp["n_beams"], sp["angle_min"], sp["angle_inc"], sp["max_range"])
ref = m.closeScanMatching(sp["ranges_ref"], sp["ranges_qry"], sp["guess"])          # batch result to compare with
wall, kern = [], []
same = True
for i in range(64):
    t0 = time.perf_counter()
    f, x, s = m.closeScanMatching(sp["ranges_ref"][i:i+1], sp["ranges_qry"][i:i+1], sp["guess"][i:i+1])
    wall.append(time.perf_counter() - t0); kern.append(m.last_kernel_seconds())
    same = same and bool(f[0] == ref[0][i]) and np.array_equal(x[0], ref[1][i]) and s[0] == ref[2][i]
import os
print("CGMR_MATCH_SPLIT=%%s: wall per call median %%.1f us, kernel median %%.1f us (min %%.1f), identical to the batch result: %%s" %% (
      os.environ.get("CGMR_MATCH_SPLIT", "default"), 1e6 * np.median(wall[8:]), 1e6 * np.median(kern[8:]), 1e6 * min(kern), same))
''' % ROOT
for split in (sys.argv[1:] or ["1", "4", "8", "16"]):
    out = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, CGMR_MATCH_SPLIT=split), capture_output=True, text=True)
    print((out.stdout.strip().splitlines() or [out.stderr[-400:]])[-1])
