"""Phase cycles (timing build) of workgroup 0 for ONE closeScanMatching call with a 6-scan reference set (the key-frame call shape)."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth, Context, load_library
from cg_mrslam_amd.matcher import ScanMatcher
ctx = Context(0)
ang = synth.LASER_ANGLE_MIN + synth.LASER_ANGLE_INC * np.arange(1081)
boxes = [(-6.0, -4.5, 6.0, 4.5), (1.0, 1.0, 2.2, 2.0), (-3.0, -2.5, -2.2, -1.0)]
poses = [np.array([0.12 * k, -0.06 * k + 0.01 * k * k, 0.04 * k]) for k in range(6)]
scans = [(synth._raycast_boxes(p[0], p[1], p[2] + ang, boxes, 30.0).astype(np.float32), p) for p in poses]
cur_true = np.array([0.85, -0.25, 0.22])
cur_r = synth._raycast_boxes(cur_true[0], cur_true[1], cur_true[2] + ang, boxes, 30.0).astype(np.float32)
m = ScanMatcher(ctx, 1081, synth.LASER_ANGLE_MIN, synth.LASER_ANGLE_INC, 30.0)
ks = []
for r in range(6):
    f, t = m.closeScanMatchingVSet(scans, 5, cur_r, cur_true + [0.06, -0.05, 0.03], 0.15)
    ks.append(m.last_kernel_seconds())
print("found", f, "kernel us", [round(1e6 * k, 1) for k in ks], "split", os.environ.get("CGMR_MATCH_SPLIT", "default"))
out = np.zeros(32, dtype=np.uint64)
load_library().cgmr_debug_mphase(C.c_void_p(out.ctypes.data))
o = out.astype(np.int64)
names = ["qry cartesian+sort", "subsample means", "ref cells+dir", "dir scan+tile init", "stamp", "window/theta", "search", "result"]
print({n: int(o[i + 1] - o[i]) for i, n in enumerate(names)}, "total", int(o[8] - o[0]))
