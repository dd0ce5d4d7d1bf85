import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth, Context, load_library
from cg_mrslam_amd.matcher import ScanMatcher
ctx = Context(0)
sp = synth.make_scan_pairs(int(os.environ.get("MPHASE_PAIRS", "8")), seed=5)
m = ScanMatcher(ctx, sp["n_beams"], sp["angle_min"], sp["angle_inc"], sp["max_range"])
for r in range(2):
    found, xyt, score = m.closeScanMatching(sp["ranges_ref"], sp["ranges_qry"], sp["guess"])
print("kernel s", m.last_kernel_seconds())
out = np.zeros(32, dtype=np.uint64)
print(load_library().cgmr_debug_mphase(C.c_void_p(out.ctypes.data)))
o = out.astype(np.int64)
names = ["qry cartesian+sort", "subsample means", "ref cells+dir", "dir scan+tile init", "stamp", "window/theta", "search", "result"]
for i, n in enumerate(names):
    print(f"{n:22s} {o[i+1]-o[i]:>10d}")
print("total", o[8]-o[0])
print("  of the first phase: load + keys", o[9]-o[0], " sort", o[1]-o[9])
if o[15] > 0:
    print("  fast search path, all wavefronts, summed over the %d angles: list %d, first pass %d, probe + liveness %d, second pass %d, candidates %d cycles; live rows per angle %.2f of 24"
          % (o[15], o[24], o[10], o[11], o[12], o[13], o[14] / o[15]))
else:
    print("  fast search path, exhaustive, all wavefronts and angles: list %d, gathers %d, [11] %d, [12] %d, candidates %d" % (o[24], o[10], o[11], o[12], o[13]))
if o[25] > 0:
    print("  probe split: lowest bounds %d, first dead total %d, best candidate finished %d, row tests %d" % (o[25], o[26], o[27], o[11]))
if o[21] > o[16] > 0:
    print("  distance-transform rasteriser: cell maps %d, along y %d (+ %d barrier, row 0), along x %d, write back %d" % (o[17]-o[16], o[18]-o[17], o[19]-o[18], o[20]-o[19], o[21]-o[20]), " before it (tile init -> here)", o[16]-o[4])
if len(sys.argv) > 1:
    import json
    json.dump({n: int(o[i + 1] - o[i]) for i, n in enumerate(names)} | {"total": int(o[8] - o[0]), "note": "cycles of one workgroup for one pair (timing build, tools/gpu_mphase.py)"},
              open(sys.argv[1], "w"))
