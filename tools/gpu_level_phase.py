"""A merged level launch (k_front_level) from the inside, per tree level of the C2 graph (timing build: CGMR_LIB=.../libcgmr_t.so): 100 MHz
clock marks -- first tile at its wait, last factor work item signalled, last tile saw its flag, last slices staged, last tile done."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth, Context, load_library
ctx = Context(0)
g = synth.make_pose_graph(10000, 40000, seed=12345, strict=True)
a = (g['poses'], g['fixed'], g['edge_from'], g['edge_to'], g['meas'], g['info'])
lib = load_library()
rc, p, chi = ctx.gn_optimize(*a, 3)
lib.cgmr_debug_leveltimes(None, 1)
rc, p, chi = ctx.gn_optimize(*a, 1)
out = np.zeros(8 * 64, dtype=np.uint64)
assert lib.cgmr_debug_leveltimes(C.c_void_p(out.ctypes.data), 0) == 0
t = out.reshape(64, 8).astype(np.float64) * 0.01
print("level | first tile waiting -> last item signalled -> last flag seen -> last slices staged -> last tile done (us after the first tile's wait)")
for l in range(64):
    if t[l, 5] == 0: continue
    t0 = t[l, 0]
    print("%3d   | last tile waiting %6.2f  signalled %6.2f  seen %6.2f (+%.2f)  staged %6.2f (+%.2f)  done %6.2f (+%.2f)" % (l, t[l, 2] - t0, t[l, 1] - t0, t[l, 3] - t0, t[l, 3] - t[l, 1], t[l, 4] - t0, t[l, 4] - t[l, 3], t[l, 5] - t0, t[l, 5] - t[l, 4]))
