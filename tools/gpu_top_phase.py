"""Cycle marks of k_top_block's own workgroup on the C2 graph (timing build: CGMR_LIB=.../libcgmr_t.so): start -> block cleared, H blocks
and the children's tables in -> children streamed -> factorised -> back-solved and stored."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth, Context, load_library
ctx = Context(0)
g = synth.make_pose_graph(10000, 40000, seed=12345, strict=True)
a = (g['poses'], g['fixed'], g['edge_from'], g['edge_to'], g['meas'], g['info'])
for r in range(2):
    rc, p, chi = ctx.gn_optimize(*a, 3)
out = np.zeros(16, dtype=np.uint64)
assert load_library().cgmr_debug_topphase(C.c_void_p(out.ctypes.data)) == 0
d = np.diff(out[:5].astype(np.int64))
print("top block cycles: clear + H blocks + tables %d, children %d, factorisation %d, backward solve + stores %d, total %d" % (d[0], d[1], d[2], d[3], int(out[4] - out[0])))
