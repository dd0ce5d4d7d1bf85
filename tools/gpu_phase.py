import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth, Context, load_library
ctx = Context(0)
g = synth.make_pose_graph(10000, 40000, seed=12345)
a = (g['poses'], g['fixed'], g['edge_from'], g['edge_to'], g['meas'], g['info'])
for r in range(2):
    rc, p, chi = ctx.gn_optimize(*a, 3)
out = np.zeros(64*8, dtype=np.uint64)
print(load_library().cgmr_debug_phase(C.c_void_p(out.ctypes.data)))
out = out.reshape(64, 8).astype(np.int64)
print("level: [record+clear, round 2 (rhs/H/maps), round 3 (children), factor B+C, stores, fused U] total | realtime ticks (100 MHz) -> shader GHz")
for l in range(22):
    d = np.diff(out[l,:7])
    tot = out[l,6]-out[l,0]
    rt = out[l,7]
    print(l, d.tolist(), 'total', tot, '| rt', rt, 'GHz %.2f' % (tot / max(rt,1) / 10.0))
