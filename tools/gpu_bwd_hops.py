"""The chained backward solve front by front along the tree's longest path (timing build: CGMR_LIB=.../libcgmr_t.so): per front the
100 MHz clock at start / L11 inverted / x of the border there / own x stored -- what a hop (parent's x stored -> own x stored) is made of."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth, Context, load_library
ctx = Context(0)
g = synth.make_pose_graph(10000, 40000, seed=12345, strict=True)
a = (g['poses'], g['fixed'], g['edge_from'], g['edge_to'], g['meas'], g['info'])
for r in range(2):
    rc, p, chi = ctx.gn_optimize(*a, 3)
lib = load_library()
out = np.zeros(8 * 8192, dtype=np.uint64)
assert lib.cgmr_debug_bwdtimes(C.c_void_p(out.ctypes.data)) == 0
t = out.reshape(-1, 8).astype(np.int64)
live = np.nonzero(t[:, 3])[0]
t0 = t[live, 0].min()
us = (t - t0) * 0.01
last = live[np.argmax(t[live, 3])]
print("fronts %d, launch span %.2f us; the front that stores last: %d" % (len(live), us[live, 3].max(), last))
# the real tree (cgmr_debug_fronts: c0, nc, ns, parent, level, nchild per front): walk up from the front that stores last
cap = 20000
fo = np.zeros(cap * 6, dtype=np.int32)
fx = np.ascontiguousarray(g["fixed"], dtype=np.uint8); ef = np.ascontiguousarray(g["edge_from"], dtype=np.int32); et = np.ascontiguousarray(g["edge_to"], dtype=np.int32)
n = lib.cgmr_debug_fronts(C.c_int(10000), C.c_void_p(fx.ctypes.data), C.c_int(len(ef)), C.c_void_p(ef.ctypes.data), C.c_void_p(et.ctypes.data), C.c_int(cap), C.c_void_p(fo.ctypes.data))
F = fo[:6 * n].reshape(n, 6)
print("front level  ns   start  loads out  x there  stored | x there->stored (in LDS, border product, v, dot + store)  parent stored->x there")
chain = [int(last)]
while F[chain[-1], 3] >= 0 and t[F[chain[-1], 3], 3] != 0: chain.append(int(F[chain[-1], 3]))
prev = None
for f in reversed(chain):
    print("%5d  %3d  %3d  %6.2f  %6.2f  %6.2f  %6.2f | %5.2f (%.2f %.2f %.2f %.2f)  %s" % (f, F[f, 4], F[f, 2], us[f, 0], us[f, 1], us[f, 2], us[f, 3], us[f, 3] - us[f, 2], us[f, 4] - us[f, 2], us[f, 5] - us[f, 4], us[f, 6] - us[f, 5], us[f, 3] - us[f, 6], "" if prev is None else "%.2f" % (us[f, 2] - us[prev, 3])))
    prev = f
inv = us[live, 1]
print("loads issued (all fronts): min %.2f median %.2f max %.2f us after the first start" % (inv.min(), np.median(inv), inv.max()))
