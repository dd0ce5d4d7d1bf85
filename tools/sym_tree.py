"""Shape of the elimination tree of the C2 graph (host only): fronts per level and the longest root-to-leaf path."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth
from cg_mrslam_amd._lib import load_library
g = synth.make_pose_graph(10000, 40000, seed=12345)
lib = load_library()
cap = 20000
out = np.zeros(cap * 6, dtype=np.int32)
fx = np.ascontiguousarray(g["fixed"], dtype=np.uint8); ef = np.ascontiguousarray(g["edge_from"], dtype=np.int32); et = np.ascontiguousarray(g["edge_to"], dtype=np.int32)
n = lib.cgmr_debug_fronts(C.c_int(10000), C.c_void_p(fx.ctypes.data), C.c_int(len(ef)), C.c_void_p(ef.ctypes.data), C.c_void_p(et.ctypes.data), C.c_int(cap), C.c_void_p(out.ctypes.data))
F = out[:6 * n].reshape(n, 6)
lev = F[:, 4]
print("fronts", n, "levels", lev.max() + 1)
print("fronts per level:", [int((lev == l).sum()) for l in range(lev.max() + 1)])
print("max nc per level:", [int(F[lev == l, 1].max()) for l in range(lev.max() + 1)])
print("mean nc per level:", [round(float(F[lev == l, 1].mean()), 1) for l in range(lev.max() + 1)])
# longest path: start from a root at max level, descend into the child with the highest level
kids = [[] for _ in range(n)]
for f in range(n):
    if F[f, 3] >= 0: kids[F[f, 3]].append(f)
f = int(np.argmax(lev))
path = []
while True:
    path.append(f)
    if not kids[f]: break
    f = max(kids[f], key=lambda q: lev[q])
print("longest path, root first (front: c0 nc ns nchild level):")
for f in path:
    print("  %5d: c0 %5d nc %2d ns %3d nchild %2d level %2d  next-front-adjacent %s" % (f, F[f, 0], F[f, 1], F[f, 2], F[f, 5], F[f, 4], "yes" if (f + 1 < n and F[f, 3] == f + 1) else "no"), " children (id:level:nc)", [(k, int(lev[k]), int(F[k, 1])) for k in kids[f]])
wide0 = [f for f in range(n) if F[f, 1] > 16 and lev[f] <= 2]
print("wide fronts at levels <= 2:", len(wide0))
for f in wide0[:12]:
    print("  %5d: c0 %5d nc %2d ns %3d nchild %2d level %2d parent %d (parent level %d nc %d)  kids %s" % (f, F[f, 0], F[f, 1], F[f, 2], F[f, 5], F[f, 4], F[f, 3], lev[F[f, 3]], F[F[f, 3], 1], [(k, int(lev[k]), int(F[k, 1])) for k in kids[f]]))
