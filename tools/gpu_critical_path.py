"""How much of the factorisation time is the level barriers?  Needs `make EXTRA=-DCGMR_PHASE_TIMING`.
Per work item of k_front_factor (start, end) in 100 MHz ticks; a dependency-driven schedule would let a front start when
its children are done instead of when the whole level below is done: critical path = max over fronts of
own duration + max over children's finish (update tiles of big fronts are charged a flat 8 us)."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth, Context, load_library
ctx = Context(0)
g = synth.make_pose_graph(10000, 40000, seed=12345)
a = (g['poses'], g['fixed'], g['edge_from'], g['edge_to'], g['meas'], g['info'])
for r in range(2):
    rc, p, chi = ctx.gn_optimize(*a, 1)
lib = load_library()
wt = np.zeros(2 * 8192, dtype=np.uint64)
lib.cgmr_debug_worktimes(C.c_void_p(wt.ctypes.data))
cap, fcap = 8192, 8192
front = np.zeros(cap, dtype=np.int32); chunk = np.zeros(cap, dtype=np.int32)
parent = np.zeros(fcap, dtype=np.int32); level = np.zeros(fcap, dtype=np.int32); ns = np.zeros(fcap, dtype=np.int32)
n = lib.cgmr_debug_worklist(ctx.h, C.c_void_p(front.ctypes.data), C.c_void_p(chunk.ctypes.data), C.c_int(cap),
                            C.c_void_p(parent.ctypes.data), C.c_void_p(level.ctypes.data), C.c_void_p(ns.ctypes.data), C.c_int(fcap))
wt = wt.astype(np.int64).reshape(-1, 2)[:n]
dur = (wt[:, 1] - wt[:, 0]) * 0.01            # us
nf = int(front[:n].max()) + 1
fdur = np.zeros(nf)
for k in range(n):
    fdur[front[k]] = max(fdur[front[k]], dur[k])          # chunks of a front run in parallel
upd = np.where(ns[:nf] > 0, 8.0, 0.0)                 # flat charge for the update-tile kernel of big fronts
lev = level[:nf]
per_level = [fdur[lev == l].max() + upd[lev == l].max() for l in range(lev.max() + 1)]
finish = np.zeros(nf)
kids = [[] for _ in range(nf)]
for f in range(nf):
    if parent[f] >= 0: kids[parent[f]].append(f)
for f in sorted(range(nf), key=lambda q: lev[q]):
    finish[f] = fdur[f] + upd[f] + (max(finish[c] for c in kids[f]) if kids[f] else 0.0)
print("work items", n, "fronts", nf, "levels", lev.max() + 1)
print("sum over levels of the slowest work item (+8 us update where needed): %.0f us" % sum(per_level))
print("critical path with dependency-driven start: %.0f us" % finish.max())
print("per level max / median duration (us):", [(round(fdur[lev == l].max(), 1), round(float(np.median(fdur[lev == l])), 1)) for l in range(lev.max() + 1)])

# phase split (cycles) of the slowest work item of every level
ph = np.zeros(8 * 8192, dtype=np.uint64)
lib.cgmr_debug_workphases(C.c_void_p(ph.ctypes.data))
ph = ph.astype(np.int64).reshape(-1, 8)[:n]
print("slowest work item per level: [record+clear, round 2, round 3 (children), factor B+C, stores] cycles | r, chunk, children rows")
for l in range(lev.max() + 1):
    items = [k for k in range(n) if lev[front[k]] == l]
    k = max(items, key=lambda q: dur[q])
    d = np.diff(ph[k, :6])
    f = front[k]
    print(l, d.tolist(), "| r", 3 * ns[f], "chunk", chunk[k], "kids", [3 * ns[c] for c in kids[f]], "dur %.1f us" % dur[k])

try:
    bc = np.zeros(4 * 8192, dtype=np.uint64)
    lib.cgmr_debug_bc(C.c_void_p(bc.ctypes.data))
    bc = bc.astype(np.int64).reshape(-1, 4)[:n]
    print("B+C split of the slowest item per level [(a) mfma update, (b) diagonal block, (c) row solves] cycles (thread 0, incl. barriers):")
    for l in range(lev.max() + 1):
        items = [k for k in range(n) if lev[front[k]] == l]
        k = max(items, key=lambda q: dur[q])
        print(l, bc[k, :3].tolist(), "r", 3 * ns[front[k]])
except AttributeError:
    pass
