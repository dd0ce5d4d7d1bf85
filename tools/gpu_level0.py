"""Level-0 (leaf) work items of k_front_factor: start/end distribution, concurrency, mean phase cycles.
Needs a library built with `make EXTRA=-DCGMR_PHASE_TIMING` (CGMR_LIB=...)."""
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
cap = 8192
front = np.zeros(cap, dtype=np.int32); chunk = np.zeros(cap, dtype=np.int32)
parent = np.zeros(cap, dtype=np.int32); level = np.zeros(cap, dtype=np.int32); ns = np.zeros(cap, dtype=np.int32)
n = lib.cgmr_debug_worklist(ctx.h, C.c_void_p(front.ctypes.data), C.c_void_p(chunk.ctypes.data), C.c_int(cap),
                            C.c_void_p(parent.ctypes.data), C.c_void_p(level.ctypes.data), C.c_void_p(ns.ctypes.data), C.c_int(cap))
wt = wt.astype(np.int64).reshape(-1, 2)[:n]
ph = np.zeros(8 * 8192, dtype=np.uint64)
lib.cgmr_debug_workphases(C.c_void_p(ph.ctypes.data))
ph = ph.astype(np.int64).reshape(-1, 8)[:n]
fp = np.zeros(8 * 8192, dtype=np.uint64)
lib.cgmr_debug_factorphases(C.c_void_p(fp.ctypes.data))
fp = fp.astype(np.int64).reshape(-1, 8)[:n]
for L in (0, 1, 2):
    items = np.array([k for k in range(n) if level[front[k]] == L])
    s = (wt[items, 0] - wt[items, 0].min()) * 0.01
    e = (wt[items, 1] - wt[items, 0].min()) * 0.01
    d = e - s
    print(f"level {L}: {len(items)} items; span {e.max():.1f} us; duration min/med/mean/max {d.min():.1f} {np.median(d):.1f} {d.mean():.1f} {d.max():.1f}")
    print("  start-time histogram (5 us bins):", np.histogram(s, bins=np.arange(0, e.max() + 5, 5))[0].tolist())
    ts = np.arange(0, e.max(), 2.0)
    print("  running at t (2 us steps):", [int(((s <= t) & (e > t)).sum()) for t in ts])
    dp = np.diff(ph[items, :6], axis=1)
    full = items[(fp[items, 4] > fp[items, 0])]
    if len(full):
        print("  first steps of the blocked factorisation, fronts with >= 2 block columns (%d) [F0, S0, U(1<-0), F1 | U(2..<-0)]:" % len(full), np.diff(fp[full, :5], axis=1).mean(axis=0).round(0).tolist())
    print("  mean phase cycles [rec+clear, round2, round3, factor, stores]:", dp.mean(axis=0).round(0).tolist(), " border rows mean", 3 * ns[front[items]].mean())
