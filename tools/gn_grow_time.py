"""optimize(5) on a 5000-vertex graph grown 50 vertices at a time (the C5 round pattern): host ordering / structure and device
time per round, with the incremental ordering (default) or without (CGMR_SYM_EXTEND=0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cg_mrslam_amd import Context, synth
ctx = Context(0)
g = synth.make_pose_graph(5000, 20000, seed=77)
k = np.argsort(np.maximum(g["edge_from"], g["edge_to"]), kind="stable")
ef, et, meas, info = (np.ascontiguousarray(g[n][k]) for n in ("edge_from", "edge_to", "meas", "info"))
last = np.maximum(ef, et)
for rep in range(2):
    ctx.set_symbolic_cache(False); ctx.set_symbolic_cache(True)
    T = []
    for r in range(100):
        nv = 50 * (r + 1); ne = int(np.searchsorted(last, nv, side="left"))
        t0 = time.perf_counter()
        rc, p, chi = ctx.gn_optimize(g["poses"][:nv], g["fixed"][:nv], ef[:ne], et[:ne], meas[:ne], info[:ne], 5)
        tm = ctx.gn_last_timing()
        T.append((tm["order"], tm["structure"], tm["upload"], tm["device"], time.perf_counter() - t0))
T = 1e3 * np.array(T)
print("CGMR_SYM_EXTEND=%s  mean ms per round: order %.3f structure %.3f upload %.3f device %.3f wall %.3f | last 20 rounds: order %.3f structure %.3f device %.3f | cache %s" % (
    os.environ.get("CGMR_SYM_EXTEND", "1"), *T.mean(axis=0), *T[-20:, [0, 1, 3]].mean(axis=0), ctx.symbolic_cache_stats()))
