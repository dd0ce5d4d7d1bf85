import sys, os, time, numpy as np
for _k, _v in {"CGMR_HOST_PIN_CALLER": "1", "CGMR_HOST_SPIN_US": "10000"}.items(): os.environ.setdefault(_k, _v)
sys.path.insert(0, os.getcwd())
from cg_mrslam_amd import synth, Context
from cg_mrslam_amd.condensed import RobotGraph
from cg_mrslam_amd.mrslam import RobotRounds, RobotWorld
R = synth.make_multi_robot(8, 5000, 20000, seed=777)
ctx = Context(0)
for r in range(3):
    rr = RobotRounds(RobotGraph(ctx, 0, 1, cap_edges=128), RobotWorld(R, r, chunk=50, closures=False))
    G = {"order": 0., "structure": 0., "upload": 0., "device": 0., "total": 0.}
    T = {"grow": 0., "optimize": 0.}
    n = rr.w.n_rounds
    for _ in range(n):
        t0 = time.perf_counter(); rr.grow(); t1 = time.perf_counter(); rr.optimize(); t2 = time.perf_counter()
        T["grow"] += t1 - t0; T["optimize"] += t2 - t1
        tm = ctx.gn_last_timing()
        for k in G: G[k] += tm[k]
    print("robot", r, {k: round(1e3 * v / n, 3) for k, v in T.items()}, {k: round(1e3 * v / n, 3) for k, v in G.items()}, ctx.symbolic_cache_stats())
    rr.g.close()
    ctx.set_symbolic_cache(False); ctx.set_symbolic_cache(True)
