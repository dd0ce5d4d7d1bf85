"""C5 at its real shape on one GPU, robots taking turns (loopback exchange): per-robot time of a round's parts, to see what
a rank of an 8-GPU run would spend beside the solo round the bench's N = 1 line reports.  argv: robots rounds [sync]"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth, Context
from cg_mrslam_amd.condensed import RobotGraph
from cg_mrslam_amd.mrslam import RobotRounds, RobotWorld, LoopbackExchange
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ASYNC = (sys.argv[3] != "sync") if len(sys.argv) > 3 else True       # argv[3]: "sync" = condensed graphs waited for, host loopback
ctxs = [Context(0) for _ in range(nr)]                     # one context per robot, as one rank per robot has (own analysis cache)
R = synth.make_multi_robot(nr, 5000, 20000, seed=777)
rounds = [RobotRounds(RobotGraph(ctxs[r], r, nr, cap_edges=128, async_condense=ASYNC), RobotWorld(R, r, chunk=50)) for r in range(nr)]
ex = LoopbackExchange([r.g for r in rounds], device=ASYNC)
G = {"order": 0.0, "structure": 0.0, "upload": 0.0, "device": 0.0}
T = {"grow": 0.0, "optimize": 0.0, "finish(ingest)": 0.0, "condense": 0.0, "start(pack)": 0.0}
built = 0
for t in range(min(n_rounds, rounds[0].w.n_rounds)):
    for r in rounds:
        t0 = time.perf_counter(); r.grow(); t1 = time.perf_counter(); r.optimize(); t2 = time.perf_counter()
        T["grow"] += t1 - t0; T["optimize"] += t2 - t1
        tm = ctxs[r.g.robot].gn_last_timing()
        for k in G: G[k] += tm[k]
    t0 = time.perf_counter(); ex.finish_all(); T["finish(ingest)"] += time.perf_counter() - t0
    t0 = time.perf_counter(); built += sum(r.condense() for r in rounds); T["condense"] += time.perf_counter() - t0
    t0 = time.perf_counter(); ex.start_all(); T["start(pack)"] += time.perf_counter() - t0
for r in rounds: r.g.condensed_wait()
n = (t + 1) * nr
print(("asynchronous condensed graphs, device loopback: " if ASYNC else "synchronous: ") + f"{nr} robots, {t + 1} rounds, {built} condensed graphs built ({built / n:.2f} per robot and round)")
for k, v in T.items(): print(f"  {k:16s} {1e3 * v / n:7.3f} ms per robot and round")
print("  of optimize:", ", ".join(f"{k} {1e3 * v / n:.3f}" for k, v in G.items()), "ms; analysis cache", ctxs[0].symbolic_cache_stats())
print(f"  total            {1e3 * sum(T.values()) / n:7.3f} ms per robot and round")
