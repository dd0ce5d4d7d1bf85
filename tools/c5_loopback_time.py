"""C5 at its real shape on one GPU (loopback exchange): per-robot time of a round's parts, to see what a rank of an 8-GPU run
would spend beside the same robots' rounds alone.  argv: robots rounds [turns|async|sync]
  turns (default)  condensed graphs on the side streams, device loopback, the robots taking turns with whole rounds (TakeTurns)
  async            the same in lock step (everybody solves, everybody ingests, everybody condenses, everybody packs)
  sync             condensed graphs waited for, host loopback, lock step (rounds 1-3)"""
import sys, os, time, numpy as np
for _k, _v in {"CGMR_HOST_PIN_CALLER": "1", "CGMR_HOST_SPIN_US": "10000"}.items():      # as bench.py opts in (a dedicated solve loop)
    os.environ.setdefault(_k, _v)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth, Context
from cg_mrslam_amd.condensed import RobotGraph
from cg_mrslam_amd.mrslam import RobotRounds, RobotWorld, LoopbackExchange
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 100
MODE = sys.argv[3] if len(sys.argv) > 3 else "turns"
ASYNC = MODE != "sync"
ctxs = [Context(0) for _ in range(nr)]                     # one context per robot, as one rank per robot has (own analysis cache)
R = synth.make_multi_robot(nr, 5000, 20000, seed=777)
# the same robots alone (own vertices and edges, no closures, no peers): the one-rank reference
solo, solo_med = [], []
for r in range(nr):
    rr = RobotRounds(RobotGraph(ctxs[r], 0, 1, cap_edges=128), RobotWorld(R, r, chunk=50, closures=False))
    per = []
    for _ in range(min(n_rounds, rr.w.n_rounds)):
        t0 = time.perf_counter(); rr.grow(); rr.optimize(); per.append(1e3 * (time.perf_counter() - t0))
    solo.append(float(np.mean(per))); solo_med.append(per)
    rr.g.close()
solo_round_med = np.median(np.mean(np.array(solo_med), axis=0))          # median over the rounds of the robots' mean
for c in ctxs:
    c.set_symbolic_cache(False); c.set_symbolic_cache(True)
rounds = [RobotRounds(RobotGraph(ctxs[r], r, nr, cap_edges=128, async_condense=ASYNC), RobotWorld(R, r, chunk=50)) for r in range(nr)]
ex = LoopbackExchange([r.g for r in rounds], device=ASYNC)
G = {"order": 0.0, "structure": 0.0, "upload": 0.0, "device": 0.0}
T = {"grow": 0.0, "optimize": 0.0, "finish(ingest)": 0.0, "condense": 0.0, "start(pack)": 0.0}
built = 0
round_ms = []
for t in range(min(n_rounds, rounds[0].w.n_rounds)):
    t_round0 = time.perf_counter()
    if MODE == "turns":
        for r in rounds:
            g = r.g
            t0 = time.perf_counter(); r.grow(); t1 = time.perf_counter(); r.optimize(); t2 = time.perf_counter()
            tm = ctxs[g.robot].gn_last_timing()
            for k in G: G[k] += tm[k]
            if t > 0: g.ingest_delivered()
            t3 = time.perf_counter(); built += r.condense(); t4 = time.perf_counter()
            g.pack(0)
            for o in rounds:
                if o is not r: g.deliver(o.g)
            t5 = time.perf_counter()
            T["grow"] += t1 - t0; T["optimize"] += t2 - t1; T["finish(ingest)"] += t3 - t2; T["condense"] += t4 - t3; T["start(pack)"] += t5 - t4
        round_ms.append(1e3 * (time.perf_counter() - t_round0) / nr)
        continue
    for r in rounds:
        t0 = time.perf_counter(); r.grow(); t1 = time.perf_counter(); r.optimize(); t2 = time.perf_counter()
        T["grow"] += t1 - t0; T["optimize"] += t2 - t1
        tm = ctxs[r.g.robot].gn_last_timing()
        for k in G: G[k] += tm[k]
    t0 = time.perf_counter(); ex.finish_all(); T["finish(ingest)"] += time.perf_counter() - t0
    t0 = time.perf_counter(); built += sum(r.condense() for r in rounds); T["condense"] += time.perf_counter() - t0
    t0 = time.perf_counter(); ex.start_all(); T["start(pack)"] += time.perf_counter() - t0
    round_ms.append(1e3 * (time.perf_counter() - t_round0) / nr)
for r in rounds: r.g.condensed_wait()
n = (t + 1) * nr
print(f"{MODE}: {nr} robots, {t + 1} rounds, {built} condensed graphs built ({built / n:.2f} per robot and round)")
for k, v in T.items(): print(f"  {k:16s} {1e3 * v / n:7.3f} ms per robot and round")
print("  of optimize:", ", ".join(f"{k} {1e3 * v / n:.3f}" for k, v in G.items()), "ms; analysis cache", ctxs[0].symbolic_cache_stats())
tot = 1e3 * sum(T.values()) / n
print(f"  total            {tot:7.3f} ms per robot and round")
print(f"  medians over the rounds: with peers {np.median(round_ms):.3f} ms, the same robots alone {solo_round_med:.3f} ms  ->  efficiency {solo_round_med / np.median(round_ms):.3f}; host load {os.getloadavg()[0]:.1f}")
print(f"  the same robots alone: mean {np.mean(solo):.3f} ms per round (" + " ".join(f"{v:.2f}" for v in solo) + f")  ->  efficiency {np.mean(solo) / tot:.3f}")
