"""What the condensed edges received from the peers do to the elimination tree: the C5 rounds of N robots in loopback, then
the analysis of robot 0's final graph with and without the received edges (host only after the rounds).  argv: robots rounds"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth, Context
from cg_mrslam_amd._lib import gn_symbolic_info
from cg_mrslam_amd.condensed import RobotGraph
from cg_mrslam_amd.mrslam import RobotRounds, RobotWorld, LoopbackExchange
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctxs = [Context(0) for _ in range(nr)]
R = synth.make_multi_robot(nr, 5000, 20000, seed=777)
rounds = [RobotRounds(RobotGraph(ctxs[r], r, nr, cap_edges=128), RobotWorld(R, r, chunk=50)) for r in range(nr)]
ex = LoopbackExchange([r.g for r in rounds])
for t in range(min(n_rounds, rounds[0].w.n_rounds)):
    for r in rounds: r.grow(); r.optimize()
    ex.finish_all()
    for r in rounds: r.condense()
    ex.start_all()
ex.finish_all()
g = rounds[0].g
cap = 200000
ef, et = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
nown = C.c_int32(0)
n = g.lib.cgmr_graph_debug_edges(g.h, C.c_int(cap), C.c_void_p(ef.ctypes.data), C.c_void_p(et.ctypes.data), C.byref(nown))
ef, et = ef[:n], et[:n]
nV = g.counts()["vertices"]
fixed = np.zeros(nV, np.uint8)
deg = np.bincount(np.concatenate([ef[nown.value:], et[nown.value:]]), minlength=nV)
print("vertices", nV, "own edges", nown.value, "received edges", n - nown.value, "largest degrees among the received edges", sorted(deg.tolist())[-8:])
for name, m in (("own edges only", nown.value), ("own + received", n)):
    i = gn_symbolic_info(nV, fixed, ef[:m], et[:m])
    print(f"{name:16s}: fronts {i['fronts']} levels {i['levels']} max border {i['max_border']} L doubles {i['L_doubles']} U doubles {i['U_doubles']} flops {i['factor_flops']}")
