"""Condensed graphs for 7 peers on a C5-sized robot graph: all peers in one call (concurrent passes on side streams)
against one peer at a time."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cg_mrslam_amd import Context, synth
from cg_mrslam_amd.condensed import RobotGraph

ctx = Context(0)
R = synth.make_multi_robot(8, 5000, 20000, seed=777)
g0 = R[0]
g = RobotGraph(ctx, 0, 8)
g.add_vertices(g0["ids"], g0["poses_all"], g0["fixed_all"])
g.add_edges(g0["ids"][g0["ef_all"]], g0["ids"][g0["et_all"]], g0["meas_all"], g0["info_all"])
for p, idx in g0["out_closures"].items():
    g.insertOutClosure(p, g0["ids"][idx])
rc, chi = g.optimize(5)
print("optimize(5)", rc, chi[-1], "%.2f ms" % (1e3 * g.last_seconds()["optimize"]))
for rep in range(3):
    t0 = time.perf_counter(); n = g.computeCondensedGraph(-1); t1 = time.perf_counter()
    each = []
    for p in g0["out_closures"]:
        t2 = time.perf_counter(); g.computeCondensedGraph(p); each.append(time.perf_counter() - t2)
    print(f"all {n} peers in one call: {1e3 * (t1 - t0):.2f} ms; one at a time: {1e3 * sum(each):.2f} ms ({', '.join('%.2f' % (1e3 * e) for e in each)})")
print({p: len(v) for p, v in g0["out_closures"].items()})
