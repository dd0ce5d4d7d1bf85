"""optimize(N) on one synthetic graph vs the oracle (argv: V E [seed] [iters]); used to bisect size-dependent problems."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth, Context
from cg_mrslam_amd._lib import gn_symbolic_info
from oracle import oracle as O
V, E = int(sys.argv[1]), int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 7
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
g = synth.make_pose_graph(V, E, seed=seed)
a = (g["poses"], g["fixed"], g["edge_from"], g["edge_to"], g["meas"], g["info"])
info = gn_symbolic_info(V, g["fixed"], g["edge_from"], g["edge_to"])
print(V, E, {k: info[k] for k in ("fronts", "levels", "max_border", "max_children")}, flush=True)
ctx = Context(0)
rc, p, chi = ctx.gn_optimize(*a, iters)
t = time.time(); rc, p, chi = ctx.gn_optimize(*a, iters); tg = time.time() - t
print("gpu ok %.3f s" % tg, ctx.gn_last_timing(), flush=True)
t = time.time(); st, p2, chi2, _ = O.gn_optimize(*a, iters); tc = time.time() - t
print("cpu %.2f s; chi2 %.6f vs %.6f rel %.2e; max pose diff %.2e" % (tc, chi[-1], chi2[-1], abs(chi[-1] - chi2[-1]) / chi2[-1], np.abs(p - p2).max()))
