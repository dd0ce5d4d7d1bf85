"""optimize(3) on a hub graph (argv: K Lc H) vs the oracle: borders of ~H poses at small vertex counts."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth, Context
from cg_mrslam_amd._lib import gn_symbolic_info
from oracle import oracle as O
K, Lc, H = (int(v) for v in sys.argv[1:4])
g = synth.make_hub_graph(K, Lc, H)
a = (g["poses"], g["fixed"], g["edge_from"], g["edge_to"], g["meas"], g["info"])
info = gn_symbolic_info(len(g["poses"]), g["fixed"], g["edge_from"], g["edge_to"])
print(len(g["poses"]), len(g["edge_from"]), {k: info[k] for k in ("fronts", "levels", "max_border", "max_children", "U_doubles")}, flush=True)
ctx = Context(0)
rc, p, chi = ctx.gn_optimize(*a, 3)
print("gpu ok", flush=True)
st, p2, chi2, _ = O.gn_optimize(*a, 3)
print("chi2 %.9g vs %.9g rel %.2e; max pose diff %.2e" % (chi[-1], chi2[-1], abs(chi[-1] - chi2[-1]) / chi2[-1], np.abs(p - p2).max()))
