import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from cg_mrslam_amd import synth, Context
from cg_mrslam_amd._lib import gn_symbolic_info
from oracle import oracle as O
ctx = Context(0)
for V, E in ((50000, 200000), (100000, 300000)):
    g = synth.make_pose_graph(V, E, seed=7)
    a = (g["poses"], g["fixed"], g["edge_from"], g["edge_to"], g["meas"], g["info"])
    info = gn_symbolic_info(V, g["fixed"], g["edge_from"], g["edge_to"])
    t = time.time(); rc, p, chi = ctx.gn_optimize(*a, 10); tg = time.time() - t
    t = time.time(); rc, p, chi = ctx.gn_optimize(*a, 10); tg = time.time() - t
    t = time.time(); st, p2, chi2, _ = O.gn_optimize(*a, 10); tc = time.time() - t
    print(V, E, {k: info[k] for k in ("fronts", "levels", "max_border", "max_children")}, "gpu s %.3f" % tg, ctx.gn_last_timing(), "cpu s %.2f" % tc,
          "chi2", chi[-1], chi2[-1], "rel", abs(chi[-1] - chi2[-1]) / chi2[-1], "dpose", np.abs(p - p2).max())
