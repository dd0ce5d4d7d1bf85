import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from cg_mrslam_amd import Context, synth
from oracle import oracle as O
ctx = Context(0)
for V, E, seed in ((600, 2000, 1), (3000, 11000, 2), (10000, 40000, 12345)):
    g = synth.make_pose_graph(V, E, seed=seed)
    a = (g["poses"], g["fixed"], g["edge_from"], g["edge_to"], g["meas"], g["info"])
    rc, p, chi = ctx.gn_optimize(*a, 6)
    st, p2, chi2, _ = O.gn_optimize(*a, 6)
    print(V, rc, st, "chi rel", np.max(np.abs(chi - chi2) / np.maximum(chi2, 1e-30)), "pose", np.abs(p - p2).max())
