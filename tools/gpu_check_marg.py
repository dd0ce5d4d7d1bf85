import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth, Context
from oracle import oracle as O
ctx = Context(0)
for (V, E, nq) in [(200, 600, 5), (2000, 7000, 40), (5000, 20000, 100)]:
    g = synth.make_pose_graph(V, E, seed=9)
    a = (g["edge_from"], g["edge_to"], g["meas"], g["info"])
    rc, p, chi = ctx.gn_optimize(g["poses"], g["fixed"], *a, 8)
    query = np.linspace(1, V - 1, nq).astype(np.int32)
    t = time.time(); cov = ctx.marginals(p, g["fixed"], *a, query); tg = time.time() - t
    t = time.time(); st, cov2 = O.marginals(p, g["fixed"], *a, query); tc = time.time() - t
    print(V, "marginals gpu %.4f cpu %.4f" % (tg, tc), "rel err", np.abs(cov - cov2).max() / np.abs(cov2).max())
    gauge = int(query[len(query) // 2])
    t = time.time(); to, est, iu, cv = ctx.condense(p, *a, gauge, query); tg = time.time() - t
    t = time.time(); n, to2, est2, iu2, cv2 = O.condense(p, *a, gauge, query); tc = time.time() - t
    print(V, "condense gpu %.4f cpu %.4f" % (tg, tc), n, len(to), "to eq", np.array_equal(to, to2), "est", np.abs(est - est2).max(),
          "info rel", np.abs(iu - iu2).max() / np.abs(iu2).max(), "cov rel", np.abs(cv - cv2).max() / np.abs(cv2).max())
    ce = ctx.covariance_estimate(p, *a, V - 1, query[:10]); st, ce2 = O.covariance_estimate(p, *a, V - 1, query[:10])
    print(V, "cov estimate rel", np.abs(ce - ce2).max() / np.abs(ce2).max())
