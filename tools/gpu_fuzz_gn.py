"""Randomised GPU-vs-oracle check of the GN path over many graph sizes / densities / fixed sets (argv: count [seed0])."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth, Context
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ctx = Context(0)
rng = np.random.default_rng(seed0)
worst = 0.0
for k in range(n):
    V = int(rng.choice([30, 90, 250, 700, 1500, 3000, 6000]))
    E = int(V * rng.uniform(1.2, 5.0))
    g = synth.make_pose_graph(V, E, seed=seed0 + k)
    fixed = g["fixed"].copy()
    for v in rng.integers(0, V, size=int(rng.integers(0, 4))): fixed[v] = 1
    ef, et, meas, info = g["edge_from"], g["edge_to"], g["meas"], g["info"]
    if k % 5 == 0:                                     # duplicate some edges
        d = rng.integers(0, len(ef), size=max(1, len(ef) // 20))
        ef, et, meas, info = np.concatenate([ef, ef[d]]), np.concatenate([et, et[d]]), np.concatenate([meas, meas[d]]), np.concatenate([info, info[d]])
    # start near the optimum (ground truth + small noise) so that rounding differences are not amplified by a diverging GN
    p0 = g["truth"] + rng.normal(0, 0.02, g["truth"].shape) if "truth" in g else g["poses"]
    a = (p0, fixed, ef, et, meas, info)
    rc, p, chi = ctx.gn_optimize(*a, 4, raise_on_cholesky=False)
    st, p2, chi2, _ = O.gn_optimize(*a, 4)
    rel = abs(chi[-1] - chi2[-1]) / max(chi2[-1], 1e-300)
    dp = np.abs(p - p2).max()
    worst = max(worst, rel)
    flag = "" if (rc == st == 0 and rel < 1e-8 and dp < 1e-6) else "   <-- MISMATCH"
    print(f"{k:3d} V {V:5d} E {len(ef):6d} fixed {int(fixed.sum()):2d} rc {rc} st {st} chi2 {chi[-1]:.6e} rel {rel:.1e} dpose {dp:.1e}{flag}", flush=True)
print("worst relative chi2 difference", worst)
