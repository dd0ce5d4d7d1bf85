"""Run the key-frame driver (cg_mrslam_amd/slam.py) on a synthetic corridor loop on the GPU and print what it did."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cg_mrslam_amd import synth, Context
from cg_mrslam_amd.matcher import ScanMatcher, LCScanMatcher
from cg_mrslam_amd.slam import GraphSLAMDriver, run_srslam
T = int(sys.argv[1]) if len(sys.argv) > 1 else 240
laps = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
tr = synth.make_trajectory(T, laps=laps)
ctx = Context(0)
la = (tr["n_beams"], tr["angle_min"], tr["angle_inc"], tr["max_range"])
slam = GraphSLAMDriver(ctx, ScanMatcher(ctx, *la), LCScanMatcher(ctx, *la))
t0 = time.time()
n = run_srslam(slam, tr["odom"], tr["scans"])
dt = time.time() - t0
g = slam.g
kinds = {k: slam.edge_kind.count(k) for k in ("odom", "sm", "lc")}
print(f"{n} key frames, {g.n_edges} edges {kinds} in {dt:.1f} s ({dt / max(n, 1) * 1e3:.0f} ms per key frame)")
# error against the truth at the key frames: key frame k was taken at some step; compare via nearest truth pose
tp = tr["truth"]
err = [np.min(np.hypot(tp[:, 0] - p[0], tp[:, 1] - p[1])) for p in g.poses]
print("max distance of an estimated key frame from the true path: %.3f m" % max(err))
print("final chi2", None if slam.last_chi2 is None else float(slam.last_chi2[-1]))
lc = [l for l in slam.log if l[0] in ("lc", "lcc")]
print("loop-closure log:", lc[:6], "..."); print("lcc:", [l for l in slam.log if l[0] == "lcc"])
