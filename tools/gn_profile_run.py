"""Run optimize(10) on the C2 graph a few times (used under rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth, Context
ctx = Context(0)
g = synth.make_pose_graph(10000, 40000, seed=12345)
a = (g['poses'], g['fixed'], g['edge_from'], g['edge_to'], g['meas'], g['info'])
for r in range(3):
    rc, p, chi = ctx.gn_optimize(*a, 10)
print(chi[-1], ctx.gn_last_timing())
