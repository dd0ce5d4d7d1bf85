import sys, os
sys.path.insert(0, os.getcwd())
from cg_mrslam_amd import synth
from cg_mrslam_amd._lib import gn_symbolic_info
g = synth.make_pose_graph(10000, 40000, seed=12345, strict=True)
for _ in range(6):
    i = gn_symbolic_info(10000, g['fixed'], g['edge_from'], g['edge_to'])
    sys.stderr.write("== order %d structure %d\n" % (i['order_us'], i['structure_us']))
