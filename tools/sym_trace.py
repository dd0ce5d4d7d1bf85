import sys,os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cg_mrslam_amd import synth, _lib
g = synth.make_pose_graph(10000, 40000, seed=12345, strict=True)
for k in range(5):
    info=_lib.gn_symbolic_info(10000, g["fixed"], g["edge_from"], g["edge_to"])
    print("order_us", info["order_us"], "structure_us", info["structure_us"], file=sys.stderr)
