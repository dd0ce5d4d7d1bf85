import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth
from cg_mrslam_amd._lib import gn_symbolic_info
g = synth.make_pose_graph()
best = None
for r in range(10):
    i = gn_symbolic_info(10000, g["fixed"], g["edge_from"], g["edge_to"])
    t = (i["order_us"], i["structure_us"])
    best = t if best is None or sum(t) < sum(best) else best
print(os.environ.get("CGMR_HOST_THREADS"), os.environ.get("CGMR_ND_SINGLE_SWEEP"), "order_us", best[0], "structure_us", best[1], "levels", i["levels"], "fronts", i["fronts"], "flops", i["factor_flops"], "maxborder", i["max_border"], "Udoubles", i["U_doubles"])
