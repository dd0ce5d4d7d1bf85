"""Tree levels / launched levels / factorisation flops of the ordering over a set of graphs (host only), for one setting of the
environment (the switches are read once per process): python tools/nd_root_eval.py  -> one line per graph and the sums."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cg_mrslam_amd import synth
from cg_mrslam_amd._lib import gn_symbolic_info
tot_l = tot_ll = 0; tot_f = 0.0
for V in (5000, 10000, 20000):
    for seed in range(8):
        g = synth.make_pose_graph(V, 4 * V, seed=1000 + 31 * seed + V)
        i = gn_symbolic_info(V, g["fixed"], g["edge_from"], g["edge_to"])
        tot_l += i["levels"]; tot_ll += i["launch_levels"]; tot_f += i["factor_flops"]
        print(V, seed, i["levels"], i["launch_levels"], i["factor_flops"], i["order_us"])
g = synth.make_pose_graph(10000, 40000, seed=12345, strict=True)
i = gn_symbolic_info(10000, g["fixed"], g["edge_from"], g["edge_to"])
print("C2", i["levels"], i["launch_levels"], i["factor_flops"])
print("SUM levels %d launch_levels %d flops %.4g" % (tot_l, tot_ll, tot_f))
