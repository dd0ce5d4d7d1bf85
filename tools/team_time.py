"""Wall time of the C4 leg (four robots of the cg_mrslam node on one GPU), three runs; argv[1] = 1: a context and a thread per
robot instead of one after the other on one context."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cg_mrslam_amd import Context
ctx = Context(0)
conc = len(sys.argv) > 1 and sys.argv[1] == "1"
for _ in range(3):
    t = bench.team_leg(ctx, concurrent=conc)
    print(t["key_frames"], t["seconds"], t["key_frames_per_s"], t["inter_robot_edges"], t["condensed_edges_held"], t["max_distance_to_true_path_m"])
