"""Wall time of the C4 leg (four robots of the cg_mrslam node on one GPU), three runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cg_mrslam_amd import Context
ctx = Context(0)
for _ in range(3):
    t = bench.team_leg(ctx)
    print(t["key_frames"], t["seconds"], t["key_frames_per_s"], t["inter_robot_edges"], t["condensed_edges_held"])
