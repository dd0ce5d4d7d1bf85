"""cProfile of the C4 leg: where the host side of the multi-robot key-frame loop spends its time."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cg_mrslam_amd import Context
ctx = Context(0)
bench.team_leg(ctx)
pr = cProfile.Profile()
pr.enable()
bench.team_leg(ctx)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(38)
