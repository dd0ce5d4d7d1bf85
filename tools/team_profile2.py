"""cProfile of the C4 leg (four robots of the cg_mrslam node in one process): where the host time per key frame goes."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cg_mrslam_amd import Context
ctx = Context(0)
bench.team_leg(ctx)                      # warm-up (library load, first launches)
pr = cProfile.Profile()
pr.enable()
out = bench.team_leg(ctx)
pr.disable()
print(out["key_frames"], out["seconds"], out["key_frames_per_s"])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
