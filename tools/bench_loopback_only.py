"""bench.py's C5 loopback leg alone (the same function, the same opt-ins): python tools/bench_loopback_only.py [rounds]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]] + (["--c5-rounds", sys.argv[1]] if len(sys.argv) > 1 else [])
import bench
args = bench.parse()
r = bench.loopback_leg(args)
print(json.dumps({k: r[k] for k in ("ms_per_robot_and_round", "round_ms_per_robot", "failed_condensed_batches")}), r["solo_round_ms_same_robots"]["mean"])
