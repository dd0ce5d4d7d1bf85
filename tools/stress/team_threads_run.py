"""Diagnostic: the C4 leg with a context and a thread per robot, REPS times in one process (argv[1], default 10), with a
native + Python backtrace on SIGSEGV (tools/stress/segv_bt.c).  Run under rocprofv3 by tools/stress/team_stress.sh: the crash
DESIGN.md section 7 records was seen only there."""
import ctypes
import faulthandler
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
faulthandler.enable(all_threads=True)
so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "segv_bt.so")
if os.path.exists(so):
    ctypes.CDLL(so).segv_bt_install()

import bench  # noqa: E402
from cg_mrslam_amd import Context  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = Context(0)
seq = bench.team_leg(ctx)
for k in range(reps):
    out = bench.team_leg(ctx, concurrent=True)
    same = all(out[q] == seq[q] for q in ("key_frames", "messages_delivered", "bytes_sent", "inter_robot_edges", "condensed_edges_held"))
    print(k, out["key_frames_per_s"], "same" if same else "DIFFERENT", flush=True)
print("__TEAM_STRESS_OK__", flush=True)
