#!/bin/bash
# per-launch durations of the last GN iteration of the C2 solve (rocprofv3 kernel trace -> tools/gn_level_trace.py); env passes through (CGMR_FUSE=0, ..)
R=$(pwd); O=/tmp/gntrace_$$; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O --output-format csv -- python $R/tools/gn_profile_run.py > $O/log 2>&1
find $O -name "*kernel_trace.csv" | head -1 | xargs python $R/tools/gn_level_trace.py
