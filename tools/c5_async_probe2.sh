#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/${1:-probe}; mkdir -p $O
python tools/c5_loopback_time.py 8 100 sync > $O/loop_sync.txt 2>&1; tail -n 8 $O/loop_sync.txt
python tools/c5_loopback_time.py 8 100 > $O/loop_async.txt 2>&1; tail -n 8 $O/loop_async.txt
GPU_MAX_HW_QUEUES=16 python tools/c5_loopback_time.py 8 100 > $O/loop_async_q16.txt 2>&1; tail -n 8 $O/loop_async_q16.txt
python tools/c5_loopback_time.py 1 100 > $O/solo_async.txt 2>&1; tail -n 8 $O/solo_async.txt
CGMR_COND_TRACE=1 python tools/c5_loopback_time.py 8 60 2>&1 | grep "queued on the side" | tail -n 32 | cut -c1-260
