#!/bin/bash
# round-4 probe: what the asynchronous condensed graphs cost the host in the one-GPU loopback
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/${1:-probe}; mkdir -p $O
python tools/c5_loopback_time.py 1 100 sync > $O/solo_sync.txt 2>&1; tail -n 8 $O/solo_sync.txt
python tools/c5_loopback_time.py 1 100 > $O/solo_async.txt 2>&1; tail -n 8 $O/solo_async.txt
CGMR_HOST_SPIN_US=10000 python tools/c5_loopback_time.py 1 100 sync > $O/solo_sync_spin.txt 2>&1; tail -n 8 $O/solo_sync_spin.txt
python tools/c5_loopback_time.py 8 100 sync > $O/loop_sync.txt 2>&1; tail -n 8 $O/loop_sync.txt
python tools/c5_loopback_time.py 8 100 > $O/loop_async.txt 2>&1; tail -n 8 $O/loop_async.txt
CGMR_HOST_SPIN_US=10000 python tools/c5_loopback_time.py 8 100 > $O/loop_async_spin.txt 2>&1; tail -n 8 $O/loop_async_spin.txt
CGMR_COND_TRACE=1 python tools/c5_loopback_time.py 8 40 2>&1 | grep "queued on the side" | tail -n 24
