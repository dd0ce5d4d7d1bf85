#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/${1:-probe}; mkdir -p $O
CGMR_GN_TRACE=1 python tools/c5_loopback_time.py 8 100 sync > $O/loop_sync.txt 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" $O/loop_sync.txt | tail -n 17
CGMR_GN_TRACE=1 python tools/c5_loopback_time.py 8 100 > $O/loop_async.txt 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" $O/loop_async.txt | tail -n 17
