#!/bin/bash
# SQ counter passes for the batched close matcher (run through gpurun from the repo root); every set is its own run
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/match_pmc
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/match_$i --output-format csv -- python $R/tools/match_profile_run.py 4096 > $O/match_$i.log 2>&1
done
cd $R
python tools/pmc_summarise.py $(find $O -name "*counter_collection.csv") > $O/pmc_summary.txt
grep -i "match_close" $O/pmc_summary.txt | head -40
