#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root; ROUND=rNN names the files):
#   PMC passes first (each its own run, kernel-trace only) for the GN and matcher kernels, condensed on the box into
#   profiles/rNN_pmc_*.json so that the profiled bench run that follows reads counters of the sources it runs; then
#   kernel-trace + stats of bench.py.  tools/make_profile_summary.py rNN turns the merged gpurun_out/prof_round into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
RND=${ROUND:-r04}
O=$R/gpurun_out/prof_round
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/gn_$n --output-format csv -- python $R/tools/gn_profile_run.py > $O/gn_$n.log 2>&1
done
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/match_$i --output-format csv -- python $R/tools/match_profile_run.py 4096 > $O/match_$i.log 2>&1
done
# FETCH_SIZE / WRITE_SIZE against known byte counts, matcher phase cycles (timing build, if it was shipped)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --kernel-trace --pmc $c -d $O/calib_$c --output-format csv -- $R/tools/ubench/fetch_calib_ubench > $O/calib_$c.log 2>&1
done
# kernel trace + stats of the C2 solve ALONE (no counters): the per-kernel averages bench.py's roofline is checked against
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/gn_c2 --output-format csv -- python $R/tools/gn_profile_run.py > $O/gn_c2.log 2>&1
find $O/gn_c2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/gn_c2_kernel_stats.csv
find $O/gn_c2 -name "*kernel_trace.csv" -delete
cd $R
[ -f cg_mrslam_amd/libcgmr_t.so ] && CGMR_MATCH_SPLIT=1 CGMR_LIB=cg_mrslam_amd/libcgmr_t.so timeout 120 python tools/gpu_mphase.py $O/match_phases.json > $O/match_phases.log 2>&1
python tools/make_profile_summary.py $RND --pmc-only > $O/pmc_only.log 2>&1
cd /tmp
# (one retry: the profiled run of the bench -- its C4 leg drives four contexts from four threads -- dies with SIGSEGV inside
# librocprofiler-sdk's queue interceptor in about one profiled process in twelve, never without the profiler:
# profiles/r06_profiler_crash.md, tools/stress/team_stress.sh)
for try in 1 2; do
  rm -rf $O/bench
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/bench --output-format csv -- python $R/bench.py > $O/bench.log 2>&1
  grep -q "^{\"metric\"" $O/bench.log && break
  cp $O/bench.log $O/bench_failed_try$try.log
done
grep "^{\"metric\"" $O/bench.log | tail -1 > $O/bench_line.json
cd $R
python tools/pmc_summarise.py $(find $O -name "*counter_collection.csv") > $O/pmc_summary.txt
find $O/bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
# the raw trace of the bench run is tens of MB (every launch of every leg) and only its statistics are used: what comes back
# through gpurun_out/ is capped at 64 MiB
find $O/bench -name "*kernel_trace.csv" -delete
head -40 $O/pmc_summary.txt
