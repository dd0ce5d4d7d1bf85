#!/bin/bash
# Diagnostic (run through gpurun from the repo root): PROCS processes (default 30) of tools/stress/team_threads_run.py with REPS
# threaded C4 legs each (default 10), every process under `rocprofv3 --kernel-trace --stats` unless PROFILER=0.  Counts the
# processes that did not finish and keeps their logs under gpurun_out/team_stress/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/team_stress
rm -rf $O; mkdir -p $O
cc -O1 -g -shared -fPIC -o $R/tools/stress/segv_bt.so $R/tools/stress/segv_bt.c
PROCS=${PROCS:-30}; REPS=${REPS:-10}
cd /tmp; export TMPDIR=/tmp
bad=0
for i in $(seq 1 $PROCS); do
  rm -rf /tmp/ts_prof
  if [ "${PROFILER:-1}" = "1" ]; then
    timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ts_prof --output-format csv -- python $R/tools/stress/team_threads_run.py $REPS > $O/run_$i.log 2>&1
  else
    timeout 300 python $R/tools/stress/team_threads_run.py $REPS > $O/run_$i.log 2>&1
  fi
  rc=$?
  if grep -q __TEAM_STRESS_OK__ $O/run_$i.log; then rm -f $O/run_$i.log; else bad=$((bad+1)); echo "process $i: rc $rc" >> $O/failed.txt; fi
done
echo "team_stress: $bad of $PROCS processes failed (REPS=$REPS, PROFILER=${PROFILER:-1}, CGMR_HIER_HOST=${CGMR_HIER_HOST:-0})" | tee $O/summary.txt
