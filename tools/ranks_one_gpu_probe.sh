#!/bin/bash
# N ranks of bench.py on ONE GPU (gloo), reduced sizes: where a C5 round goes (phases of rank 0, time-outs) as the rank count grows -- round 4 found the
# 16-CPU quota of the GPU boxes with it (OMP_NUM_THREADS, helper pool size).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for n in 2 4 8; do
  echo "== $n ranks"
  CGMR_GN_TRACE=1 CGMR_BENCH_SINGLE_DEVICE=1 CGMR_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus $n --steps 2 --warmup 1 --vertices 2000 --edges 7000 --match-pairs 0 --c5-vertices 600 --c5-edges 2100 --c5-chunk 50 --no-cpu-baseline > /tmp/rp_$n.log 2>&1
  grep "^\[gn\]" /tmp/rp_$n.log | head -4
  python - <<P /tmp/rp_$n.log
import json, sys
ln = [l for l in open(sys.argv[1]) if l.startswith('{"metric"')]
d = json.loads(ln[-1]); e = d["exchange"]
print({k: e[k] for k in ("round_ms_mean_max", "solo_round_ms_mean_max", "backward_solve_timeouts_rank0", "ms_per_round_rank0")})
P
done
