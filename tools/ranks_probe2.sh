#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for q in 2 1; do
  echo "== 8 ranks, GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q CGMR_GN_TRACE=1 CGMR_BENCH_SINGLE_DEVICE=1 CGMR_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 8 --steps 2 --warmup 1 --vertices 2000 --edges 7000 --match-pairs 0 --c5-vertices 600 --c5-edges 2100 --c5-chunk 50 --no-cpu-baseline > /tmp/rp_$q.log 2>&1
  grep "^\[gn\]" /tmp/rp_$q.log | head -2
  python - <<P /tmp/rp_$q.log
import json, sys
ln = [l for l in open(sys.argv[1]) if l.startswith('{"metric"')]
d = json.loads(ln[-1]); e = d["exchange"]
print({k: e[k] for k in ("round_ms_mean_max", "solo_round_ms_mean_max", "backward_solve_timeouts_rank0", "ms_per_round_rank0")})
P
done
echo "== 8 ranks, helpers not spinning (CGMR_HOST_SPIN_US=200), default queues"
CGMR_HOST_SPIN_US=200 CGMR_GN_TRACE=1 CGMR_BENCH_SINGLE_DEVICE=1 CGMR_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 8 --steps 2 --warmup 1 --vertices 2000 --edges 7000 --match-pairs 0 --c5-vertices 600 --c5-edges 2100 --c5-chunk 50 --no-cpu-baseline > /tmp/rp_s.log 2>&1
grep "^\[gn\]" /tmp/rp_s.log | head -2
python - <<P /tmp/rp_s.log
import json, sys
ln = [l for l in open(sys.argv[1]) if l.startswith('{"metric"')]
d = json.loads(ln[-1]); e = d["exchange"]
print({k: e[k] for k in ("round_ms_mean_max", "solo_round_ms_mean_max", "backward_solve_timeouts_rank0", "ms_per_round_rank0")})
P
