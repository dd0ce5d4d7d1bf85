#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  "$@" > /tmp/rp.log 2>&1
  python - <<P /tmp/rp.log
import json, sys
ln = [l for l in open(sys.argv[1]) if l.startswith('{"metric"')]
d = json.loads(ln[-1]); e = d["exchange"]
print({k: e[k] for k in ("round_ms_mean_max", "solo_round_ms_mean_max", "weak_scaling_efficiency_vs_solo", "ms_per_round_rank0")})
P
  grep thrott /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo
}
A="--gpus 8 --steps 2 --warmup 1 --vertices 2000 --edges 7000 --match-pairs 0 --c5-vertices 600 --c5-edges 2100 --c5-chunk 50 --no-cpu-baseline"
echo "== OMP_NUM_THREADS=1"
run env OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 CGMR_BENCH_SINGLE_DEVICE=1 CGMR_BENCH_BACKEND=gloo timeout 600 python bench.py $A
echo "== OMP_NUM_THREADS=1 + blocking sync (HIP)"
run env OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 HIP_LAUNCH_BLOCKING=0 CGMR_HOST_PIN=0 CGMR_BENCH_SINGLE_DEVICE=1 CGMR_BENCH_BACKEND=gloo timeout 600 python bench.py $A
