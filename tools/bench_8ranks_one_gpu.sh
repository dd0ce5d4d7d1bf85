#!/bin/bash
# bench.py --gpus 8 at full size on ONE GPU: all ranks share cuda:0, collectives over gloo (everything of the 8-rank run but
# xGMI).  The line goes to gpurun_out/<dir>/bench_8ranks_one_gpu.json; copy it to profiles/rNN_bench_8ranks_one_gpu.json.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/${1:-ranks8}; mkdir -p $O
CGMR_BENCH_SINGLE_DEVICE=1 CGMR_BENCH_BACKEND=gloo timeout 1500 python bench.py --gpus 8 --steps 10 --warmup 3 --match-pairs 0 --no-cpu-baseline > $O/bench_8ranks.log 2>&1
grep "^{\"metric\"" $O/bench_8ranks.log | tail -1 > $O/bench_8ranks_one_gpu.json
python - <<'P' $O/bench_8ranks_one_gpu.json
import json, sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], "ms/step", d["ms_per_step"], "host_sym", d["host_symbolic_ms_per_step"], "dev", d["device_ms_per_step"])
print("exchange", {k: d["exchange"][k] for k in ("robots", "rounds", "transport", "round_ms_mean_max", "solo_round_ms_mean_max", "weak_scaling_efficiency_vs_solo", "optimize5_ms_mean_max", "condense_ms_mean_max")})
for q in d["host_pool_of_every_rank"]: print(" ", q)
P
