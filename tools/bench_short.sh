#!/bin/bash
# one short GN-only bench line condensed to: value ms_per_step host_ms device_ms (helper for A/B runs through gpurun)
python bench.py --match-pairs 0 --no-cpu-baseline --steps ${1:-30} | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['host_symbolic_ms_per_step'], d['device_ms_per_step'], d['chi2_final'])"
