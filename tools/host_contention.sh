#!/bin/bash
# The host analysis next to somebody else's load: N busy processes (default 192 of the box's 256 hardware threads) spin for
# the duration; the cold optimize(10) breakdown with and without the waiting thread taking over queued jobs.
N=${1:-192}
python - <<PY &
import multiprocessing as mp, time
def spin(t):
    e = time.time() + t
    while time.time() < e: pass
ps = [mp.Process(target=spin, args=(40,)) for _ in range($N)]
[p.start() for p in ps]; [p.join() for p in ps]
PY
HOG=$!
sleep 3
echo "with $N busy processes:"
for i in 1 2; do python tools/gn_breakdown.py; done
echo "CGMR_HOST_STEAL=0:"
for i in 1 2; do CGMR_HOST_STEAL=0 python tools/gn_breakdown.py; done
echo "CGMR_HOST_PIN=0:"
for i in 1 2; do CGMR_HOST_PIN=0 python tools/gn_breakdown.py; done
wait $HOG
echo "idle box:"
python tools/gn_breakdown.py
