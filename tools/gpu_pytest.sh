#!/bin/bash
# GPU test suite on the gpurun box: prints the pytest summary (and the failures), nothing of the RCCL banner.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout ${TMO:-1000} python -m pytest tests -q -m gpu "$@" > gpurun_out/pytest_gpu.log 2>&1
grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -5
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu.log | head -20
