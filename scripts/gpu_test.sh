#!/bin/bash
# GPU tests only: scripts/gpu_test.sh [pytest args]
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -s "$@" > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest rc=$rc"; grep -E "^\[|passed|failed|FAILED|Error" gpurun_out/pytest_gpu.log | tail -70
exit 0
