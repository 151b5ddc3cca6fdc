#!/bin/bash
# scripts/gpu_pytest.sh <pytest args>: run pytest on the GPU box, log to gpurun_out/
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest "$@" > gpurun_out/pytest_sel.log 2>&1
rc=$?; echo "pytest rc=$rc"; grep -E "^\[|passed|failed|FAILED|Error|error|assert" gpurun_out/pytest_sel.log | tail -60
exit 0
