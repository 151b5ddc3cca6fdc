#!/bin/bash
# Round 3: a changed build -> the whole GPU suite, then the three configs' step times
mkdir -p gpurun_out
TAG=${1:-x}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest_$TAG.log 2>&1
rc=$?; tail -5 gpurun_out/r03_pytest_$TAG.log; [ $rc -ne 0 ] && exit $rc
for c in cfg3 cfg4 cfg5; do timeout -k 10 200 python scripts/prof_cfg.py $c 20 2>&1 | grep "ms per call"; done | tee gpurun_out/r03_cfg_times_$TAG.txt
