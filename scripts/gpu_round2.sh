#!/bin/bash
# round-2 GPU pass: the driver's bench command first (it is the top item), then the parity tests.
# Stops after a timed-out/killed step.  scripts/gpu_round2.sh [tag]
TAG=${1:-a}
mkdir -p gpurun_out
timeout -k 10 420 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_$TAG.log 2>&1
rc=$?; echo "bench rc=$rc"; tail -c 3000 gpurun_out/bench_$TAG.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_$TAG.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/pytest_$TAG.log
exit 0
