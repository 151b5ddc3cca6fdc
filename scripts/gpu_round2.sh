#!/bin/bash
# round-2 GPU pass: parity tests (all, no -x), then the driver's bench command.  Stops after a timed-out/killed step.
# scripts/gpu_round2.sh [tag] [pytest args...]
TAG=${1:-a}; shift
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q "$@" > gpurun_out/pytest_$TAG.log 2>&1
rc=$?; echo "pytest rc=$rc"; grep -E "^\[|passed|failed|FAILED|rror" gpurun_out/pytest_$TAG.log | tail -60
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 420 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_$TAG.log 2>&1
rc=$?; echo "bench rc=$rc"; tail -c 1500 gpurun_out/bench_$TAG.log
exit 0
