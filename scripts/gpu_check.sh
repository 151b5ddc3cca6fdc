#!/bin/bash
# first GPU pass: parity tests, smoke, bench. Stops after a timed-out/killed step.
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -s > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -30 gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
rc=$?; echo "smoke rc=$rc"; tail -5 gpurun_out/smoke.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 400 python bench.py > gpurun_out/bench.log 2>&1
rc=$?; echo "bench rc=$rc"; tail -5 gpurun_out/bench.log
exit 0
