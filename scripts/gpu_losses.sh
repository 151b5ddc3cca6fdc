#!/bin/bash
# loss / gradient tests + the three BASELINE configs' loss and value_and_grad times
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_losses.py tests/test_gpu_grad.py -m gpu -q -x > gpurun_out/pytest_losses.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/pytest_losses.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
for c in cfg3 cfg4 cfg5; do python scripts/prof_cfg.py $c 20 2>&1 | grep "ms per call"; done
