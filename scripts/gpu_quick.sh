#!/bin/bash
# quick A/B on the metric kernel: selected parity tests (minimal build is enough) + the headline without extras
# scripts/gpu_quick.sh <tag> [pytest -k expression]
TAG=${1:-q}; K=${2:-"piecewise or table_path or table_rows or config2 or per_sample or never_allocate or capturable or wild"}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "$K" > gpurun_out/pytest_$TAG.log 2>&1
rc=$?; echo "pytest rc=$rc"; grep -E "^\[pwl|passed|failed|FAILED|rror" gpurun_out/pytest_$TAG.log | tail -30
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/bench_$TAG.log 2>&1
rc=$?; echo "bench rc=$rc"; python3 - <<PY
import json
for l in open("gpurun_out/bench_$TAG.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print("value %.2f G/s  frac %.4f  launch_ms %.4f  build_ms %.4f  path %s" % (d["value"]/1e9, r["frac"], r["launch_ms"], r["table_build_ms_per_launch"], r["path"]))
PY
exit 0
