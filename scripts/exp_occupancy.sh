#!/bin/bash
# Occupancy sweep of flow_pwl_kernel (the driver's bench, no extras): workgroup size x workgroups per CU.
# The kernel uses 96 registers: up to 5 waves per SIMD fit.
mkdir -p gpurun_out
run() { # threads blocks_per_cu min_lds_kb
  CNF_PWL_THREADS=$1 CNF_PWL_BLOCKS_PER_CU=$2 CNF_PWL_MIN_LDS_KB=$3 timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/occ_$1_$2.log 2>&1
  python3 - <<PY
import json
for l in open("gpurun_out/occ_$1_$2.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print("threads %4d x %d per CU (lds %3d KB): %.2f G/s  launch_ms %.4f  path %s" % ($1, $2, $3, d["value"]/1e9, r["launch_ms"], r["path"]))
PY
}
run 1024 1 82
export CNF_PWL_WINDOW=1     # 128-row LDS window (47 KB per workgroup) so that up to 3 workgroups fit a CU
run 1024 1 82
run 512 2 70
run 512 3 50
run 640 2 70
run 320 3 50
run 384 3 50
run 448 2 70
