#!/bin/bash
# pwl_build_kernel compiled for 4 (126 registers, two workgroups per CU) vs 6 waves per SIMD (80 registers + spills,
# three workgroups per CU): table build time per launch and the headline.  Rebuilds the library on the box.
mkdir -p gpurun_out
for w in 4 6; do
  CNF_EXTRA_FLAGS="-DPWL_BUILD_WAVES=$w" python -m cnf_ot_amd.build --minimal --force > /dev/null 2>&1 || { echo "build failed: $w"; exit 1; }
  echo "== PWL_BUILD_WAVES=$w"
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('value %.2f G/s launch_ms %.4f build_ms %.4f'%(d['value']/1e9, r['launch_ms'], r['table_build_ms_per_launch']))"
done
