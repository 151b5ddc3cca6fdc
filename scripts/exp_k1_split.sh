#!/bin/bash
# What the accumulation costs in the table backward: vjp_pwl_kernel as built vs without its accumulations
# (-DCNF_VJP_PWL_NO_ATOMICS: wrong gradients, timing only).  Rebuilds the library on the box (minimal configs).
mkdir -p gpurun_out
for v in "" "-DCNF_VJP_PWL_NO_ATOMICS"; do
  CNF_EXTRA_FLAGS="$v" python -m cnf_ot_amd.build --minimal --force > /dev/null 2>&1 || { echo "build failed: $v"; exit 1; }
  echo "== variant '${v:-as built}'"
  timeout -k 10 200 python scripts/exp_vjp_tables.py 2>&1 | grep "131072.*tables"
done
