#!/bin/bash
# A/B: the backward conditioner's two A operands fetched at the top of the function (pinned there by a scheduling
# barrier) vs where the compiler places them.  Rebuilds the library on the box (minimal configs) per variant.
mkdir -p gpurun_out
for v in "" "-DCNF_BWD_HOIST_A"; do
  CNF_EXTRA_FLAGS="$v" python -m cnf_ot_amd.build --minimal --force > /dev/null 2>&1 || { echo "build failed: $v"; exit 1; }
  echo "== variant '${v:-compiler-placed}'"
  for c in cfg3 cfg4 cfg5; do python scripts/prof_cfg.py $c 10 2>&1 | grep "value_and_grad"; done
done
