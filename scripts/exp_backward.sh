#!/bin/bash
# A/B of the backward kernels' conditioner: MFMA (2 waves/SIMD, register-capped) vs MFMA with 512 registers
# (1 wave/SIMD) vs the vector-ALU form.  Rebuilds the library on the box (minimal configs) for each variant.
mkdir -p gpurun_out
for v in "" "-DCNF_BWD_OCC1" "-DCNF_BWD_VALU"; do
  CNF_EXTRA_FLAGS="$v" python -m cnf_ot_amd.build --minimal --force > /dev/null 2>&1 || { echo "build failed: $v"; exit 1; }
  echo "== variant '${v:-mfma, capped 256 regs}'"
  for c in cfg3 cfg4 cfg5; do python scripts/prof_cfg.py $c 10 2>&1 | grep "value_and_grad"; done
done
CNF_EXTRA_FLAGS="" python -m cnf_ot_amd.build --minimal --force > /dev/null 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_grad.py -m gpu -q -x -s -k "not other_dims" > gpurun_out/pytest_bwd.log 2>&1
echo "pytest rc=$?"; grep -E "^\[|passed|failed|FAILED|rror" gpurun_out/pytest_bwd.log | tail -30
python scripts/exp_host_profile.py cfg5 2>&1 | head -45
