#!/bin/bash
# the randomised soak of the table path (scripts/soak_pwl.py): 4 seeds x 60 cases at L = 2, one seed at L = 3 and L = 4
mkdir -p gpurun_out
for seed in 50 51 52 53; do
  timeout -k 10 280 python scripts/soak_pwl.py $seed 60 2 > gpurun_out/soak_r02b_$seed.log 2>&1; echo "seed $seed rc=$?"; tail -1 gpurun_out/soak_r02b_$seed.log
done
timeout -k 10 280 python scripts/soak_pwl.py 54 40 3 > gpurun_out/soak_r02b_L3.log 2>&1; echo "L3 rc=$?"; tail -1 gpurun_out/soak_r02b_L3.log
timeout -k 10 280 python scripts/soak_pwl.py 55 40 4 > gpurun_out/soak_r02b_L4.log 2>&1; echo "L4 rc=$?"; tail -1 gpurun_out/soak_r02b_L4.log
grep -h "WORSE" gpurun_out/soak_r02b_*.log | cut -c1-400 | head -20
