#!/bin/bash
# Rehearsal of the multi-rank bench paths on a ONE-GPU box: 2 (then 4) ranks share cuda:0, gloo for the
# barrier / all-reduce.  Plumbing evidence (sharding, the one all-reduce, the JSON line), NOT a scaling curve.
mkdir -p gpurun_out
run() {  # run <ranks> <logname> <bench args...>
  local n=$1 log=$2; shift 2
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus $n --backend gloo --share-device "$@" > gpurun_out/$log 2>&1
  echo "$log rc=$?"; grep '^{' gpurun_out/$log | cut -c1-700
}
run 2 rehearse_2_sample.log --steps 4 --warmup 1 --slices 2000
run 4 rehearse_4_sample.log --steps 4 --warmup 1 --slices 2000
run 2 rehearse_2_cfg3.log --workload cfg3 --steps 20 --warmup 5
run 2 rehearse_2_cfg4.log --workload cfg4 --steps 10 --warmup 3
python bench.py --workload cfg3 --steps 20 --warmup 5 > gpurun_out/rehearse_1_cfg3.log 2>&1; grep '^{' gpurun_out/rehearse_1_cfg3.log | cut -c1-400
python bench.py --workload cfg4 --steps 10 --warmup 3 > gpurun_out/rehearse_1_cfg4.log 2>&1; grep '^{' gpurun_out/rehearse_1_cfg4.log | cut -c1-400
