#!/bin/bash
# Rehearsal of the multi-rank bench path on a ONE-GPU box: 2 ranks share cuda:0, gloo for the barrier/all-reduce.
mkdir -p gpurun_out
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 1024 --warmup 128 --backend gloo --share-device > gpurun_out/bench_2rank_rehearsal.log 2>&1
echo "rehearsal rc=$?"; tail -3 gpurun_out/bench_2rank_rehearsal.log | cut -c1-600
