#!/bin/bash
# Round 3, item 1: `python bench.py --gpus N` with no launcher in front (2 gloo ranks sharing the one GPU), and the
# RCCL init / all-reduce / barrier path with ONE rank (`--force-dist --backend nccl`).
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --share-device --steps 4 --warmup 1 --slices 2000 \
  > gpurun_out/r03_selflaunch_2.log 2>&1
echo "selflaunch rc=$?"; grep '^{' gpurun_out/r03_selflaunch_2.log | cut -c1-900
timeout -k 10 300 python bench.py --gpus 1 --force-dist --backend nccl --steps 4 --warmup 1 --slices 2000 --no-extras \
  --no-cpu-baseline > gpurun_out/r03_nccl_1.log 2>&1
echo "nccl1 rc=$?"; grep '^{' gpurun_out/r03_nccl_1.log | cut -c1-900; tail -3 gpurun_out/r03_nccl_1.log | cut -c1-300
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --share-device --workload cfg3 --steps 10 --warmup 3 \
  > gpurun_out/r03_selflaunch_cfg3.log 2>&1
echo "selflaunch cfg3 rc=$?"; grep '^{' gpurun_out/r03_selflaunch_cfg3.log | cut -c1-700
