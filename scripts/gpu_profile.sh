#!/bin/bash
# rocprofv3 passes over bench.py: (1) kernel trace + stats of the DRIVER'S command, (2..) PMC counters, each in its
# own run (no trace domains other than --kernel-trace next to --pmc).  Usage: scripts/gpu_profile.sh <tag>
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/trace.log 2>&1
rc=$?; echo "trace rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
grep '^{' $OUT/trace.log > $OUT/bench_line.json
# the same without the per_call / config / evaluator sections (which launch the metric kernel at other sizes and pull
# its average away from the timed region's): this table's average for the metric kernel is the one to compare with
# the bench line's roofline.launch_ms
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_metric -- $BENCH > $OUT/trace_metric.log 2>&1
rc=$?; echo "trace_metric rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_$name -- $BENCH > $OUT/pmc_$name.log 2>&1
  rc=$?; echo "pmc $name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 $f | cut -c1-200
