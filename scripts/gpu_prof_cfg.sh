#!/bin/bash
# rocprofv3 kernel stats of one BASELINE config's loss + value_and_grad: scripts/gpu_prof_cfg.sh cfg4 [tag]
CFG=${1:-cfg4}; TAG=${2:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_$CFG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scripts/prof_cfg.py $CFG 20 > $OUT/trace.log 2>&1
rc=$?; echo "trace rc=$rc"; tail -3 $OUT/trace.log
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -25 $f
exit 0
