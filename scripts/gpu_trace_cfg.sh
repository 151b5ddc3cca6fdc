#!/bin/bash
# rocprofv3 kernel trace + stats of one BASELINE config's value_and_grad (or loss) only: scripts/gpu_trace_cfg.sh cfg5 r03g [reps] [vg|loss]
CFG=${1:-cfg5}; TAG=${2:-r03}; REPS=${3:-20}; WHAT=${4:-vg}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trace_${TAG}_$CFG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scripts/prof_cfg.py $CFG $REPS $WHAT > $OUT/trace.log 2>&1
rc=$?; echo "trace rc=$rc"; tail -2 $OUT/trace.log
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats.csv && python3 - $OUT/kernel_stats.csv $REPS <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); reps = int(sys.argv[2]) + 1
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time per call: {tot / reps / 1e6:.3f} ms over {reps} calls")
for r in rows[:24]:
  print(f'{int(r["Calls"]) / reps:6.2f} x {float(r["AverageNs"]) / 1e3:8.1f} us = {int(r["TotalDurationNs"]) / reps / 1e3:8.1f} us/call  {r["Name"][:110]}')
PY
