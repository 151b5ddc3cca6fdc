#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trace_step_$1
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scripts/prof_step.py $1 > $OUT/trace.log 2>&1
echo "rc=$?"
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats.csv && python3 - $OUT/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); reps = 203
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time per step: {tot / reps / 1e3:.1f} us")
for r in rows[:16]:
  print(f'{int(r["Calls"]) / reps:6.2f} x {float(r["AverageNs"]) / 1e3:8.1f} us = {int(r["TotalDurationNs"]) / reps / 1e3:8.1f} us/step  {r["Name"][:100]}')
PY
