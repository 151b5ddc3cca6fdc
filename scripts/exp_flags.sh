#!/bin/bash
# A/B of compile-time experiment switches on the driver's bench (no extras).  Rebuilds the library on the box
# (minimal configs) per variant:  scripts/exp_flags.sh "" "-DCNF_PWL_NO_PREFETCH" ...
mkdir -p gpurun_out
for v in "$@"; do
  CNF_EXTRA_FLAGS="$v" python -m cnf_ot_amd.build --minimal --force > /dev/null 2>&1 || { echo "build failed: $v"; exit 1; }
  for rep in 1 2; do
    timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/flags.log 2>&1
    python3 - "$v" <<'PY'
import json, sys
for l in open("gpurun_out/flags.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print("%-28s %.2f G/s  launch_ms %.4f  build_ms %.4f" % (sys.argv[1] or "(default)", d["value"]/1e9, r["launch_ms"], r["table_build_ms_per_launch"]))
PY
  done
done
