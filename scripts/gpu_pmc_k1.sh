#!/bin/bash
# PMC counters of the table backward's kernel (vjp_pwl_kernel) over scripts/exp_vjp_tables.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_k1
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-30)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/$name -- python3 $R/scripts/exp_vjp_tables.py > $OUT/$name.log 2>&1
  rc=$?; echo "pmc $name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("$OUT/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        if "vjp_pwl_kernel" in r["Kernel_Name"]:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(k)
    for n, v in sorted(c.items()):
        print("   %-24s mean %.5g  max %.5g  (n=%d)" % (n, sum(v) / len(v), max(v), len(v)))
PY
