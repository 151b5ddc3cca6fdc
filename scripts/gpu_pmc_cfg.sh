#!/bin/bash
# rocprofv3 kernel stats + PMC counters (each counter set in its own run, only --kernel-trace next to --pmc) of one
# BASELINE config's loss + value_and_grad: scripts/gpu_pmc_cfg.sh cfg4 r03a   ->  gpurun_out/prof_<tag>_<cfg>/ and
# (collected) profiles/<tag>_<cfg>/{kernel_stats.csv, pmc_summary.json}
CFG=${1:-cfg4}; TAG=${2:-r03}; REPS=${3:-10}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_$CFG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scripts/prof_cfg.py $CFG $REPS > $OUT/trace.log 2>&1
rc=$?; echo "trace rc=$rc"; tail -2 $OUT/trace.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_$name -- python3 $R/scripts/prof_cfg.py $CFG $REPS > $OUT/pmc_$name.log 2>&1
  rc=$?; echo "pmc $name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
cd $R && python3 scripts/collect_cfg_profile.py ${TAG}_$CFG
