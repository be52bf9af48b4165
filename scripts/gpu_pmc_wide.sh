#!/bin/bash
# Kernel stats + PMC passes (own runs) of the stress workload's training step: the 128-wide fused network kernel.
TAG=${1:-wide}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $REPO/scripts/prof_cfg4.py > $OUT/prof.log 2>&1
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; cut -d, -f1-4 $f | cut -c1-150 | head -12; done
run_pass () { local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -o pmc -- python $REPO/scripts/prof_cfg4.py > $OUT/pmc_$name.log 2>&1
  echo "pass $name exit $?"; }
run_pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
run_pass wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU
cd $REPO
python scripts/parse_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
grep -A18 "k_mlp_train_wide" $OUT/pmc_summary.txt | head -40
