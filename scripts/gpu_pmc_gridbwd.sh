#!/bin/bash
# PMC passes over the grid backward alone (scripts/prof_grid_backward.py).  Usage: bash scripts/gpu_pmc_gridbwd.sh [tag]
TAG=${1:-pmc_gridbwd}
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp
run_pass () { local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -o pmc -- python $ROOT/scripts/prof_grid_backward.py > $OUT/pmc_$name.log 2>&1
  echo "pass $name exit $?"; }
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
run_pass tcc TCC_HIT_sum TCC_MISS_sum
run_pass wait SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
run_pass inst SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM
run_pass ea TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run_pass tccreq TCC_REQ_sum TCC_WRITE_sum TCC_READ_sum TCC_ATOMIC_sum
cd $ROOT
python scripts/parse_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_summary.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -size +5M -delete
