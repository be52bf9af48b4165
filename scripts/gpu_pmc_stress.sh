#!/bin/bash
# PMC passes for the network kernels of the stress workload (BASELINE configs[4], 128 x 4): where do their cycles go?
TAG=${1:-r02s}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run_pass () { local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -o pmc -- python $OLDPWD/bench.py --workload stress --steps 6 --warmup 2 --no-cpu-baseline --dominant adam > $OUT/pmc_$name.log 2>&1
  echo "pass $name exit $?"; }
run_pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
run_pass wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU
run_pass mem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT
cd $OLDPWD
python scripts/parse_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
grep -A30 "k_mlp_backward\|k_mlp_forward" $OUT/pmc_summary.txt | head -90
