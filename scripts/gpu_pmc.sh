#!/bin/bash
# PMC passes (own runs, kernel-trace only -- never combined with sys/hip/hsa traces) for the bench workload.
# Usage (repo root, GPU box): bash scripts/gpu_pmc.sh [tag]   -> gpurun_out/<tag>/pmc_*/..., summarised by scripts/parse_pmc.py
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp
run_pass () {  # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -o pmc -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inference --api native --dominant grid_forward > $OUT/pmc_$name.log 2>&1
  echo "pass $name exit $?"
}
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
run_pass tcc TCC_HIT_sum TCC_MISS_sum
run_pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run_pass mfma SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU
cd $OLDPWD
python scripts/parse_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_summary.txt
# keep the summaries, drop bulky per-dispatch files
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
