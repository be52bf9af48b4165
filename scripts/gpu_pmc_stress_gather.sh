#!/bin/bash
# PMC passes for the gather of the stress workload (BASELINE configs[4]: sixteen 16 MiB tables against 4 MiB L2s): what bounds
# k_grid_forward_tiles there?  Separate passes, kernel-trace only.  Usage (repo root, GPU box): bash scripts/gpu_pmc_stress_gather.sh [tag]
TAG=${1:-r04sg}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp
run_pass () { local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -o pmc -- python $OLDPWD/bench.py --workload stress --steps 6 --warmup 2 --no-cpu-baseline --no-inference --dominant grid_forward > $OUT/pmc_$name.log 2>&1
  echo "pass $name exit $?"; }
run_pass fetch FETCH_SIZE
run_pass tcc TCC_HIT_sum TCC_MISS_sum
run_pass sq SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pass mem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM
cd $OLDPWD
python scripts/parse_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
grep -A24 "k_grid_forward_tiles" $OUT/pmc_summary.txt | head -60
