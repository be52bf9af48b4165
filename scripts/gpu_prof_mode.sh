#!/bin/bash
# rocprofv3 kernel trace of bench.py under one grid-backward mode.  Usage: bash scripts/gpu_prof_mode.sh <mode> [tag]
MODE=${1:-bucketed}; TAG=${2:-prof_$MODE}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 TCNN_GRID_BACKWARD=$MODE
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/rocprof.log 2>&1 )
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cut -c1-150 $f | head -14; done
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete; find $OUT/prof -name "*.db" -delete
