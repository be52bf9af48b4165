#!/bin/bash
# One GPU-box profiling session for profiles/: rocprofv3 kernel trace + stats of the bench command, then the PMC passes.
# Usage (repo root, GPU box): bash scripts/gpu_profile.sh [tag]   -> gpurun_out/<tag>/...
TAG=${1:-r02}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit" | head -8 > $OUT/rocminfo.txt
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" > $OUT/lscpu.txt
timeout 300 python bench.py > $OUT/bench_n1.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-300 $OUT/bench_n1.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $OLDPWD/bench.py --steps 50 --warmup 10 --no-cpu-baseline --api native > $OUT/rocprof.log 2>&1 )
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; cut -c1-170 $f | head -12; done
find $OUT/prof -name "*kernel_trace.csv" -delete; find $OUT/prof -name "*.db" -delete
bash scripts/gpu_pmc.sh $TAG > $OUT/pmc.log 2>&1; tail -5 $OUT/pmc.log
