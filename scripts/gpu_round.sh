#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, bench, rocprofv3 kernel trace.  Everything lands in gpurun_out/.
# Usage (from the repo root on the GPU box):  bash scripts/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit" | head -8 > $OUT/rocminfo.txt
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" > $OUT/lscpu.txt
echo "== smoke" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 ; echo "smoke exit $?" | tee -a $OUT/smoke.log
echo "== pytest -m gpu" ; timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1 ; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
echo "== bench" ; timeout 900 python bench.py --steps 100 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err ; echo "bench exit $?" ; cat $OUT/bench.json
for B in 0 32768 65536; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --lds-budget $B > $OUT/bench_lds$B.json 2>> $OUT/bench.err
done
echo "== rocprofv3 kernel trace"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o trace -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OLDPWD/$OUT/rocprof.log 2>&1 )
find $OUT/prof -name "*stats*" | head; for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -25 $f; done
# keep only the small summaries (the raw trace can be large)
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
echo done
