#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, bench (+ A/B of the grid-backward modes), rocprofv3 kernel trace.
# Everything lands in gpurun_out/<tag>/.   Usage (repo root, GPU box):  bash scripts/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit" | head -8 > $OUT/rocminfo.txt
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" > $OUT/lscpu.txt
if [ -n "$RUN_MICROBENCH" ] && [ -x scripts/microbench_atomics.bin ]; then timeout 300 scripts/microbench_atomics.bin > $OUT/microbench_atomics.txt 2>&1; fi
echo "== smoke" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 ; echo "smoke exit $?" | tee -a $OUT/smoke.log
echo "== pytest -m gpu" ; timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider --tb=line > $OUT/pytest_gpu.log 2>&1 ; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
grep -E "^/|passed|failed|Error" $OUT/pytest_gpu.log | cut -c1-400 | tail -30
echo "== bench" ; timeout 900 python bench.py --steps 100 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err ; echo "bench exit $?" ; cut -c1-1800 $OUT/bench.json; tail -3 $OUT/bench.err
for MODE in ${AB_MODES:-bucketed sliced_f16 sliced_f32 atomic}; do
  TCNN_GRID_BACKWARD=$MODE timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_$MODE.json 2>> $OUT/bench.err
  python - <<EOF
import json
d=json.load(open("$OUT/bench_$MODE.json")); print("$MODE", round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["stages_ms"].items()})
EOF
done
echo "== rocprofv3 kernel trace"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o trace -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OLDPWD/$OUT/rocprof.log 2>&1 )
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cut -c1-160 $f | head -14; done
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
find $OUT/prof -name "*.db" -delete
echo done
