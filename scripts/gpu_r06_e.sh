#!/bin/bash
# Round 6, GPU call E: the weight-gradient finalize inside the optimizer's launch -- full GPU suite, then A/B (TCNN_FINALIZE_SEPARATE=1 makes
# bench.py call tcnn_set_finalize_in_optimizer(0)), kernel trace.
OUT=$PWD/gpurun_out/r06e; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
echo "gpu suite rc=$? $(tail -1 $OUT/pytest_gpu.log)"; grep -E "^FAILED|^ERROR|Error" $OUT/pytest_gpu.log | head -10
for rep in 1 2 3; do
  for sep in 0 1; do
    TCNN_FINALIZE_SEPARATE=$sep timeout 120 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --api native 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hash separate=$sep', round(d['ms_per_step'],4), 'resident', round(d.get('ms_per_step_resident',0),4), {k:round(v,4) for k,v in d['stages_ms'].items() if v>0})"
  done
done 2>&1 | tee $OUT/ab.txt
for w in mlp hash_shipped stress; do for sep in 0 1; do
  TCNN_FINALIZE_SEPARATE=$sep timeout 200 python bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline --api native 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w separate=$sep', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stages_ms'].items() if v>0})"
done; done 2>&1 | tee -a $OUT/ab.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $OLDPWD/bench.py --steps 50 --warmup 10 --no-cpu-baseline --api native > $OUT/rocprof.log 2>&1 )
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; cut -c1-150 $f | head -10; done
find $OUT/prof -name "*kernel_trace.csv" -delete; find $OUT/prof -name "*.db" -delete
