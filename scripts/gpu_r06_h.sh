#!/bin/bash
# Round 6, GPU call H: bf16 suite again; stage times at the per-GPU batches of strong scaling (the inputs of DESIGN section 6's prediction)
OUT=$PWD/gpurun_out/r06h; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q > $OUT/pytest_bf16.log 2>&1
echo "bf16 rc=$? $(grep -E 'passed|failed' $OUT/pytest_bf16.log | tail -1)"; grep -E "^E  .*(assert|Error)" $OUT/pytest_bf16.log | head -12
for b in 262144 131072 65536 32768; do
  timeout 200 python bench.py --batch $b --steps 200 --warmup 20 --no-cpu-baseline --api native 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $b', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stages_ms'].items() if v>0})"
done 2>&1 | tee $OUT/batches.txt
