#!/bin/bash
# Round 6, GPU call F: bfloat16 owner pass with a per-slice fixed-point exponent -- the bf16 suite, the fp16 grid tests, stress timing in both precisions.
OUT=$PWD/gpurun_out/r06f; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q > $OUT/pytest_bf16.log 2>&1
echo "bf16 rc=$? $(grep -E 'passed|failed' $OUT/pytest_bf16.log | tail -1)"; grep -E "^FAILED|^ERROR|^E  " $OUT/pytest_bf16.log | head -20
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py -m gpu -x -q -k "bucket or grid or backward or headline or owner or stress" > $OUT/pytest_grid.log 2>&1
echo "fp16 grid rc=$? $(grep -E 'passed|failed' $OUT/pytest_grid.log | tail -1)"
for p in fp16 bf16; do for w in stress hash; do
  timeout 200 python bench.py --workload $w --precision $p --steps 30 --warmup 10 --no-cpu-baseline --api native 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w $p', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stages_ms'].items() if v>0}, 'wide slices', d.get('grid_owner_wide_slices'))"
done; done 2>&1 | tee $OUT/bench.txt
