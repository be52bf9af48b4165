#!/bin/bash
# Round 6, GPU call G: bf16 owner exponent (suite + timing), the compiled torch binding (tests with both bindings, torch_binding legs with and without it,
# host time per step).
OUT=$PWD/gpurun_out/r06g; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q > $OUT/pytest_bf16.log 2>&1
echo "bf16 rc=$? $(grep -E 'passed|failed' $OUT/pytest_bf16.log | tail -1)"; grep -E "^E  .*(assert|Error)" $OUT/pytest_bf16.log | head -12
for p in bf16; do for w in stress hash; do
  timeout 200 python bench.py --workload $w --precision $p --steps 30 --warmup 10 --no-cpu-baseline --api native 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w $p', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stages_ms'].items() if v>0})"
done; done 2>&1 | tee $OUT/bench_bf16.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "torch_modules or fp32_encoding or second_order or module" > $OUT/pytest_binding.log 2>&1
echo "binding tests rc=$? $(grep -E 'passed|failed' $OUT/pytest_binding.log | tail -1)"; grep -E "^E  |^FAILED" $OUT/pytest_binding.log | head -12
for ext in 1 0; do
  for args in "--workload hash" "--workload hash_shipped" "--workload hash --batch 65536" "--workload hash_shipped --batch 65536"; do
    TCNN_TORCH_EXT=$ext timeout 300 python bench.py $args --steps 200 --warmup 20 --no-cpu-baseline --api both 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); tb=d.get('torch_binding',{}); print('ext=$ext $args native', round(d['ms_per_step'],4), 'torch', {k:(round(v,4) if isinstance(v,float) else v) for k,v in tb.items() if k in ('ms_per_step','gpu_ms_per_step','ratio_to_native_step','ms_per_step_fused_adam','ratio_fused_adam','error')})"
  done
done 2>&1 | tee $OUT/binding.txt
for ext in 1 0; do for w in hash_shipped hash; do echo "== ext=$ext $w"; TCNN_TORCH_EXT=$ext timeout 300 python scripts/prof_torch_binding.py $w 200 2>&1 | tail -2; done; done | tee $OUT/binding_host.txt
