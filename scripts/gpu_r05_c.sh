#!/bin/bash
# Round 5, third GPU call: device AddressSanitizer attempt, finalize-aside A/B, torch binding after the trims, the whole GPU suite.
OUT=$PWD/gpurun_out/r05c; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
# ---- device ASan: needs an xnack+ agent
( export HSA_XNACK=1; /opt/rocm/bin/rocminfo | grep -i "xnack\|Marketing" | head -4
  ASAN=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
  for W in hash mlp stress; do
    LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0 TCNN_HIP_LIBRARY=$PWD/tiny-cuda-nn_amd/lib/variants/asan.so \
      timeout 400 python bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --api native --worker > $OUT/asan_$W.json 2> $OUT/asan_$W.err
    echo "asan $W rc=$? $(cut -c1-120 $OUT/asan_$W.json) | $(grep -i "AddressSanitizer\|error\|xnack\|No binary\|invalid device function" $OUT/asan_$W.err | head -3 | cut -c1-300)"
  done ) 2>&1 | tee $OUT/asan.txt
# ---- finalize beside the encoding's backward
for i in 1 2 3; do for v in 0 1; do
  TCNN_FINALIZE_ASIDE=$v timeout 120 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --api native --no-inference 2>$OUT/fa_$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('finalize_aside $v', round(d['ms_per_step'],4), 'resident', round(d.get('ms_per_step_resident',0),4), {k:round(v,4) for k,v in d['stages_ms'].items() if v>0})" >> $OUT/ab.txt 2>&1
done; done
cat $OUT/ab.txt
for W in hash hash_shipped; do timeout 300 python bench.py --workload $W --steps 500 --warmup 50 --no-cpu-baseline > $OUT/bench_$W.json 2> $OUT/bench_$W.err; python -c "import json; d=json.load(open('$OUT/bench_$W.json')); print('$W', d['ms_per_step'], d['torch_binding'])"; done
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head
echo done
