#!/bin/bash
# Round 5: the Identity encoding folded into the network kernel's input loads (configs[1]): parity, A/B, kernel stats.
OUT=$PWD/gpurun_out/r05i; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity_full.py -q -k "identity_encoding_itself" > $OUT/pytest_id.log 2>&1; echo "parity rc=$? $(tail -1 $OUT/pytest_id.log)"; grep -E "^E  " $OUT/pytest_id.log | head -5
for rep in 1 2; do for v in 1 0; do
  TCNN_MLP_F32_INPUT=$v timeout 120 python bench.py --workload mlp --steps 300 --warmup 50 --no-cpu-baseline 2>$OUT/mlp_$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mlp f32_input=$v', round(d['ms_per_step'],4), 'inference', round(d['inference']['ms_per_call'],4), {k:round(v,4) for k,v in d['stages_ms'].items() if v>0}, round(d['roofline']['mfma']['frac'],4))"
done; done 2>&1 | tee $OUT/ab.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mlp -o trace -- python $OLDPWD/bench.py --workload mlp --steps 30 --warmup 10 --no-cpu-baseline > $OUT/rocprof_mlp.log 2>&1 )
for f in $(find $OUT/prof_mlp -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats_mlp.csv; cut -c1-150 $f | head -8; done
find $OUT/prof_mlp -name "*kernel_trace.csv" -delete; find $OUT/prof_mlp -name "*.db" -delete
timeout 300 python bench.py --workload mlp --steps 200 --warmup 30 > $OUT/bench_mlp.json 2>$OUT/bench_mlp.err; cut -c1-250 $OUT/bench_mlp.json
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head
echo done
