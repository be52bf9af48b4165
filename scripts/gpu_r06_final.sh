#!/bin/bash
# Round-6 evidence for profiles/ at the commit that ships: the driver's command twice, the default bench, rocprofv3 kernel statistics of the
# workloads, the PMC passes (traffic), the bf16 benches, the spilling-instance comparison, the fresh-process soak, then the whole GPU suite.
# Usage (repo root, GPU box): bash scripts/gpu_r06_final.sh [tag]
TAG=${1:-r06final}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_$i.json 2> $OUT/driver_$i.err; echo "driver cmd $i rc=$? $(python3 -c "import json; d=json.load(open('$OUT/driver_$i.json')); print(d['ms_per_step'], d['ms_per_step_resident'], d['inference']['ms_per_call'], d['roofline']['frac'], d['torch_binding']['ms_per_step'])")"; done
bash scripts/gpu_profile.sh $TAG
for W in mlp stress hash_shipped; do
  timeout 300 python bench.py --workload $W --steps 200 --warmup 30 > $OUT/bench_$W.json 2>> $OUT/bench.err; echo "$W: $(cut -c1-200 $OUT/bench_$W.json)"
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$W -o trace -- python $OLDPWD/bench.py --workload $W --steps 30 --warmup 10 --no-cpu-baseline --api native > $OUT/rocprof_$W.log 2>&1 )
  for f in $(find $OUT/prof_$W -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats_$W.csv; cut -c1-150 $f | head -8; done
  find $OUT/prof_$W -name "*kernel_trace.csv" -delete; find $OUT/prof_$W -name "*.db" -delete
done
timeout 300 python bench.py --workload stress --precision bf16 --steps 200 --warmup 30 --no-cpu-baseline > $OUT/bench_stress_bf16.json 2>> $OUT/bench.err
timeout 300 python bench.py --precision bf16 --steps 200 --warmup 30 --no-cpu-baseline > $OUT/bench_hash_bf16.json 2>> $OUT/bench.err
for B in 65536 16384; do timeout 200 python bench.py --batch $B --steps 500 --warmup 50 --no-cpu-baseline > $OUT/bench_batch_$B.json 2>> $OUT/bench.err; python3 -c "import json; d=json.load(open('$OUT/bench_batch_$B.json')); print('batch $B: native', round(d['ms_per_step'],4), 'ms; torch binding', round(d['torch_binding']['ms_per_step'],4), 'ratio', round(d['torch_binding']['ratio_to_native_step'],2), 'fused adam', d['torch_binding'].get('with_fused_adam',{}).get('ratio_to_native_step'))"; done

bash scripts/soak_first_steps.sh ${SOAK_N:-8} $OUT/soak_first_steps.txt
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head
echo done
