#!/bin/bash
# Round 5, second GPU call: full GPU suite on the restructured owner pass, lane variants of the pipelined backward, backward modes on the
# shipped config, the Adam layout microbenchmark, the torch binding's profile.
OUT=$PWD/gpurun_out/r05b; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2; do timeout 120 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --api native --no-inference 2>$OUT/base_$i.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base', round(d['ms_per_step'],4), 'resident', round(d.get('ms_per_step_resident',0),4), {k:round(v,4) for k,v in d['stages_ms'].items() if v>0})" >> $OUT/ab.txt 2>&1; done
for v in "2 0" "2 2" "4 2" "4 0"; do set -- $v
  TCNN_BACKWARD_OVERLAP=$1 TCNN_BACKWARD_OVERLAP_LANES=$2 timeout 120 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --api native --no-inference 2>$OUT/ov_$1_$2.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap groups $1 lanes $2', round(d['ms_per_step'],4), 'resident', round(d.get('ms_per_step_resident',0),4), {k:round(v,4) for k,v in d['stages_ms'].items() if v>0})" >> $OUT/ab.txt 2>&1
done
for m in bucketed sliced_f16 sliced_f32 atomic; do
  TCNN_GRID_BACKWARD=$m timeout 120 python bench.py --workload hash_shipped --steps 200 --warmup 30 --no-cpu-baseline --api native --no-inference 2>$OUT/sh_$m.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hash_shipped backward $m', round(d['ms_per_step'],4), 'resident', round(d.get('ms_per_step_resident',0),4), {k:round(v,4) for k,v in d['stages_ms'].items() if v>0})" >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
timeout 300 scripts/microbench_adam_layout.bin > $OUT/microbench_adam_layout.txt 2>&1; cat $OUT/microbench_adam_layout.txt
for W in hash hash_shipped; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_torch_$W -o trace -- python $OLDPWD/scripts/prof_torch_binding.py $W 200 > $OUT/torch_$W.log 2>&1 )
  tail -3 $OUT/torch_$W.log
  for f in $(find $OUT/prof_torch_$W -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats_torch_$W.csv; cut -c1-150 $f | head -25; done
  find $OUT/prof_torch_$W -name "*kernel_trace.csv" -delete; find $OUT/prof_torch_$W -name "*.db" -delete
done
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
echo done
