#!/bin/bash
# A/B of library variants (scripts/build_variant*.sh) on bench.py: bash scripts/exp_ab.sh [--workload W] base variantA ...  -> gpurun_out/ab/log.txt
OUT=gpurun_out/ab; mkdir -p $OUT
ARGS=""
if [ "$1" = "--workload" ]; then ARGS="--workload $2"; shift 2; fi
for v in "$@"; do
  if [ $v = base ]; then unset TCNN_HIP_LIBRARY; else export TCNN_HIP_LIBRARY=$PWD/tiny-cuda-nn_amd/lib/variants/$v.so; fi
  timeout 120 python bench.py $ARGS --steps 100 --warmup 20 --no-cpu-baseline --api native 2>$OUT/$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$ARGS $v', round(d['value']/1e6,1), round(d['ms_per_step'],4), 'resident', round(d.get('ms_per_step_resident',0),4), 'inference', round(d.get('inference',{}).get('ms_per_call',0),4), {k:round(v,4) for k,v in d['stages_ms'].items() if v>0})" >> $OUT/log.txt 2>&1
done
