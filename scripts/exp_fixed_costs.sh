#!/bin/bash
# Fixed costs of the step's kernels: timing-only diagnostic builds that return at the top of a kernel (or after its prologue), stage times from bench.py's HIP events.
# Variants: scripts/build_variant_one.sh NAME FILE "-DTCNN_EXP_DIAG_EMPTY_..." (results are wrong on purpose)
OUT=gpurun_out/fixedcosts; mkdir -p $OUT; : > $OUT/log.txt
for v in base "$@"; do
  if [ $v = base ]; then unset TCNN_HIP_LIBRARY; else export TCNN_HIP_LIBRARY=$PWD/tiny-cuda-nn_amd/lib/variants/$v.so; fi
  timeout 120 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>$OUT/$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stages_ms'].items() if v>0})" >> $OUT/log.txt 2>&1
done
cat $OUT/log.txt
