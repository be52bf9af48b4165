#!/bin/bash
# A/B of the register-resident MLP training kernel variants (tiny-cuda-nn_amd/lib/variants/*.so) against the
# workgroup-tiled kernel, plus the GPU tests that touch the training pass.
TAG=${1:-wave}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in "$@"; do
  export TCNN_HIP_LIBRARY=$PWD/tiny-cuda-nn_amd/lib/variants/$v.so
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --tb=short -k "activation or fused or training_step or smoke or optimizer_step" > $OUT/pytest_$v.log 2>&1; echo "$v pytest exit $?"; tail -2 $OUT/pytest_$v.log
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_$v.json 2>> $OUT/bench.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_$v.json")); print("$v", round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["stages_ms"].items() if v>0})
PY
done
unset TCNN_HIP_LIBRARY
TCNN_MLP_TRAIN_WAVE=0 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_tiled.json 2>> $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench_tiled.json")); print("tiled", round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["stages_ms"].items() if v>0})
PY
