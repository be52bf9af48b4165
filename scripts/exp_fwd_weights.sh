#!/bin/bash
OUT=gpurun_out/${1:-fwdw}; mkdir -p $OUT; export TMPDIR=/tmp
for W in "3 8" "4 10" "4 12" "5 12" "8 8"; do
  set -- $W
  TCNN_FWD_W_SMALL=$1 TCNN_FWD_W_DENSE=$2 timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --dominant grid_forward > $OUT/bench_w$1_$2.json 2>> $OUT/err.log
  python - <<PY
import json
d=json.load(open("$OUT/bench_w$1_$2.json")); print("weights $1 $2: step %.4f ms  grid_forward %.4f ms (timed region %.4f)" % (d["ms_per_step"], d["stages_ms"]["grid_forward"], d["roofline"]["avg_launch_ms"]))
PY
done
