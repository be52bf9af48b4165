#!/bin/bash
# Forward gather diagnostics: TCNN_GRID_FWD_EXP bits 1 = odd corners not loaded, 2 = positions synthesised, 4 = no stores.
OUT=gpurun_out/${2:-fwdx}; mkdir -p $OUT; export TMPDIR=/tmp
for E in $1; do
  TCNN_GRID_FWD=2 TCNN_GRID_FWD_EXP=$E timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --dominant grid_forward > $OUT/bench_exp$E.json 2>> $OUT/err.log
  python - <<PY
import json
d=json.load(open("$OUT/bench_exp$E.json")); print("exp $E step %.4f ms  grid_forward %.4f ms" % (d["ms_per_step"], d["stages_ms"]["grid_forward"]))
PY
done
