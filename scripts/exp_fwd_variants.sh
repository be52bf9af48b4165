#!/bin/bash
# A/B of the forward-gather variants (TCNN_GRID_FWD: 0 per-sample form, 1/2/4 samples per thread of the tiled form).
# Usage (GPU box, repo root): bash scripts/exp_fwd_variants.sh "0 1 2 4" [tag]
OUT=gpurun_out/${2:-fwd}; mkdir -p $OUT; export TMPDIR=/tmp
for V in $1; do
  TCNN_GRID_FWD=$V timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --dominant grid_forward > $OUT/bench_fwd$V.json 2>> $OUT/err.log
  python - <<PY
import json
d=json.load(open("$OUT/bench_fwd$V.json")); print("variant $V step %.4f ms  grid_forward %.4f ms (timed region %.4f)  loss %.5f" % (d["ms_per_step"], d["stages_ms"]["grid_forward"], d["roofline"]["avg_launch_ms"], d["final_loss"]))
PY
done
