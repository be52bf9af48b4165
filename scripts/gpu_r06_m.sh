#!/bin/bash
# Round 6, GPU call M: how much of the driver's 20-step figure is start-up (clocks, first touches)?  The driver's command with 5 / 50 / 300 warm-up steps.
OUT=$PWD/gpurun_out/r06m; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2 3; do
for W in 5 50 300; do
  python bench.py --gpus 1 --steps 20 --warmup $W --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('warmup $W', round(d['ms_per_step'],4), 'resident', round(d.get('ms_per_step_resident',0),4))" | tee -a $OUT/warmup.txt
done; done
python bench.py --gpus 1 --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1000 steps', round(d['ms_per_step'],4), 'resident', round(d.get('ms_per_step_resident',0),4))" | tee -a $OUT/warmup.txt
