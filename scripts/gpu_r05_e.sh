#!/bin/bash
# Round 5, fifth GPU call: the gather's tail workgroups (TCNN_GRID_FWD_TAIL_TILES: 0 = whole tiles throughout, the form of rounds 2-4).
OUT=$PWD/gpurun_out/r05e; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do for t in 0 32 64 128 256 512; do
  TCNN_GRID_FWD_TAIL_TILES=$t timeout 120 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --api native 2>$OUT/t_$t.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tail $t', round(d['ms_per_step'],4), 'resident', round(d.get('ms_per_step_resident',0),4), 'gather(event)', round(d['roofline']['avg_launch_ms'],4), 'inference', round(d['inference']['ms_per_call'],4), {k:round(v,4) for k,v in d['stages_ms'].items() if v>0})" >> $OUT/ab.txt 2>&1
done; done
for W in stress hash_shipped; do for t in 0 64 128 256; do
  TCNN_GRID_FWD_TAIL_TILES=$t timeout 200 python bench.py --workload $W --steps 100 --warmup 20 --no-cpu-baseline --api native 2>>$OUT/w.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W tail $t', round(d['ms_per_step'],4), 'inference', round(d['inference']['ms_per_call'],4), {k:round(v,4) for k,v in d['stages_ms'].items() if v>0})" >> $OUT/ab.txt 2>&1
done; done
cat $OUT/ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "grid or owner_pass or forward" > $OUT/pytest_grid.log 2>&1; echo "grid tests rc=$? $(tail -1 $OUT/pytest_grid.log)"
echo done
