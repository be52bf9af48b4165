#!/bin/bash
# Forward gather: levels that fit LDS through k_grid_forward_lds (TCNN_GRID_FWD_LDS_BYTES: 0 = none, 32768 = level 0 only, default 152 KiB)
OUT=gpurun_out/fwdlds; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --tb=short -k "grid or forward or full or encod" > $OUT/pytest.log 2>&1; echo "pytest rc=$? $(tail -n 1 $OUT/pytest.log)"
run() { python bench.py --steps 200 --warmup 30 --no-cpu-baseline $2 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e6,1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stages_ms'].items() if v>0})"; }
TCNN_GRID_FWD_LDS_BYTES=0 run lds_none
TCNN_GRID_FWD_LDS_BYTES=32768 run lds_level0
run lds_default
TCNN_GRID_FWD_LDS_BYTES=0 run lds_none_again
run lds_default_again
echo "--- per level kind, LDS off / on"
TCNN_GRID_FWD_LDS_BYTES=0 python scripts/exp_grid_forward_levels.py 2>&1 | grep -v amdgpu.ids
python scripts/exp_grid_forward_levels.py 2>&1 | grep -v amdgpu.ids
