#!/bin/bash
# A/B of the bucket-owner forms (pass B of the grid backward): packed (512 / 1024 threads) vs 64-bit fixed point per value,
# each with and without the optimizer step inside the owner pass
OUT=gpurun_out/owner; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --tb=short -k "grid or backward or full or step or fused" > $OUT/pytest.log 2>&1; echo "pytest rc=$? $(tail -n 1 $OUT/pytest.log)"
run() { python bench.py --steps 200 --warmup 30 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e6,1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stages_ms'].items() if v>0})"; }
run packed512
TCNN_FUSED_OPTIMIZER=1 run packed512_fusedadam
TCNN_GRID_OWNER=fixed64 TCNN_FUSED_OPTIMIZER=1 run fixed64_fusedadam
TCNN_HIP_LIBRARY=$PWD/tiny-cuda-nn_amd/lib/variants/own1024.so run packed1024
TCNN_HIP_LIBRARY=$PWD/tiny-cuda-nn_amd/lib/variants/own1024.so TCNN_FUSED_OPTIMIZER=1 run packed1024_fusedadam
TCNN_HIP_LIBRARY=$PWD/tiny-cuda-nn_amd/lib/variants/own256.so run packed256
run packed512_again
