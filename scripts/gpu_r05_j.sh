#!/bin/bash
# Round 5: a module's inference through the fp32-input inference kernel (padded 16-bit output) + the whole GPU suite at the shipping commit.
OUT=$PWD/gpurun_out/r05j; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity_full.py -q -k "identity_encoding_itself" > $OUT/pytest_id.log 2>&1; echo "parity rc=$? $(tail -1 $OUT/pytest_id.log)"; grep -E "^E  " $OUT/pytest_id.log | head -5
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head
for i in 1 2; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_$i.json 2> $OUT/driver_$i.err; echo "driver cmd $i rc=$? $(python3 -c "import json; d=json.load(open('$OUT/driver_$i.json')); print(d['ms_per_step'], d['ms_per_step_resident'], d['roofline']['frac'])")"; done
echo done
