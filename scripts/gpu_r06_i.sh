#!/bin/bash
# Round 6, GPU call I: grid kernels with one load round in their prologues (owner: item descriptor; gather: four segments at once) vs the previous commit
OUT=$PWD/gpurun_out/r06i; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py -m gpu -x -q -k "bucket or grid or backward or headline or owner or stress or second_order" > $OUT/pytest_grid.log 2>&1
echo "grid tests rc=$? $(grep -E 'passed|failed' $OUT/pytest_grid.log | tail -1)"; grep -E "^E  |^FAILED" $OUT/pytest_grid.log | head
rm -f gpurun_out/ab/log.txt
for rep in 1 2 3; do bash scripts/exp_ab.sh prevgrid base 2>/dev/null; done
bash scripts/exp_ab.sh --workload stress prevgrid base 2>/dev/null
bash scripts/exp_ab.sh --workload hash_shipped prevgrid base 2>/dev/null
cp gpurun_out/ab/log.txt $OUT/ab_log.txt; sort $OUT/ab_log.txt | cut -c1-260
