#!/bin/bash
# Round 5, last GPU call: the GPU suite under the canary allocator and under the fence allocator at the shipping commit.
OUT=$PWD/gpurun_out/r05k; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for mode in canary fence; do
  TCNN_DEBUG_ALLOC=$mode timeout 1200 python -m pytest tests -m gpu -q -k "not direct and not distributed and not multi_gpu and not launches_its_own" > $OUT/pytest_$mode.log 2>&1; echo "$mode suite rc=$? $(grep -E 'passed|failed|error' $OUT/pytest_$mode.log | tail -1)"; grep -E "^FAILED|^ERROR" $OUT/pytest_$mode.log | head -5
done
echo done
