#!/bin/bash
OUT=gpurun_out/fence2
mkdir -p $OUT
: > $OUT/summary.txt
for a in 16 64 128 256 4096; do
  TCNN_DEBUG_ALLOC=fence TCNN_DEBUG_ALLOC_ALIGN=$a TCNN_PRECISION=bf16 timeout 600 python3 -m pytest tests/bf16_cases.py -m gpu -q -p no:cacheprovider --tb=line > $OUT/bf16_align$a.out 2>&1
  echo "bf16 cases align $a rc=$? $(tail -n 1 $OUT/bf16_align$a.out)" >> $OUT/summary.txt
done
TCNN_DEBUG_ALLOC=fence timeout 1200 python3 -m pytest tests -m gpu -q -p no:cacheprovider --tb=line --deselect tests/test_gpu_bf16.py > $OUT/pytest_fence.out 2>&1
echo "fp16 suite fence align 64 rc=$? $(tail -n 1 $OUT/pytest_fence.out)" >> $OUT/summary.txt
TCNN_DEBUG_ALLOC=fence TCNN_DEBUG_ALLOC_ALIGN=16 timeout 1200 python3 -m pytest tests -m gpu -q -p no:cacheprovider --tb=line --deselect tests/test_gpu_bf16.py > $OUT/pytest_fence16.out 2>&1
echo "fp16 suite fence align 16 rc=$? $(tail -n 1 $OUT/pytest_fence16.out)" >> $OUT/summary.txt
cat $OUT/summary.txt
