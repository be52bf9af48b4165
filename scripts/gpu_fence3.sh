#!/bin/bash
OUT=gpurun_out/fence3
mkdir -p $OUT
: > $OUT/summary.txt
for reuse in 0 1; do
for rep in 1 2; do
  TCNN_DEBUG_ALLOC=fence TCNN_DEBUG_ALLOC_REUSE_VA=$reuse timeout 1200 python3 -m pytest tests -m gpu -q -p no:cacheprovider --tb=line --deselect tests/test_gpu_bf16.py > $OUT/pytest_reuse${reuse}_$rep.out 2>&1
  echo "fp16 suite fence reuse_va=$reuse rep $rep rc=$? $(tail -n 1 $OUT/pytest_reuse${reuse}_$rep.out)" >> $OUT/summary.txt
done
done
TCNN_DEBUG_ALLOC=fence TCNN_PRECISION=bf16 timeout 600 python3 -m pytest tests/bf16_cases.py -m gpu -q -p no:cacheprovider --tb=line > $OUT/bf16.out 2>&1
echo "bf16 cases fence keep-va rc=$? $(tail -n 1 $OUT/bf16.out)" >> $OUT/summary.txt
cat $OUT/summary.txt
