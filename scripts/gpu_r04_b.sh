#!/bin/bash
# round 4, second half: direct-exchange link check + --dp auto on one GPU; paired-gather variants A/B
OUT=gpurun_out/r04b; mkdir -p $OUT gpurun_out/ab; rm -f gpurun_out/ab/log.txt
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_distributed.py -x -q -k "direct" > $OUT/t_direct.txt 2>&1; tail -5 $OUT/t_direct.txt
timeout 900 python -m pytest tests/test_gpu_bench.py -x -q -k "two_ranks" > $OUT/t_bench2.txt 2>&1; tail -5 $OUT/t_bench2.txt
for v in fwdp1 fwdp2; do
  TCNN_HIP_LIBRARY=$PWD/tiny-cuda-nn_amd/lib/variants/$v.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py -x -q -k "grid or forward or headline or encoding" > $OUT/t_$v.txt 2>&1; tail -3 $OUT/t_$v.txt
done
bash scripts/exp_ab.sh base fwdp1 fwdp2 base fwdp1 fwdp2
cat gpurun_out/ab/log.txt
