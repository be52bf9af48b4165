#!/bin/bash
# The parked round-2 finding: k_mlp_train_wave<64,32,1> with spilled registers gave wrong, run-to-run varying results.
# Variants (scripts/build_variant.sh): spill_rt = -DTCNN_EXP_RUNTIME_EXTERNAL; spill_rt_nopN = the same + N + 1 wait states after
# the output layer's MFMA (-DTCNN_EXP_NOP_AFTER_OUTPUT_MFMA=N).
OUT=gpurun_out/spill; mkdir -p $OUT; : > $OUT/log.txt
for lib in "" spill_rt spill_rt_nop1 spill_rt_nop3 spill_rt_nop15; do
  for n in 2048 262144; do
    echo "=== library [${lib:-shipped}] n=$n" >> $OUT/log.txt
    TCNN_HIP_LIBRARY=${lib:+tiny-cuda-nn_amd/lib/variants/$lib.so} timeout 300 python3 scripts/exp_spill_wave.py $n 6 2>&1 | grep -v amdgpu.ids >> $OUT/log.txt
  done
done
cat $OUT/log.txt
