#!/bin/bash
# Round 6, GPU call K: region-pass gather on the stress shape -- timing builds (csrc/exp_diag.h, TCNN_EXP_FWD_REGION_LOG2)
OUT=$PWD/gpurun_out/r06k; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f gpurun_out/ab/log.txt
bash scripts/exp_ab.sh --workload stress base region20 region19 base region20 region19 2>/dev/null
bash scripts/exp_ab.sh base region19 2>/dev/null
cp gpurun_out/ab/log.txt $OUT/ab_log.txt; sort $OUT/ab_log.txt | cut -c1-260
for v in region20 base; do
  if [ $v = base ]; then unset TCNN_HIP_LIBRARY; else export TCNN_HIP_LIBRARY=$PWD/tiny-cuda-nn_amd/lib/variants/$v.so; fi
  timeout 700 bash scripts/gpu_pmc_stress_gather.sh r06k/pmc_$v > $OUT/pmc_$v.txt 2>&1
  grep -A24 "k_grid_forward_tiles" $OUT/pmc_$v/pmc_summary.txt | head -30
done
