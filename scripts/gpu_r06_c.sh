#!/bin/bash
# Round 6, GPU call C: network kernels without store-latency waits -- A/B on the three workloads (r05 = round 5's two network kernels).
OUT=$PWD/gpurun_out/r06c; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "network or mlp or fused or golden or wide or reference" > $OUT/pytest_subset.log 2>&1
echo "subset rc=$? $(tail -1 $OUT/pytest_subset.log)"; grep -E "^FAILED|^ERROR|Error" $OUT/pytest_subset.log | head -10
rm -f gpurun_out/ab/log.txt
for rep in 1 2; do
  bash scripts/exp_ab.sh r05 base nopf 2>/dev/null
  bash scripts/exp_ab.sh --workload mlp r05 base nopf 2>/dev/null
  bash scripts/exp_ab.sh --workload stress r05 base 2>/dev/null
done
cp gpurun_out/ab/log.txt $OUT/ab_log.txt; sort $OUT/ab_log.txt | cut -c1-260
