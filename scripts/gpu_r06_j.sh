#!/bin/bash
# Round 6, GPU call J: the network training kernel with all eight waves of a CU in one workgroup (256 slabs) vs four-wave workgroups (512 slabs)
OUT=$PWD/gpurun_out/r06j; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py tests/test_gpu_bench.py -m gpu -x -q -k "mlp or network or train or headline or fused or slab or finalize or pcg32 or wave" > $OUT/pytest_mlp.log 2>&1
echo "mlp tests rc=$? $(grep -E 'passed|failed' $OUT/pytest_mlp.log | tail -1)"; grep -E "^E  |^FAILED" $OUT/pytest_mlp.log | head
rm -f gpurun_out/ab/log.txt
for rep in 1 2 3; do bash scripts/exp_ab.sh wave4 base 2>/dev/null; done
bash scripts/exp_ab.sh --workload hash_shipped wave4 base 2>/dev/null
bash scripts/exp_ab.sh --workload mlp wave4 base 2>/dev/null
cp gpurun_out/ab/log.txt $OUT/ab_log.txt; sort $OUT/ab_log.txt | cut -c1-260
cd /tmp
for v in wave4 base; do
  if [ $v = base ]; then unset TCNN_HIP_LIBRARY; else export TCNN_HIP_LIBRARY=/root/repo/tiny-cuda-nn_amd/lib/variants/$v.so; fi
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$v -o p -- python /root/repo/bench.py --steps 100 --warmup 20 --no-cpu-baseline --api native > $OUT/rocprof_$v.log 2>&1
  f=$(find $OUT/prof_$v -name "*kernel_stats.csv" | head -1); echo "== $v"; head -8 $f | cut -c1-60,150-260
done
