#!/bin/bash
# Round 6, GPU call B: network kernel variants (bank-conflict-free transpose tiles, weight fragments a phase ahead) A/B + SQ counters.
OUT=$PWD/gpurun_out/r06b; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "network or mlp or fused or golden or wide or reference" > $OUT/pytest_subset.log 2>&1
echo "subset rc=$? $(tail -1 $OUT/pytest_subset.log)"; grep -E "^FAILED|^ERROR|Error" $OUT/pytest_subset.log | head -10
rm -f gpurun_out/ab/log.txt
for rep in 1 2 3; do
  bash scripts/exp_ab.sh "$@" 2>/dev/null
done
cp gpurun_out/ab/log.txt $OUT/ab_log.txt; sort $OUT/ab_log.txt | cut -c1-250
cd /tmp
for v in new; do
  unset TCNN_HIP_LIBRARY
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/$v/pmc_sq -o pmc -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inference --api native > $OUT/pmc_sq_$v.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/$v/pmc_wait -o pmc -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inference --api native > $OUT/pmc_wait_$v.log 2>&1
done
cd $OLDPWD
for v in new; do python scripts/parse_pmc.py $OUT/$v > $OUT/pmc_summary_$v.txt 2>&1; echo "== $v"; grep -A18 "mlp_train_wave" $OUT/pmc_summary_$v.txt | head -20; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
