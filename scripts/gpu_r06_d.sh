#!/bin/bash
# Round 6, GPU call D: the record scatter without waits behind its queue stores (oldgrid = the committed grid kernels) + SQ counters of
# the 128-wide network kernel on the stress workload.
OUT=$PWD/gpurun_out/r06d; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py -m gpu -x -q -k "bucket or grid or backward or headline or owner" > $OUT/pytest_subset.log 2>&1
echo "subset rc=$? $(tail -1 $OUT/pytest_subset.log)"; grep -E "^FAILED|^ERROR|Error" $OUT/pytest_subset.log | head -10
rm -f gpurun_out/ab/log.txt
for rep in 1 2 3; do
  bash scripts/exp_ab.sh oldgrid base 2>/dev/null
done
bash scripts/exp_ab.sh --workload stress oldgrid base 2>/dev/null
bash scripts/exp_ab.sh --workload hash_shipped oldgrid base 2>/dev/null
cp gpurun_out/ab/log.txt $OUT/ab_log.txt; sort $OUT/ab_log.txt | cut -c1-260
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/stress/pmc_sq -o pmc -- python $OLDPWD/bench.py --workload stress --steps 6 --warmup 2 --no-cpu-baseline --no-inference --api native > $OUT/pmc_sq_stress.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/stress/pmc_wait -o pmc -- python $OLDPWD/bench.py --workload stress --steps 6 --warmup 2 --no-cpu-baseline --no-inference --api native > $OUT/pmc_wait_stress.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/hash/pmc_sq -o pmc -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inference --api native > $OUT/pmc_sq_hash.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/hash/pmc_wait -o pmc -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inference --api native > $OUT/pmc_wait_hash.log 2>&1
cd $OLDPWD
for v in stress hash; do python scripts/parse_pmc.py $OUT/$v > $OUT/pmc_summary_$v.txt 2>&1; done
grep -A17 "mlp_train_wide" $OUT/pmc_summary_stress.txt | head -20
grep -A17 "k_grid_bucket_scatter\|k_grid_bucket_owner\|k_grid_forward_tiles\|k_adam_step" $OUT/pmc_summary_hash.txt | head -90
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
