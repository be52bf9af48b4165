#!/bin/bash
# Round 6, GPU call A: the network kernel with LDS transposes / three-instruction masks -- parity first, then A/B against the round-5
# kernel and the two half-way variants, kernel trace and SQ counters of the new kernel.
OUT=$PWD/gpurun_out/r06a; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py -m gpu -x -q -k "network or mlp or fused or headline or golden or wide or reference or stress or module or torch" > $OUT/pytest_subset.log 2>&1
echo "subset rc=$? $(tail -1 $OUT/pytest_subset.log)"; grep -E "^FAILED|^ERROR|Error" $OUT/pytest_subset.log | head -10
rm -f gpurun_out/ab/log.txt
for rep in 1 2; do
  bash scripts/exp_ab.sh r05 base mfmatr oldmask 2>/dev/null   # "base" = the default library = the new kernel; r05 = round 5's kernel
done
unset TCNN_HIP_LIBRARY
for v in r05 default; do
  if [ $v = r05 ]; then export TCNN_HIP_LIBRARY=$PWD/tiny-cuda-nn_amd/lib/variants/r05.so; else unset TCNN_HIP_LIBRARY; fi
  timeout 120 python bench.py --workload mlp --steps 100 --warmup 20 --no-cpu-baseline --api native 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mlp $v', round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['stages_ms'].items() if x>0})" >> gpurun_out/ab/log.txt 2>&1
  timeout 200 python bench.py --workload stress --steps 30 --warmup 10 --no-cpu-baseline --api native 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stress $v', round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['stages_ms'].items() if x>0})" >> gpurun_out/ab/log.txt 2>&1
done
unset TCNN_HIP_LIBRARY
cp gpurun_out/ab/log.txt $OUT/ab_log.txt; cat $OUT/ab_log.txt
# kernel trace of the new library
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $OLDPWD/bench.py --steps 50 --warmup 10 --no-cpu-baseline --api native > $OUT/rocprof.log 2>&1 )
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; cut -c1-150 $f | head -12; done
find $OUT/prof -name "*kernel_trace.csv" -delete; find $OUT/prof -name "*.db" -delete
# SQ counters, new and base
cd /tmp
for v in new r05; do
  if [ $v = r05 ]; then export TCNN_HIP_LIBRARY=$OLDPWD/tiny-cuda-nn_amd/lib/variants/r05.so; else unset TCNN_HIP_LIBRARY; fi
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/$v/pmc_sq -o pmc -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inference --api native > $OUT/pmc_sq_$v.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/$v/pmc_wait -o pmc -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inference --api native > $OUT/pmc_wait_$v.log 2>&1
done
unset TCNN_HIP_LIBRARY
cd $OLDPWD
for v in new r05; do python scripts/parse_pmc.py $OUT/$v > $OUT/pmc_summary_$v.txt 2>&1; echo "== $v"; grep -A18 "mlp_train_wave" $OUT/pmc_summary_$v.txt | head -40; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
