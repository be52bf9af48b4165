#!/bin/bash
# Round 5, sixth GPU call: re-sweep of the build-time launch geometries after the round's codegen changes (owner pass restructured), the PMC passes
# of the stress shape.
OUT=$PWD/gpurun_out/r05f; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do
  bash scripts/exp_ab.sh base rw1536 rw2560 rw3072 og3 mw384 mw768 mw1024
done
cat gpurun_out/ab/log.txt; cp gpurun_out/ab/log.txt $OUT/sweep.txt
cd /tmp
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "tcc TCC_HIT_sum TCC_MISS_sum"; do set -- $pass; name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -o pmc -- python $OLDPWD/bench.py --workload stress --steps 6 --warmup 2 --no-cpu-baseline --no-inference --api native --dominant adam > $OUT/pmc_$name.log 2>&1
  echo "stress pass $name exit $?"
done
cd $OLDPWD
python scripts/parse_pmc.py $OUT > $OUT/pmc_stress_summary.txt 2>&1; grep -B2 -A12 "k_adam_step\|k_grid_forward_tiles\|k_grid_bucket" $OUT/pmc_stress_summary.txt | head -80
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
echo done
