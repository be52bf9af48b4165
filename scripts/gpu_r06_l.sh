#!/bin/bash
# Round 6, GPU call L: region-pass gather, the work plan's price of a (slice, tile) item swept (3 / 4 / 6 / 8 / 11 units; a 2 MiB hashed level is 8)
OUT=$PWD/gpurun_out/r06l; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f gpurun_out/ab/log.txt
bash scripts/exp_ab.sh --workload stress base region20c3 region20 region20c6 region20c8 region20c11 base region20c3 region20 region20c6 region20c8 region20c11 2>/dev/null
cp gpurun_out/ab/log.txt $OUT/ab_log.txt; sort $OUT/ab_log.txt | cut -c1-260
