#!/bin/bash
# per-kernel times of the grid backward for 2..32 hashed levels (rocprofv3 kernel stats, one process per level count)
OUT=$PWD/gpurun_out/bwdlevels; mkdir -p $OUT; export TMPDIR=/tmp
for L in 2 4 8 16 32; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/L$L -o t -- python $OLDPWD/scripts/exp_bwd_levels.py $L 2>&1 | grep "L=" )
  for f in $(find $OUT/L$L -name "*kernel_stats.csv"); do grep -E "bucket_scatter|bucket_owner|backward_sliced" $f | awk -F, -v L=$L '{gsub(/"/,""); split($1,a,"("); print "  L=" L, substr(a[1],1,50), "avg_us", $4/1000}'; done
  find $OUT/L$L -name "*kernel_trace.csv" -delete; find $OUT/L$L -name "*.db" -delete
done
