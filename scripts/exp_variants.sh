export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
ROOT=$PWD
run() { tag=$1; shift; out=/tmp/prof_$tag; rm -rf $out
  ( cd /tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python $ROOT/scripts/prof_grid_backward.py > /dev/null 2>&1 )
  echo "=== $tag"; for f in $(find $out -name "*kernel_stats.csv"); do grep -E "bucket|sliced" $f | awk -F, '{gsub(/"/,"",$1); printf "  %-40s calls %s avg %.1f us\n", substr($1,14,36), $2, $4/1000}'; done; }
for v in "$@"; do
  case $v in
    base) run base X=1;;
    slice*) run $v TCNN_GRID_LDS_SLICE_BYTES=${v#slice};;
    *) run $v TCNN_HIP_LIBRARY=$ROOT/tiny-cuda-nn_amd/lib/variants/$v.so;;
  esac
done
