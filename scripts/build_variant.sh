#!/bin/bash
# Builds a tuning variant of the library: scripts/build_variant.sh NAME "-DTCNN_FOO=1 ..."  ->  tiny-cuda-nn_amd/lib/variants/NAME.so
# Run a process against it with TCNN_HIP_LIBRARY=tiny-cuda-nn_amd/lib/variants/NAME.so
set -e
NAME=$1; DEFS=$2
ROOT=$(cd $(dirname $0)/.. && pwd)
OBJ=/tmp/tcnn_variant_$NAME; mkdir -p $OBJ $ROOT/tiny-cuda-nn_amd/lib/variants
for f in grid_kernels mlp_kernels mlp_train_wave mlp_train_wide elementwise_kernels api; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function $DEFS -c $ROOT/tiny-cuda-nn_amd/csrc/$f.hip -o $OBJ/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tiny-cuda-nn_amd/lib/variants/$NAME.so $OBJ/*.o
echo built $ROOT/tiny-cuda-nn_amd/lib/variants/$NAME.so
