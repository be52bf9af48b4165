#!/bin/bash
# A tuning variant that differs from the shipped library in ONE source file: scripts/build_variant_one.sh NAME FILE "-DTCNN_FOO=1 ..."
#   -> tiny-cuda-nn_amd/lib/variants/NAME.so (the other objects are the shipped build's, tiny-cuda-nn_amd/lib/obj; run `make` first)
# Run a process against it with TCNN_HIP_LIBRARY=tiny-cuda-nn_amd/lib/variants/NAME.so
set -e
NAME=$1; FILE=$2; DEFS=$3
case "$DEFS" in *TCNN_EXP_*) DEFS="-DTCNN_EXPERIMENT $DEFS";; esac  # csrc/exp_diag.h: a switch without the flag does not compile
ROOT=$(cd $(dirname $0)/.. && pwd)
OBJ=/tmp/tcnn_variant_$NAME; mkdir -p $OBJ $ROOT/tiny-cuda-nn_amd/lib/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function $DEFS -c $ROOT/tiny-cuda-nn_amd/csrc/$FILE.hip -o $OBJ/$FILE.o
OTHERS=$(ls $ROOT/tiny-cuda-nn_amd/lib/obj/*.o | grep -v "/$FILE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tiny-cuda-nn_amd/lib/variants/$NAME.so $OBJ/$FILE.o $OTHERS
echo built $ROOT/tiny-cuda-nn_amd/lib/variants/$NAME.so
