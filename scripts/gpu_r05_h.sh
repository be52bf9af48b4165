#!/bin/bash
# same-box A/B of the PyTorch binding's Python side: the package as shipped against a copy with the previous modules.py
OUT=$PWD/gpurun_out/r05h; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
rm -rf /tmp/pkg_old && cp -r tiny-cuda-nn_amd /tmp/pkg_old && cp scripts/ab_tmp/modules_old.py /tmp/pkg_old/tinycudann/modules.py
for rep in 1 2 3; do
  for W in hash_shipped hash; do
    echo "new $W $(timeout 200 python scripts/prof_torch_binding.py $W 300 2>&1 | grep ms_per_step | head -1 | cut -c1-160)"
    echo "old $W $(TCNN_PKG_DIR=/tmp/pkg_old timeout 200 python scripts/prof_torch_binding.py $W 300 2>&1 | grep ms_per_step | head -1 | cut -c1-160)"
  done
done 2>&1 | tee $OUT/binding_ab.txt
echo done
