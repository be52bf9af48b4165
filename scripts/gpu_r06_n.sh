#!/bin/bash
# Round 6, GPU call N: training_step as one graph launch -- the test, then plain vs captured over batch sizes
OUT=$PWD/gpurun_out/r06n; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity_full.py -m gpu -x -q -k "graph_launch" > $OUT/pytest.log 2>&1; echo "test rc=$?"; tail -25 $OUT/pytest.log | cut -c1-250
timeout 600 python scripts/exp_graph_capture.py 2>&1 | tee $OUT/graph.txt | tail -8
