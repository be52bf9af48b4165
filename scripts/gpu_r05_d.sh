#!/bin/bash
# Round 5, fourth GPU call: fragment-wide activations (parity + the spilling-instance comparison again), headline sanity, the fixed sweep test.
OUT=$PWD/gpurun_out/r05d; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "activation or owner_pass or network or wide or module" > $OUT/pytest_act.log 2>&1; echo "act rc=$? $(tail -1 $OUT/pytest_act.log)"
timeout 300 python scripts/exp_spilling_instances.py > $OUT/spilling_instances.txt 2>&1; cut -c1-150 $OUT/spilling_instances.txt
for i in 1 2; do timeout 120 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --api native --no-inference 2>$OUT/base_$i.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline', round(d['ms_per_step'],4), 'resident', round(d.get('ms_per_step_resident',0),4), {k:round(v,4) for k,v in d['stages_ms'].items() if v>0})"; done
timeout 200 python bench.py --workload stress --steps 100 --warmup 20 --no-cpu-baseline --api native --no-inference 2>$OUT/stress.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stress', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stages_ms'].items() if v>0})"
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head
echo done
