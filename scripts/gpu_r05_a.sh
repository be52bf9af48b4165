#!/bin/bash
# Round 5, first GPU call: the multi-GPU fixes under test, the pipelined backward (parity + A/B), the bench's new legs.
OUT=$PWD/gpurun_out/r05a; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_distributed.py -x -q > $OUT/pytest_dist.log 2>&1; echo "dist rc=$? $(tail -1 $OUT/pytest_dist.log)"
timeout 600 python -m pytest tests/test_gpu_parity_full.py -x -q -k "pipelined_over or reading_the_master or written_through" > $OUT/pytest_par.log 2>&1; echo "parity rc=$? $(tail -1 $OUT/pytest_par.log)"
for g in 1 2 3 4 6 8 16; do
  TCNN_BACKWARD_OVERLAP=$g timeout 120 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --api native --no-inference 2>$OUT/ov_$g.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap $g', round(d['ms_per_step'],4), 'resident', round(d.get('ms_per_step_resident',0),4), {k:round(v,4) for k,v in d['stages_ms'].items() if v>0})" >> $OUT/overlap.txt 2>&1
done
cat $OUT/overlap.txt
timeout 900 python -m pytest tests/test_gpu_bench.py -x -q > $OUT/pytest_bench.log 2>&1; echo "bench tests rc=$? $(tail -1 $OUT/pytest_bench.log)"
timeout 200 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $OUT/bench_1000.json 2>$OUT/bench_1000.err
timeout 200 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --resident-first > $OUT/bench_1000_rf.json 2>>$OUT/bench_1000.err
python - <<PY
import json
for f in ("bench_1000","bench_1000_rf"):
    try:
        d=json.load(open("$OUT/%s.json"%f)); print(f, d["ms_per_step"], d["ms_per_step_resident"], d.get("torch_binding",{}).get("ms_per_step"), d["protocol"]["adam_touched_fraction"], d["roofline"]["stages"]["adam"])
    except Exception as e: print(f, "failed", e)
PY
timeout 200 python bench.py --workload hash_shipped --steps 1000 --warmup 100 > $OUT/bench_shipped.json 2>$OUT/bench_shipped.err; cut -c1-400 $OUT/bench_shipped.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_shipped -o trace -- python $OLDPWD/bench.py --workload hash_shipped --steps 50 --warmup 10 --no-cpu-baseline --api native > $OUT/rocprof_shipped.log 2>&1 )
for f in $(find $OUT/prof_shipped -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats_shipped.csv; cut -c1-160 $f | head -12; done
find $OUT/prof_shipped -name "*kernel_trace.csv" -delete; find $OUT/prof_shipped -name "*.db" -delete
echo done
