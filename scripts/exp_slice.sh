run() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e6,1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stages_ms'].items() if v>0})"; }
run default
TCNN_GRID_LDS_SLICE_BYTES=65536 run slice64k
TCNN_GRID_LDS_SLICE_BYTES=32768 run slice32k
run default
