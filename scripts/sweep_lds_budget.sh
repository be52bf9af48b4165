#!/bin/bash
# headline step time and grid-backward stage times for a range of owner-slice LDS budgets (bench.py --lds-budget)
for b in 32768 49152 65536 98304 131072; do
  timeout 200 python bench.py --no-cpu-baseline --steps 300 --warmup 50 --lds-budget $b 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stages_ms']
print($b, round(d['ms_per_step'], 4), 'scatter', round(s['grid_backward_scatter'], 4), 'owner', round(s['grid_backward'], 4))"
done
