# A/B of library variants on the bench: bash scripts/ab_variant.sh base variantA variantB ...
for v in "$@"; do
  if [ $v = base ]; then unset TCNN_HIP_LIBRARY; else export TCNN_HIP_LIBRARY=$PWD/tiny-cuda-nn_amd/lib/variants/$v.so; fi
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stages_ms'].items() if v>0})"
done
