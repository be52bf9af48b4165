#!/bin/bash
# Copies the summaries of an evidence run (scripts/gpu_r06_final.sh <tag>, merged back under gpurun_out/<tag>/) into profiles/ under the
# round's names and regenerates profiles/traffic.json from its PMC passes.  Usage: scripts/collect_evidence.sh <tag> [round prefix, default r06]
set -e
cd "$(dirname "$0")/.."
tag=${1:?tag}; r=${2:-r06}
src=gpurun_out/$tag
cpy() { [ -s "$src/$1" ] && cp "$src/$1" "profiles/${r}_$2" || echo "missing: $src/$1"; }
cpy bench_n1.json bench_n1.json
cpy driver_1.json bench_driver_cmd.json
cpy driver_2.json bench_driver_cmd_2.json
cpy bench_mlp.json bench_mlp.json
cpy bench_stress.json bench_stress.json
cpy bench_hash_shipped.json bench_shipped.json
cpy bench_hash_bf16.json bench_hash_bf16.json
cpy bench_stress_bf16.json bench_stress_bf16.json
cpy bench_batch_65536.json bench_batch_65536.json
cpy bench_batch_16384.json bench_batch_16384.json
cpy kernel_stats.csv kernel_stats.csv
cpy kernel_stats_mlp.csv kernel_stats_mlp.csv
cpy kernel_stats_stress.csv kernel_stats_stress.csv
cpy kernel_stats_hash_shipped.csv kernel_stats_shipped.csv
cpy pmc_summary.txt pmc_summary.txt
cpy pmc_summary.json pmc_summary.json
cpy soak_first_steps.txt soak_first_steps.txt
cpy lscpu.txt lscpu.txt
cpy rocminfo.txt rocminfo.txt
cpy COMMIT.txt COMMIT.txt
[ -s "$src/pytest.log" ] && tail -n 40 "$src/pytest.log" > "profiles/${r}_gpu_tests.txt"
[ -s "$src/pmc_summary.json" ] && python scripts/make_traffic.py "$src/pmc_summary.json" > profiles/traffic.json
ls -la profiles | grep "${r}_" | wc -l
