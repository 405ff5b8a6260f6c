#!/bin/bash
# r5_records.sh -- the round-5 records kept under profiles/: PMC + kernel stats of the headline bench, the default
# bench line (with the CPU baseline), --survey-seed, variant 4b, the function sweep, config 4 / 5 as embed(),
# configs 2 and 3, shard emulations of configs 4 and 5, the kernel sequence of a config-4 iteration
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5rec; rm -rf $O; mkdir -p $O
cd $R
bash tools/pmc_traffic.sh > $O/pmc_run.log 2>&1
cp gpurun_out/pmc/pmc_traffic.json gpurun_out/pmc/kernel_stats.csv $O/ 2>/dev/null
cp $R/gpurun_out/pmc/pmc_traffic.json $R/profiles/r05_pmc_traffic.json 2>/dev/null   # (the bench lines below pick it up)
python bench.py 2>/dev/null | tail -1 > $O/bench_line.json
python bench.py --no-cpu-baseline --survey-seed 2>/dev/null | tail -1 > $O/survey_seed_bench_line.json
python bench.py --no-cpu-baseline --variant 4b 2>/dev/null | tail -1 > $O/config4b_bench_line.json
python bench.py --config 4 --embed 2>/dev/null | tail -1 > $O/config4_embed_bench_line.json
bash tools/r5_c4trace.sh > $O/config4_iteration_sequence.log 2>&1
cp gpurun_out/c4trace/sequence.txt $O/config4_iteration_sequence.txt 2>/dev/null
cp gpurun_out/c4trace/kernel_stats.csv $O/config4_embed_kernel_stats.csv 2>/dev/null
python bench.py --config 2 2>/dev/null | tail -1 > $O/config2_bench_line.json
python bench.py --config 3 2>/dev/null | tail -1 > $O/config3_bench_line.json
python bench.py --config 5 2>/dev/null | tail -1 > $O/config5_bench_line.json
python bench.py --config 5 --embed 2>/dev/null | tail -1 > $O/config5_embed_bench_line.json
for W in 2 4 8; do
  python bench.py --no-cpu-baseline --emulate-world $W --steps 50 --blocks 5 2>/dev/null | tail -1 > $O/shard_W$W.json
  python bench.py --config 5 --emulate-world $W --steps 20 --blocks 5 2>/dev/null | tail -1 > $O/config5_shard_W$W.json
done
bash tools/r5_functions.sh > /dev/null 2>&1
cp gpurun_out/r05_function_sweep.txt $O/function_sweep.txt
ls -la $O
