#!/bin/bash
# r6_records.sh -- the round-6 records kept under profiles/: PMC + kernel stats of the headline bench and of three
# secondaries (4b, d = 3, n = 2M), the default bench line (with the CPU baseline), the driver's command line, the regimes
# beyond the old feasibility rule, embed records, shard emulations
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r6rec; rm -rf $O; mkdir -p $O
cd $R
pmc() { # tag, bench args...
  tag=$1; shift
  bash tools/pmc_traffic.sh "$@" > $O/pmc_run$tag.log 2>&1
  cp gpurun_out/pmc/pmc_traffic.json $O/pmc_traffic$tag.json 2>/dev/null
  cp gpurun_out/pmc/kernel_stats.csv $O/kernel_stats$tag.csv 2>/dev/null
  cp gpurun_out/pmc/pmc_traffic.json $R/profiles/r06_pmc_traffic$tag.json 2>/dev/null   # (the bench lines below pick it up)
}
pmc ""
pmc _4b --variant 4b
pmc _d3 --dim 3
pmc _n2m --n 2000000
pmc _c5 --config 5
python bench.py 2>/dev/null | tail -1 > $O/bench_line.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_line_driver_command.json
python bench.py --no-cpu-baseline --survey-seed 2>/dev/null | tail -1 > $O/survey_seed_bench_line.json
python bench.py --no-cpu-baseline --variant 4b 2>/dev/null | tail -1 > $O/config4b_bench_line.json
python bench.py --no-cpu-baseline --dim 3 2>/dev/null | tail -1 > $O/d3_bench_line.json
python bench.py --no-cpu-baseline --n 2000000 2>/dev/null | tail -1 > $O/n2m_bench_line.json
python bench.py --no-cpu-baseline --n 4000000 --steps 50 2>/dev/null | tail -1 > $O/n4m_bench_line.json
python bench.py --no-cpu-baseline --graph hub 2>/dev/null | tail -1 > $O/hub_bench_line.json
python bench.py --no-cpu-baseline --graph powerlaw 2>/dev/null | tail -1 > $O/powerlaw_bench_line.json
python bench.py --no-cpu-baseline --graph powerlaw --dim 3 2>/dev/null | tail -1 > $O/powerlaw_d3_bench_line.json
python bench.py --config 4 --embed 2>/dev/null | tail -1 > $O/config4_embed_bench_line.json
python bench.py --config 4 --embed --graph clusters 2>/dev/null | tail -1 > $O/config4_embed_clusters_bench_line.json
python bench.py --config 2 2>/dev/null | tail -1 > $O/config2_bench_line.json
python bench.py --config 3 2>/dev/null | tail -1 > $O/config3_bench_line.json
python bench.py --config 5 2>/dev/null | tail -1 > $O/config5_bench_line.json
for W in 2 4 8; do
  python bench.py --no-cpu-baseline --emulate-world $W --steps 50 --blocks 5 2>/dev/null | tail -1 > $O/shard_W$W.json
  python bench.py --config 4 --embed --emulate-world $W --steps 30 2>/dev/null | tail -1 > $O/config4_embed_shard_W$W.json
done
ls -la $O
