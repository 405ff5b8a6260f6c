#!/bin/bash
# pmc_traffic.sh [bench args] -- rocprofv3 passes of bench.py on the GPU box: kernel stats + HBM-side counters
# (separate --pmc passes, as MI355X_MICROARCH.md prescribes).  Results under gpurun_out/pmc/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench_stats_line.json 2> $OUT/stats.err
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$tag --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > /dev/null 2> $OUT/pmc_$tag.err
done
cd $R && python tools/pmc_traffic.py $OUT $OUT/pmc_traffic.json
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
