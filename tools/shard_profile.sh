#!/bin/bash
# shard_profile.sh -- per-rank kernel of a W-way vertex-range shard, one rank emulated on one GPU:
# bench.py --emulate-world W (HIP events around the evaluation: ring kernel + combine + launch gaps)
# and a rocprofv3 --kernel-trace --stats pass of the same command (kernel durations alone).
# Results under gpurun_out/shard/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/shard
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for W in 2 4 8; do
  python $R/bench.py --emulate-world $W --steps 200 --no-cpu-baseline 2> $OUT/W$W.err | tail -1 > $OUT/W${W}_bench_line.json
  rocprofv3 --kernel-trace --stats -d $OUT/prof$W --output-format csv -- python $R/bench.py --emulate-world $W --steps 100 --no-cpu-baseline > /dev/null 2> $OUT/prof$W.err
  f=$(find $OUT/prof$W -name "*kernel_stats.csv" | head -1)
  # the ring and combine kernels only (the rest is problem generation and the plan build)
  (head -1 $f; grep -E "k_fused_ring|k_ring_combine" $f) > $OUT/W${W}_kernel_stats.csv
  rm -rf $OUT/prof$W
done
cd $R && python - <<'PY'
import csv, glob, json, os
out = {}
for W in (2, 4, 8):
    line = json.load(open("gpurun_out/shard/W%d_bench_line.json" % W))
    rows = list(csv.DictReader(open("gpurun_out/shard/W%d_kernel_stats.csv" % W)))
    ring = [r for r in rows if "k_fused_ring" in r["Name"] and "true, true, true" in r["Name"]]
    comb = [r for r in rows if "k_ring_combine" in r["Name"]]
    out[str(W)] = {
        "events_ms_per_evaluation": line["roofline"]["kernel_ms"],
        "ms_per_step_back_to_back": line["ms_per_step"],
        "rocprof_ring_kernel_ms": float(ring[0]["AverageNs"]) * 1e-6 if ring else None,
        "rocprof_combine_kernel_ms": float(comb[0]["AverageNs"]) * 1e-6 if comb else None,
        "parallelism": line["config"]["parallelism"],
    }
json.dump(out, open("gpurun_out/shard/shard_emulation.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
