#!/bin/bash
# r5_c4trace.sh -- the kernel sequence of steady-state embed() iterations at config 4 (Centered): start offsets, durations, gaps
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/c4trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/p --output-format csv -- python $R/bench.py --config 4 --embed --steps 30 > $OUT/line.json 2> $OUT/err.txt
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/c4trace/p/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_fused_ring" in r["Kernel_Name"]]
# the Centered solve comes first: take three consecutive iterations out of its steady state
lo = idx[40]
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
with open("gpurun_out/c4trace/sequence.txt", "w") as out:
    for r in rows[lo:idx[43] + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        out.write("%9.1f us  gap %5.1f  +%6.1f us  %s\n" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:70]))
        prev_end = e
print(open("gpurun_out/c4trace/sequence.txt").read())
PY
f=$(find $OUT/p -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv
