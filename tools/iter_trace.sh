#!/bin/bash
# iter_trace.sh -- the kernel sequence of one steady-state embed() iteration of config 2 (rocprofv3 kernel trace)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/itrace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/p --output-format csv -- python $R/bench.py --config 2 > $OUT/line.json 2> $OUT/err.txt
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/itrace/p/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 3 iterations: find the last occurrences of the fused kernel
idx = [i for i, r in enumerate(rows) if "k_fused_flat" in r["Kernel_Name"] or "k_fused_small" in r["Kernel_Name"]]
lo = idx[-4]
t0 = int(rows[lo]["Start_Timestamp"])
with open("gpurun_out/itrace/sequence.txt", "w") as out:
    for r in rows[lo:idx[-1] + 12]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        out.write("%9.1f us  +%6.1f us  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:90]))
print(open("gpurun_out/itrace/sequence.txt").read())
PY
f=$(find $OUT/p -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv
