#!/bin/bash
# r5_seqtrace.sh "<bench.py arguments>" KERNEL_SUBSTRING [SKIP] -- kernel sequence (start, gap, duration) of three steady-state iterations of an embed() record
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/seqtrace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/p --output-format csv -- python $R/bench.py $1 > $OUT/line.json 2> $OUT/err.txt
cd $R
KSUB="$2" SKIP="${3:-40}" python - <<'PY'
import csv, glob, os
f = glob.glob("gpurun_out/seqtrace/p/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if os.environ["KSUB"] in r["Kernel_Name"]]
k = min(int(os.environ["SKIP"]), len(idx) - 4)
lo = idx[k]
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
for r in rows[lo:idx[k + 3] + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  gap %5.1f  +%6.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:70]))
    prev_end = e
PY
