#!/bin/bash
# prof_cmd.sh NAME cmd... -- rocprofv3 kernel stats of a command on the GPU box; the per-kernel summary lands in gpurun_out/NAME_kernel_stats.csv
name=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$name --output-format csv -- "$@" > $R/gpurun_out/prof_$name.out 2>&1
cd $R
f=$(find gpurun_out/prof_$name -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f gpurun_out/${name}_kernel_stats.csv && python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:18]:
    print("%-90s calls %6s  avg %9.1f us  total %9.3f ms  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
