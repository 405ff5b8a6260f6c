#!/bin/bash
# r5_steptrace.sh -- back-to-back evaluations of the headline bench under rocprofv3: kernel durations and the gaps between launches
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/steptrace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/p --output-format csv -- python $R/bench.py --steps 50 --blocks 3 --no-cpu-baseline > $OUT/line.json 2> $OUT/err.txt
cd $R
python - <<'PY'
import csv, glob, statistics
f = glob.glob("gpurun_out/steptrace/p/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = {}
seq = []
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    seq.append((s, e, r["Kernel_Name"][:50]))
# the longest run of consecutive codebook-stream ring kernels
ring = [i for i, x in enumerate(seq) if "k_fused_ring" in x[2]]
runs, cur = [], [ring[0]]
for a, b in zip(ring, ring[1:]):
    if b == a + 1: cur.append(b)
    else: runs.append(cur); cur = [b]
runs.append(cur)
best = max(runs, key=len)
durs = [(seq[i][1] - seq[i][0]) / 1e3 for i in best]
gaps = [(seq[b][0] - seq[a][1]) / 1e3 for a, b in zip(best, best[1:])]
per = [(seq[b][0] - seq[a][0]) / 1e3 for a, b in zip(best, best[1:])]
print("run of %d back-to-back ring launches: duration median %.1f us (min %.1f max %.1f), gap median %.2f us (min %.2f max %.2f), start-to-start median %.1f us"
      % (len(best), statistics.median(durs), min(durs), max(durs), statistics.median(gaps), min(gaps), max(gaps), statistics.median(per)))
others = sorted({x[2] for x in seq if "k_fused_ring" not in x[2]})
print("other kernels in the trace:", len(others))
PY
tail -c 600 $OUT/line.json | head -c 300; echo
