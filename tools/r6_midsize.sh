#!/bin/bash
# r6_midsize.sh -- the evaluation between the MNIST-sized problems and the benchmark shape: n = 50k .. 750k at out-degree 50
# (uniform graph, d = 2 and 3, Log1p), which kernel the library picks and ms per 1e8 half-edges
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; O=gpurun_out/r6_midsize.txt; : > $O
for D in 2 3; do for N in 50000 100000 200000 300000 500000 750000; do
  python bench.py --no-cpu-baseline --n $N --dim $D --blocks 5 $MID_ARGS 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); c = r['config']; rl = c.get('ring_layout') or {}
print('d=$D n=%7d  %.4f ms per evaluation  %.3f ms per 1e8 half-edges  %s  %s' % ($N, r['ms_per_step'], c.get('ms_per_1e8_half_edges', 0.0), 'ring %dx%d' % (rl.get('row_blocks', 0), rl.get('col_groups', 0)) if rl.get('built') else 'CSR', c.get('parameter_stream', '')[:40]))" >> $O 2>&1
done; done
cat $O
