#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; O=gpurun_out/r6_midsize2.txt; : > $O
for P in auto 1; do for D in 2; do for N in 20000 50000 100000 150000; do for DEG in 50 20; do
  if [ $P = 1 ]; then export MDE_PANEL=1; else unset MDE_PANEL; fi
  python bench.py --no-cpu-baseline --n $N --dim $D --blocks 5 --degree $DEG 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); c = r['config']; rl = c.get('ring_layout') or {}
print('panel=$P d=$D n=%7d deg=$DEG  %.4f ms per evaluation  %.3f ms per 1e8 half-edges  %s' % ($N, r['ms_per_step'], c.get('ms_per_1e8_half_edges', 0.0), 'ring %dx%d R=%d' % (rl.get('row_blocks', 0), rl.get('col_groups', 0), rl.get('rows_per_block', 0)) if rl.get('built') else 'CSR'))" >> $O 2>&1
done; done; done; done
cat $O
