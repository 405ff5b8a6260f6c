#!/bin/bash
# r6_lowdeg.sh -- large tables at low degree (n = 1M / 2M, out-degree 5 .. 20, d = 2): auto, CSR forced, ring forced
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; O=gpurun_out/r6_lowdeg.txt; : > $O
for N in 1000000 2000000; do for DEG in 5 10 20; do for P in auto 0 1; do
  if [ $P = auto ]; then unset MDE_PANEL; else export MDE_PANEL=$P; fi
  python bench.py --no-cpu-baseline --n $N --blocks 5 --degree $DEG --steps 100 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); c = r['config']; rl = c.get('ring_layout') or {}
print('panel=%-4s n=%7d deg=%2d  %.4f ms per evaluation  %.3f ms per 1e8 half-edges  %s' % ('$P', $N, $DEG, r['ms_per_step'], c.get('ms_per_1e8_half_edges', 0.0), 'ring %dx%d R=%d' % (rl.get('row_blocks', 0), rl.get('col_groups', 0), rl.get('rows_per_block', 0)) if rl.get('built') else 'CSR'))" >> $O 2>&1
done; done; done
cat $O
