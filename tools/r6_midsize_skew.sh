#!/bin/bash
# r6_midsize_skew.sh -- the mid-regime layout choice on skewed graphs (preferential attachment, one hub): auto against CSR forced
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; O=gpurun_out/r6_midsize_skew.txt; : > $O
for G in powerlaw hub clusters; do for N in 100000 200000 300000; do for DEG in 50 20; do for P in auto 0; do
  if [ $P = auto ]; then unset MDE_PANEL; else export MDE_PANEL=$P; fi
  python bench.py --no-cpu-baseline --n $N --blocks 5 --degree $DEG --graph $G 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); c = r['config']; rl = c.get('ring_layout') or {}
print('%-9s panel=%-4s n=%7d deg=%2d  %.4f ms per evaluation  %s' % ('$G', '$P', $N, $DEG, r['ms_per_step'], 'ring %dx%d R=%d hubs %d dealt %s' % (rl.get('row_blocks', 0), rl.get('col_groups', 0), rl.get('rows_per_block', 0), rl.get('hub_rows', 0), rl.get('permuted')) if rl.get('built') else 'CSR'))" >> $O 2>&1
done; done; done; done
cat $O
