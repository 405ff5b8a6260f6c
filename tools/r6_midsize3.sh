#!/bin/bash
# r6_midsize3.sh -- auto mode after the mid-regime rule: the sweep of r6_midsize2.sh, d = 3, configs 2 / 3, the GPU tests
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; O=gpurun_out/r6_midsize3.txt; : > $O
for D in 2 3; do for N in 20000 50000 100000 150000 300000; do for DEG in 50 20; do
  python bench.py --no-cpu-baseline --n $N --dim $D --blocks 5 --degree $DEG 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); c = r['config']; rl = c.get('ring_layout') or {}
print('auto d=$D n=%7d deg=$DEG  %.4f ms per evaluation  %.3f ms per 1e8 half-edges  %s' % ($N, r['ms_per_step'], c.get('ms_per_1e8_half_edges', 0.0), 'ring %dx%d R=%d' % (rl.get('row_blocks', 0), rl.get('col_groups', 0), rl.get('rows_per_block', 0)) if rl.get('built') else 'CSR'))" >> $O 2>&1
done; done; done
for C in 2 3; do python bench.py --config $C 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('config $C', r['ms_per_step'], json.dumps(r['config'])[:600])" >> $O 2>&1; done
python -m pytest tests -x -q -m gpu 2>&1 | tail -12 >> $O
cat $O
