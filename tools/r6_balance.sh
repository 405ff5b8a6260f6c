#!/bin/bash
# r6_balance.sh -- the sweep-balance figure of the ring layout (MDE_RING_STATS=1) and the evaluation time in auto mode
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; O=gpurun_out/r6_balance.txt; : > $O
run() { # label, bench args
  L=$1; shift
  MDE_RING_STATS=1 python bench.py --no-cpu-baseline --blocks 5 --steps 50 "$@" 2> /tmp/bal.err | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); c = r['config']; rl = c.get('ring_layout') or {}
print('%-34s %.4f ms per evaluation  %s' % ('$L', r['ms_per_step'], 'ring %dx%d its %d' % (rl.get('row_blocks', 0), rl.get('col_groups', 0), rl.get('iterations', 0)) if rl.get('built') else 'CSR'))" >> $O 2>&1
  grep "sweep balance" /tmp/bal.err | tail -1 >> $O
}
run "uniform 1M deg50"
run "uniform 1M deg50 d3" --dim 3
run "uniform 2M deg50" --n 2000000
run "4b pushpull 1M" --variant 4b
run "hub 1M" --graph hub
run "powerlaw 1M" --graph powerlaw
run "clusters 1M deg50" --graph clusters
run "clusters 1M deg20" --graph clusters --degree 20
for N in 100000 300000; do for DEG in 50 20; do
  run "uniform $N deg$DEG" --n $N --degree $DEG
  run "clusters $N deg$DEG" --n $N --degree $DEG --graph clusters
  run "powerlaw $N deg$DEG" --n $N --degree $DEG --graph powerlaw
done; done
python bench.py --config 3 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('config 3', r['ms_per_step'], r['config']['kernel_layout'])" >> $O
python bench.py --config 2 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('config 2', r['ms_per_step'], r['config']['kernel_layout'])" >> $O
cat $O
