#!/bin/bash
# r6_clusters.sh -- planted-cluster graphs (2/3 of the links inside clusters of 1000 consecutive items) on the ring kernel:
# default row -> wave map (contiguous ranges), the sweep-balanced map (MDE_RING_ASSIGN=1), the CSR kernels
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; O=gpurun_out/r6_clusters.txt; : > $O
for N in 100000 300000 1000000; do for DEG in 50 20; do for MODE in "auto" "assign" "csr" "ring" "ring+assign"; do
  unset MDE_PANEL MDE_RING_ASSIGN
  case $MODE in assign) export MDE_RING_ASSIGN=1;; csr) export MDE_PANEL=0;; ring) export MDE_PANEL=1;; ring+assign) export MDE_PANEL=1 MDE_RING_ASSIGN=1;; esac
  python bench.py --no-cpu-baseline --n $N --blocks 5 --degree $DEG --graph clusters --steps 50 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); c = r['config']; rl = c.get('ring_layout') or {}
print('clusters %-11s n=%7d deg=%2d  %.4f ms per evaluation  %s' % ('$MODE', $N, $DEG, r['ms_per_step'], 'ring %dx%d R=%d its %d' % (rl.get('row_blocks', 0), rl.get('col_groups', 0), rl.get('rows_per_block', 0), rl.get('iterations', 0)) if rl.get('built') else 'CSR'))" >> $O 2>&1
done; done; done
cat $O
