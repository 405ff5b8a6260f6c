#!/bin/bash
# r5_b2b.sh NAME... -- bench.py's back-to-back step time (no CPU baseline, 5 blocks of 200 steps) per library variant ("." = the product)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for name in "$@"; do
  if [ "$name" = "." ]; then unset PYMDE_AMD_LIB_VARIANT; else export PYMDE_AMD_LIB_VARIANT=$R/tools/variants/$name/libmde_hip.so; fi
  python $R/bench.py --no-cpu-baseline --blocks 5 $B2B_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s ms_per_step %.4f  blocks %s  kernel_ms %.4f' % ('$name', d['ms_per_step'], d['ms_per_step_blocks'], d['roofline']['kernel_ms']))"
done
