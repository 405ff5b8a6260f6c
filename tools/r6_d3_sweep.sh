#!/bin/bash
# round 6: d = 3 on the ring kernel -- chunk width (256 / 512 columns: tools/variants/c3_512) and column groups per row block (MDE_RING_Q)
export PROBE_MODES=auto MDE_RING_STATS=1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() { # variant Q case
  if [ "$1" = "." ]; then unset PYMDE_AMD_LIB_VARIANT; else export PYMDE_AMD_LIB_VARIANT=$R/tools/variants/$1/libmde_hip.so; fi
  if [ "$2" = "-" ]; then unset MDE_RING_Q; else export MDE_RING_Q=$2; fi
  python tools/r6_cliff_probe.py $3 2> /tmp/err.txt | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('$1 Q=$2 $3', r.get('kernel_ms'), r.get('ms_per_1e8_half_edges'), 'loss', r.get('loss'), r.get('error',''))"
  grep "mde ring\] d=" /tmp/err.txt | sed 's/.*mde ring\] //; s/; loss terms.*//'
}
for c in d3:1000000:50:3:uniform d3s:250000:100:3:uniform d3ba:1000000:50:3:ba; do
  for v in . c3_512; do run $v - $c; done
done
for q in 2 3 5 9; do run . $q d3:1000000:50:3:uniform; run c3_512 $q d3:1000000:50:3:uniform; done
