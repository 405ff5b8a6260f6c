#!/bin/bash
# round 6: reaction time of the hand-shake on sparse streams (s_sleep 0 in the consumers' and producers' polls), with and without MDE_RING_FWORDS
export PROBE_MODES=auto
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in . sleep0 sleep0f; do
  if [ "$v" = "." ]; then unset PYMDE_AMD_LIB_VARIANT; else export PYMDE_AMD_LIB_VARIANT=$R/tools/variants/$v/libmde_hip.so; fi
  python tools/r6_cliff_probe.py base:1000000:50:2:uniform n2m:2000000:50:2:uniform n4m:4000000:50:2:uniform d3:1000000:50:3:uniform 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('$v', r.get('case'), r.get('kernel_ms'), r.get('ms_per_1e8_half_edges'), r.get('error',''))"
done
