#!/bin/bash
# round 6: producers publishing independently (-DMDE_RING_FWORDS=1, tools/variants/fwords) against the in-order chain
export PROBE_MODES=auto
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in . fwords . fwords; do
  if [ "$v" = "." ]; then unset PYMDE_AMD_LIB_VARIANT; else export PYMDE_AMD_LIB_VARIANT=$R/tools/variants/$v/libmde_hip.so; fi
  python tools/r6_cliff_probe.py base:1000000:50:2:uniform n2m:2000000:50:2:uniform n4m:4000000:50:2:uniform d3:1000000:50:3:uniform hub:1000000:50:2:hub ba:1000000:50:2:ba 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('$v', r.get('case'), r.get('kernel_ms'), r.get('ms_per_1e8_half_edges'), 'loss', r.get('loss'), r.get('error',''))"
done
tools/r5_b2b.sh . fwords . fwords
B2B_ARGS="--variant 4b" tools/r5_b2b.sh . fwords
B2B_ARGS="--no-codebook" tools/r5_b2b.sh . fwords
