#!/bin/bash
# round 6: is the ring kernel staging-bound on large tables?  producer depth / count variants (tools/build_variant.sh)
export PROBE_MODES=1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in . d6 d8 p8 p8d2; do
  if [ "$v" = "." ]; then unset PYMDE_AMD_LIB_VARIANT; else export PYMDE_AMD_LIB_VARIANT=$R/tools/variants/$v/libmde_hip.so; fi
  python tools/r6_cliff_probe.py base:1000000:50:2:uniform n2m:2000000:50:2:uniform n4m:4000000:50:2:uniform d3:1000000:50:3:uniform 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('$v', r.get('case'), r.get('kernel_ms'), r.get('ms_per_1e8_half_edges'), r.get('error',''))"
done
