#!/bin/bash
# round 6: pair window (MDE_RING_SPAN) and bank-class cap (MDE_RING_CAP) on sparse streams at the tallest row blocks, ring forced
export PROBE_MODES=1 MDE_RING_STATS=1
run() { # case span cap
  MDE_RING_SPAN=$2 MDE_RING_CAP=$3 python tools/r6_cliff_probe.py $1 2> /tmp/err.txt | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('$1 span=$2 cap=$3', r.get('kernel_ms'), r.get('ms_per_1e8_half_edges'), r.get('error',''))"
  grep "mde ring\] d=" /tmp/err.txt | sed 's/.*placement 1: //; s/; loss terms.*//'
}
for sc in "4 4" "4 64" "5 4" "5 64" "6 4" "6 64" "7 64"; do run n2m:2000000:50:2:uniform $sc; done
for sc in "4 64" "5 64" "6 64" "7 64"; do run n4m:4000000:50:2:uniform $sc; done
for sc in "6 4" "6 64" "7 64" "8 64" "9 64"; do run d3:1000000:50:3:uniform $sc; done
for sc in "4 64" "5 4" "5 64"; do run base:1000000:50:2:uniform $sc; done
