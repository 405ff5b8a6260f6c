#!/bin/bash
# round 6: ring geometry (rows per block -> ring slots -> pair window) on sparse streams, ring forced
export PROBE_MODES=1 MDE_RING_STATS=1
run() { # case rows span
  MDE_RING_ROWS=$2 MDE_RING_SPAN=$3 python tools/r6_cliff_probe.py $1 2> /tmp/err.txt | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('$1 R=$2 span=$3', r.get('kernel_ms'), r.get('ms_per_1e8_half_edges'), r.get('error',''))"
  grep "mde ring\] d=" /tmp/err.txt | sed 's/.*placement 1: //'
}
for rs in "7816 4" "6912 6" "6016 7" "6016 8" "5120 9" "5120 10" "4096 11" "4096 12"; do run n2m:2000000:50:2:uniform $rs; done
for rs in "7816 4" "6016 8" "5120 10" "4096 12" "4096 13" "3072 14" "3072 15"; do run n4m:4000000:50:2:uniform $rs; done
for rs in "5200 6" "4608 8" "4096 10" "3584 12" "3072 13" "3072 14"; do run d3:1000000:50:3:uniform $rs; done
