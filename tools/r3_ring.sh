#!/bin/bash
# r3_ring.sh -- round-3 ablations of the LDS-ring kernel on the GPU box (kbench through the C ABI)
out=gpurun_out/r3ring; mkdir -p $out; rm -f $out/abl.txt; export MDE_PANEL=1 MDE_RING_STATS=1
run() { # label, env...
  echo "== $1" >> $out/abl.txt; shift
  env "$@" timeout 120 ./tools/kbench 1000000 50 10 2>&1 | grep -E "mde ring|fused Log1p d=2|codebook stream|forward-only|check|error|Quadratic|PushPull" >> $out/abl.txt
}
run "default" X=1
[ -n "$R3_FULL" ] && run "cap 64, no placement" MDE_RING_CAP=64 MDE_RING_PLACE=0
if [ -d tools/variants/abl ]; then
for dbg in ${R3_DBGS:-512 513 2 3 4}; do
  run "ablate build, MDE_RING_DBG=$dbg" LD_LIBRARY_PATH=tools/variants/abl MDE_RING_DBG=$dbg
done
fi
cat $out/abl.txt
