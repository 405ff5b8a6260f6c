#!/bin/bash
# r3_probe.sh -- where the waves of the ring kernel spend their time (ablate build, MDE_RING_DBG & 512)
out=gpurun_out/r3probe; mkdir -p $out; export MDE_PANEL=1
for dbg in 512 516 513 520; do
  echo "== MDE_RING_DBG=$dbg" >> $out/probe.txt
  LD_LIBRARY_PATH=tools/variants/abl MDE_RING_DBG=$dbg timeout 120 ./tools/kbench 1000000 50 10 2>&1 | grep -E "mde ring probe|fused Log1p d=2 G" >> $out/probe.txt
done
cat $out/probe.txt
