#!/bin/bash
# r4_vhdr.sh -- round 4, first question: does fetching the iteration headers with a vector load
# (instead of s_load, which the LDS waits drain) remove the exposed HBM round trip per stream block?
out=gpurun_out/r4a; mkdir -p $out; rm -f $out/*.txt; export MDE_PANEL=1 MDE_RING_STATS=1
run() {  # name lib env...
  local name=$1 lib=$2; shift 2
  echo "== $name $*" >> $out/var.txt
  env LD_LIBRARY_PATH=$lib "$@" timeout 120 ./tools/kbench 1000000 50 10 2>&1 | grep -E "mde ring|fused Log1p d=2|fused PushPull|layout_build|check|error|plan_build" >> $out/var.txt
}
run product pymde_amd
run shdr tools/variants/shdr
run ncw12 tools/variants/ncw12
run ncw14 tools/variants/ncw14
run pfb4 tools/variants/pfb4
for dbg in 129 4225 64 513 0; do run abl_vhdr tools/variants/abl MDE_RING_DBG=$dbg; done
for dbg in 129 4225; do run abl_shdr tools/variants/abl_s MDE_RING_DBG=$dbg; done
cut -c1-300 $out/var.txt
