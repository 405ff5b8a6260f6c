#!/bin/bash
# abl.sh -- timing ablations of the LDS-ring kernel on the GPU box (build the probe library first:
#   tools/build_variant.sh abl "-DMDE_RING_ABLATE=1").  MDE_RING_DBG bits: 1 consumers never wait,
#   2 no staging, 4 no evaluation, 8 producers never wait for a slot, 64 consumers skip, 128 producers skip,
#   512 shader-clock probes per role (tools/r3_probe.sh), 1024 packed words from L2, 2048 next accumulators read before this iteration's write,
#   256 print the distribution of the workgroups' main-phase durations (shader clocks; measured: 527k / 544k / 569k
#   min / mean / max at config 4, no XCD skew -- every CU runs at the same per-iteration rate, there is no tail).
mkdir -p gpurun_out/abl; export MDE_PANEL=1 MDE_RING_STATS=1
for dbg in 0 1 2 3 4 64 194; do
  echo "== MDE_RING_DBG=$dbg" >> gpurun_out/abl/abl.txt
  LD_LIBRARY_PATH=tools/variants/abl MDE_RING_DBG=$dbg timeout 60 ./tools/kbench 1000000 50 10 2>&1 | grep -E "mde ring|codebook stream \(4|fused Log1p d=2 G|forward-only|check" >> gpurun_out/abl/abl.txt
done
cat gpurun_out/abl/abl.txt
