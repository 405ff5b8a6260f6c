#!/bin/bash
# r3_variants.sh "V1[:ENV=..,ENV=..]" ... -- kbench on compile-time variants of the ring kernel (tools/build_variant.sh);
# V = "." is the product library
out=gpurun_out/r3var; mkdir -p $out; rm -f $out/var.txt; export MDE_PANEL=1 MDE_RING_STATS=1
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  echo "== variant $v $envs" >> $out/var.txt
  lib=tools/variants/$v; [ "$v" = "." ] && lib=pymde_amd; [ -f $lib/libmde_hip.so ] || { echo "missing $lib/libmde_hip.so" >> $out/var.txt; continue; }
  env LD_LIBRARY_PATH=$lib $envs timeout 120 ./tools/kbench 1000000 50 10 2>&1 | grep -E "mde ring|fused Log1p d=2|check|error" >> $out/var.txt
done
cut -c1-420 $out/var.txt
