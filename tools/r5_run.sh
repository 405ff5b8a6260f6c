#!/bin/bash
# r5_run.sh OUT "name:lib[:ENV=..,ENV=..]" ... -- kbench over library variants and graphs (round 5); lib "." = the product
out=gpurun_out/$1; shift; mkdir -p $out; rm -f $out/var.txt; export MDE_PANEL=1 MDE_RING_STATS=1 KB_LOG1P_ONLY=1
for spec in "$@"; do
  IFS=: read -r name lib envs <<< "$spec"
  [ "$lib" = "." ] && lib=pymde_amd || lib=tools/variants/$lib
  echo "== $name ($lib) $envs" >> $out/var.txt
  env LD_LIBRARY_PATH=$lib $(echo $envs | tr ',' ' ') timeout 60 ./tools/kbench 1000000 50 10 2>&1 | grep -E "mde ring|fused|layout_build|check|error|plan_build|graph" >> $out/var.txt
done
grep -E "^==|codebook stream \(4|fused Log1p d=2 G|probe\]|check|forward|iterations per|padding" $out/var.txt | cut -c1-260
