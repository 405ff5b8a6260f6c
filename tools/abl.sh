mkdir -p gpurun_out/c19; export MDE_PANEL=1 MDE_RING_STATS=1
run() { echo "== $1 DBG=$2 n=$4" >> gpurun_out/c19/abl.txt; LD_LIBRARY_PATH=$3 MDE_RING_DBG=$2 timeout 60 ./tools/kbench $4 $5 10 2>&1 | grep -E "mde ring|codebook stream \(4|fused Log1p d=2 G|check|rc=|rror" >> gpurun_out/c19/abl.txt; }
for v in s4d2 s5d2 s6d2 s4d1 s6d1 s8d1 s6d0; do run $v 0 tools/variants/$v 1000000 50; done
cat gpurun_out/c19/abl.txt
