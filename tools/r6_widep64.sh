#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; O=gpurun_out/r6_widep64.txt; : > $O
python tools/r6_widep_check.py >> $O 2>&1 || echo "CHECK FAILED" >> $O
for P in 0 1; do echo "== d = 64 MDE_WIDE_P=$P" >> $O; LOC_D=64 MDE_WIDE_P=$P python tools/d128_locality.py >> $O 2>&1; done
echo "== d = 64 shuffled MDE_WIDE_P=1" >> $O; LOC_SHUFFLE=1 LOC_D=64 python tools/d128_locality.py >> $O 2>&1
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "pipelined or processing_order" 2>&1 | tail -3 >> $O
tail -45 $O
