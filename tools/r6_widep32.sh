#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; O=gpurun_out/r6_widep32.txt; : > $O
python tools/r6_widep_check.py >> $O 2>&1 || echo "CHECK FAILED" >> $O
for P in 0 1; do echo "== d = 32 MDE_WIDE_P=$P" >> $O; LOC_D=32 MDE_WIDE_P=$P python tools/d128_locality.py >> $O 2>&1; done
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "pipelined or processing_order or all_dims" 2>&1 | tail -3 >> $O
tail -34 $O
