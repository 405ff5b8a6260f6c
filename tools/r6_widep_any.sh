#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; O=gpurun_out/r6_widep_any.txt; : > $O
python tools/r6_widep_check.py 2>&1 | grep -v amdgpu.ids | tail -4 >> $O || echo "CHECK FAILED" >> $O
for D in 6 10 50 12 128; do LOC_D=$D python tools/d128_locality.py 2>&1 | grep -v amdgpu.ids | grep -E "^d =|uniform|   1000:|    100:" >> $O; done
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "pipelined or processing_order or all_dims" 2>&1 | tail -3 >> $O
cat $O
