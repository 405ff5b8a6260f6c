#!/bin/bash
# r6_widep.sh -- correctness of the pipelined general-d kernel, then tools/d128_locality.py old / new / chunk sweep
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; O=gpurun_out/r6_widep.txt; : > $O
python tools/r6_widep_check.py >> $O 2>&1 || echo "CHECK FAILED" >> $O
for P in 0 1; do echo "== MDE_WIDE_P=$P" >> $O; MDE_WIDE_P=$P python tools/d128_locality.py >> $O 2>&1; done
for C in 9 10 12 13; do echo "== MDE_WIDE_P=1 MDE_WIDE_CHUNK=$C" >> $O; MDE_WIDE_CHUNK=$C python tools/d128_locality.py >> $O 2>&1; done
echo "== d = 256" >> $O; LOC_D=256 python tools/d128_locality.py >> $O 2>&1
tail -80 $O
