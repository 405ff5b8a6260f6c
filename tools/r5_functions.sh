#!/bin/bash
# r5_functions.sh -- kernel time of every distortion function on the config-4 graph (ring kernel): profiles/r05_function_sweep.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
out=gpurun_out/r05_function_sweep.txt; rm -f $out
for f in log1p quadratic linear cubic huber log log1p2 l_huber l_quadratic logistic power power15 sigmoid hinge invpower logratio l_cubic l_power l_logistic l_fractional l_softfractional runtime; do
  python bench.py --no-cpu-baseline --function $f --steps 50 --blocks 3 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('%-18s ms_per_step %.4f  kernel_ms %.4f  stream %s | %s' % ('$f', r['ms_per_step'], r['roofline']['kernel_ms'], r['config']['parameter_stream'][:9], r['config']['workload'].split('d=2, ')[1].split(';')[0]))" >> $out
done
cat $out
