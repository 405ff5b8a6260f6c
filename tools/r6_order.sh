#!/bin/bash
# r6_order.sh -- the breadth-first processing order: correctness with an order forced on ragged graphs, then the window
# graphs of tools/d128_locality.py under a random renumbering, without and with the order; build time of the order
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; O=gpurun_out/r6_order.txt; : > $O
CHECK_ORDER=1 python tools/r6_widep_check.py >> $O 2>&1 || echo "CHECK FAILED" >> $O
echo "== shuffled, MDE_ROW_ORDER=0" >> $O; LOC_SHUFFLE=1 MDE_ROW_ORDER=0 python tools/d128_locality.py >> $O 2>&1
echo "== shuffled, MDE_ROW_ORDER=1" >> $O; LOC_SHUFFLE=1 python tools/d128_locality.py >> $O 2>&1
echo "== as given, MDE_ROW_ORDER=1" >> $O; python tools/d128_locality.py >> $O 2>&1
python - >> $O 2>&1 <<'P'
import time, torch, sys
sys.path.insert(0, ".")
from pymde_amd.average_distortion import EdgePlan
dev = torch.device("cuda", 0); g = torch.Generator(device=dev); g.manual_seed(0)
n, deg = 500000, 40
src = torch.arange(n, device=dev).repeat_interleave(deg)
for window in (100, 1000, 10000):
    off = torch.randint(1, window + 1, (n * deg,), device=dev, generator=g)
    dst = (src + off) % n
    perm = torch.randperm(n, device=dev, generator=g)
    e = perm[torch.stack([src, dst], 1)]
    e = torch.unique(torch.stack([e.min(1).values, e.max(1).values], 1), dim=0).contiguous()
    plan = EdgePlan(n, e); torch.cuda.synchronize()
    t0 = time.time(); info = plan.row_order(1); torch.cuda.synchronize(); t1 = time.time()
    print("window %d shuffled: order built in %.3f s: %s" % (window, t1 - t0, info))
P
tail -70 $O
