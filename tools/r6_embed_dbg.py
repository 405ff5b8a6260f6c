# r6_embed_dbg.py -- the n = 20k problem of tools/r6_prob.py (the reference descends on it at d = 8 / 16 under both
# constraints: 1.07672 -> 0.50055 Standardized, 1.07702 -> 0.47643 Centered at d = 8): what this package does
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pymde_amd
from r6_prob import problem
dev = torch.device("cuda", 0)
e, w = problem()
n = 20000
for d in [int(a) for a in sys.argv[1:]] or [2, 8, 16]:
    for cname, c in (("Standardized", pymde_amd.Standardized()), ("Centered", pymde_amd.Centered())):
        torch.manual_seed(0)
        f = pymde_amd.penalties.PushAndPull(torch.tensor(w, device=dev), pymde_amd.penalties.Log1p, pymde_amd.penalties.Log)
        mde = pymde_amd.MDE(n, d, torch.tensor(e, device=dev), f, constraint=c, device=dev)
        mde.embed(max_iter=60, eps=0.0, verbose=bool(os.environ.get("DBG_VERBOSE")))
        s = mde.solve_stats
        print("ours d=%d %-12s distortion %.5f -> %.5f (%d its)  residual norms %s  step sizes %s" % (
            d, cname, s.average_distortions[0], float(mde.value), s.iterations,
            ["%.2e" % x for x in s.residual_norms[:4]], ["%.2e" % x for x in s.step_size_percents[:4]]), flush=True)
