"""Time a full embed() at the BASELINE config-4 scale (secondary metric: s/iteration)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pymde_amd

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
edges, w, X = bench.make_workload(dev, n=n)
for cname, c in (("centered", pymde_amd.Centered()), ("standardized", pymde_amd.Standardized())):
    t0 = time.time()
    mde = pymde_amd.MDE(n, 2, edges, pymde_amd.penalties.Log1p(w), constraint=c)
    torch.cuda.synchronize()
    t1 = time.time()
    X0 = c.project_onto_constraint(X.clone())
    mde.embed(X=X0, max_iter=3)          # warm-up (plan layout, parameter permutation)
    torch.cuda.synchronize()
    t2 = time.time()
    mde.embed(X=X0, max_iter=iters, eps=1e-12)
    torch.cuda.synchronize()
    t3 = time.time()
    s = mde.solve_stats
    print("%s: MDE() %.2fs warmup %.2fs | %d iterations in %.3fs = %.2f ms/iter | E %.5f -> %.5f"
          % (cname, t1 - t0, t2 - t1, s.iterations, t3 - t2, 1e3 * (t3 - t2) / s.iterations,
             s.average_distortions[0], s.average_distortions[-1]))
