# r6_embed_dims.py -- end-to-end embed() across embedding dimensions (Standardized and Centered): a planted-cluster k-NN-like
# problem at n = 100k; ms per iteration, the constraint's residual at the solution, the descent of the objective
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pymde_amd
dev = torch.device("cuda", 0)
n, deg = int(os.environ.get("EMB_N", "100000")), 15
g = torch.Generator(device=dev); g.manual_seed(0)
src = torch.arange(n, device=dev).repeat_interleave(deg)
dst = (src + torch.randint(1, 200, (n * deg,), device=dev, generator=g)) % n          # attractive: near neighbours
neg = torch.randint(0, n, (n * deg,), device=dev, generator=g)                          # repulsive: random pairs
e = torch.cat([torch.stack([src, dst], 1), torch.stack([src, neg], 1)])
e = e[e[:, 0] != e[:, 1]]
e = torch.unique(torch.stack([e.min(1).values, e.max(1).values], 1), dim=0)
w = torch.ones(e.shape[0], device=dev); w[torch.randperm(e.shape[0], device=dev, generator=g)[: e.shape[0] // 2]] = -1.0
for d in [int(a) for a in sys.argv[1:]] or [2, 3, 5, 8, 10, 16, 32, 50, 64, 100, 128]:
    for cname, c in (("Standardized", pymde_amd.Standardized()), ("Centered", pymde_amd.Centered())):
        f = pymde_amd.penalties.PushAndPull(w, pymde_amd.penalties.Log1p, pymde_amd.penalties.Log)
        mde = pymde_amd.MDE(n, d, e, f, constraint=c, device=dev)
        torch.manual_seed(0)
        mde.embed(max_iter=5); torch.cuda.synchronize()
        v0 = float(mde.value)
        t0 = time.perf_counter(); mde.embed(max_iter=60, eps=0.0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        its = max(int(mde.solve_stats.iterations), 1)
        X = mde.X.double()
        res = float((X.T @ X / n - torch.eye(d, device=dev, dtype=torch.float64)).abs().max()) if cname == "Standardized" else float(X.mean(0).abs().max())
        print("d=%4d %-12s %.3f ms per iteration (%d its)  distortion %.5f -> %.5f  constraint residual %.1e" % (d, cname, 1e3 * dt / its, its, v0, float(mde.value), res), flush=True)
