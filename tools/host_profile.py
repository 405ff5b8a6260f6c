"""cProfile of embed() on a mid-size problem: where does the host time per iteration go?"""
import cProfile, pstats, sys, os, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pymde_amd

dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
n, k = 70000, 15
i = np.repeat(np.arange(n), k)
j = (i + 1 + rng.integers(0, n - 1, n * k)) % n
e = np.unique(np.sort(np.stack([i, j], 1), 1), axis=0)
w = np.where(rng.random(len(e)) < 0.4, -1.0, rng.choice([1.0, 2.0], len(e))).astype(np.float32)
edges = torch.tensor(e, device=dev)
for cname, c in (("centered", pymde_amd.Centered()), ("standardized", pymde_amd.Standardized())):
    f = pymde_amd.penalties.PushAndPull(torch.tensor(w, device=dev))
    mde = pymde_amd.MDE(n, 2, edges, f, constraint=c)
    mde.embed(max_iter=5)
    torch.cuda.synchronize()
    t0 = time.time()
    mde.embed(max_iter=200, eps=1e-12)
    torch.cuda.synchronize()
    dt = time.time() - t0
    it = mde.solve_stats.iterations
    print("%s: %d iterations, %.3f ms/iter" % (cname, it, 1e3 * dt / it))
pr = cProfile.Profile()
pr.enable()
mde.embed(max_iter=200, eps=1e-12)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
