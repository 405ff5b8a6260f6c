# host_floor.py -- seconds per embed() iteration when the GPU work is negligible (tiny problem): what the
# Python + ctypes + launch + read-back loop costs by itself
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import pymde_amd
rng = np.random.default_rng(0)
for cname in ("Standardized", "Centered"):
    n, p = 3000, 30000
    i = rng.integers(0, n, p); j = (i + 1 + rng.integers(0, n - 1, p)) % n
    edges = torch.tensor(np.stack([i, j], 1), device='cuda')
    w = torch.tensor(rng.uniform(0.5, 2.0, p).astype(np.float32), device='cuda')
    mde = pymde_amd.MDE(n, 2, edges, pymde_amd.penalties.Log1p(w), constraint=getattr(pymde_amd, cname)())
    mde.embed(max_iter=5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mde.embed(max_iter=300, eps=0.0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(cname, "n=3000: %.1f us per iteration (%d iterations)" % (1e6 * dt / mde.solve_stats.iterations, mde.solve_stats.iterations))
