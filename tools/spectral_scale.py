"""Time quadratic.spectral at the BASELINE config-2 scale (n = 70k, k = 15 kNN-like graph)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pymde_amd
from pymde_amd import quadratic
rng = np.random.default_rng(0)
n, k = 70000, 15
src = np.repeat(np.arange(n), k); dst = (src + rng.integers(1, 200, n * k)) % n
key = np.unique(np.minimum(src, dst).astype(np.int64) * n + np.maximum(src, dst))
edges = torch.tensor(np.stack([key // n, key % n], 1), device="cuda")
w = torch.ones(edges.shape[0], device="cuda")
for cg, mi in ((True, 40), (True, 400)):
    torch.manual_seed(0); torch.cuda.synchronize(); t0 = time.time()
    X = quadratic.spectral(n, 2, edges, w, cg=cg, max_iter=mi)
    torch.cuda.synchronize(); dt = time.time() - t0
    mde = pymde_amd.MDE(n, 2, edges, pymde_amd.penalties.Quadratic(w), constraint=pymde_amd.Standardized())
    print("spectral n=%d p=%d cg=%s max_iter=%d: %.3f s, E=%.6g" % (n, edges.shape[0], cg, mi, dt, float(mde.average_distortion(X))))
