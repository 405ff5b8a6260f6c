"""Where does preserve_neighbors() spend its time at MNIST scale (n = 70k, 784 features)?"""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pymde_amd
from pymde_amd import preprocess, quadratic

dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
n, nf = 70000, 784
centers = rng.standard_normal((10, nf)).astype(np.float32) * 3
data = torch.tensor(centers[rng.integers(0, 10, n)] + rng.standard_normal((n, nf)).astype(np.float32), device=dev)
def tm(f, *a, **k):
    torch.cuda.synchronize(); t0 = time.time(); r = f(*a, **k); torch.cuda.synchronize(); return r, time.time() - t0
(e, w), t = tm(preprocess.k_nearest_neighbors, data, 15); print("kNN %.3f s" % t)
(e, w), t = tm(preprocess.k_nearest_neighbors, data, 15); print("kNN %.3f s (2nd)" % t)
import pymde_amd.quadratic as q
orig = q._lobpcg
def counted(*a, **kw):
    t0 = time.time(); out = orig(*a, **kw); torch.cuda.synchronize()
    print("  lobpcg: %.3f s" % (time.time() - t0)); return out
q._lobpcg = counted
X, t = tm(quadratic.spectral, n, 2, e, w, cg=True, max_iter=1000, device=dev); print("spectral %.3f s" % t)
pr = cProfile.Profile(); pr.enable()
X, t = tm(quadratic.spectral, n, 2, e, w, cg=True, max_iter=1000, device=dev); pr.disable(); print("spectral %.3f s (2nd)" % t)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
neg, t = tm(preprocess.sample_edges, n, e.shape[0], exclude=e, seed=0, device=dev); print("negative sampling %.3f s" % t)
mde, t = tm(pymde_amd.preserve_neighbors, data); print("preserve_neighbors total %.3f s" % t)
