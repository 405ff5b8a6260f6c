"""BASELINE config 2 stand-in (MNIST is not available offline): 70 000 points from a 10-component
Gaussian mixture in R^784; full preserve_neighbors pipeline on the GPU, stage by stage."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pymde_amd
from pymde_amd import preprocess, quadratic
n, nf = int(sys.argv[1]) if len(sys.argv) > 1 else 70000, 784
rng = np.random.default_rng(0)
centers = rng.standard_normal((10, nf)) * 2.0
labels = rng.integers(0, 10, n)
data = torch.tensor((centers[labels] + rng.standard_normal((n, nf))).astype(np.float32), device="cuda")
def t():
    torch.cuda.synchronize(); return time.time()
for rep in range(2):
    t0 = t(); e, w = preprocess.k_nearest_neighbors(data, 15); t1 = t()
    print("rep %d: k-NN graph (k=15, n=%d, %d features): %.3f s -> %d edges" % (rep, n, nf, t1 - t0, len(e)))
torch.manual_seed(0)
t0 = t()
mde = pymde_amd.preserve_neighbors(data, embedding_dim=2, constraint=pymde_amd.Standardized(), seed=0)
t1 = t()
X = mde.embed(max_iter=300, eps=1e-5)
t2 = t()
s = mde.solve_stats
print("preserve_neighbors(): %.3f s (kNN + spectral init + negative sampling + plan); embed(): %.3f s, %d iterations, "
      "distortion %.5f -> %.5f, residual %.2e; p = %d edges" % (t1 - t0, t2 - t1, s.iterations, s.average_distortions[0],
      s.average_distortions[-1], s.residual_norms[-1], int(mde.p)))
