"""Degree skew of the config-2 stand-in graph and what it costs the CSR kernel: degree quantiles and the
time of one fused forward+backward (MDE_GROUP = lanes per row is read from the environment by the library)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pymde_amd

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0)
n, nf = 70_000, 784
centers = 4.0 * torch.randn((10, nf), device=dev, generator=g)
data = centers[torch.randint(0, 10, (n,), device=dev, generator=g)] + torch.randn((n, nf), device=dev, generator=g)
mde = pymde_amd.preserve_neighbors(data, embedding_dim=2, n_neighbors=15, attractive_penalty=pymde_amd.penalties.Log1p,
                                   repulsive_penalty=pymde_amd.penalties.LogRatio, constraint=pymde_amd.Standardized(), device=dev)
e = mde.edges
deg = torch.bincount(e.reshape(-1), minlength=n).cpu().numpy()
if os.environ.get("MDE_GROUP", "") in ("", "0"):
    print("half-edges per row: mean %.1f  median %d  p99 %d  p99.9 %d  max %d; rows > 256: %d, > 1024: %d" % (
        deg.mean(), np.median(deg), np.percentile(deg, 99), np.percentile(deg, 99.9), deg.max(),
        (deg > 256).sum(), (deg > 1024).sum()))
X = mde.constraint.initialization(n, 2, dev)
x = X.clone().requires_grad_(True)
for _ in range(5):
    E = mde.average_distortion(x); E.backward(); x.grad = None
torch.cuda.synchronize()
ts = []
for _ in range(50):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); E = mde.average_distortion(x); E.backward(); b.record(); b.synchronize()
    ts.append(a.elapsed_time(b)); x.grad = None
print("MDE_GROUP=%s: fwd+bwd through autograd median %.1f us" % (os.environ.get("MDE_GROUP", "auto"), 1e3 * np.median(ts)))
