"""Standardized projections at d = 32..128 against float64 torch, and their timings at config-5 size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pymde_amd

dev = torch.device("cuda", 0)
c = pymde_amd.Standardized()
torch.manual_seed(0)


import sys as _s
shapes = [] if "--time-only" in _s.argv else [(333, 32), (777, 64), (1000, 96), (5000, 128), (70001, 128), (500000, 128)]
for (n, d) in shapes:
    X = c.initialization(n, d, device=dev).contiguous()
    Z = torch.randn((n, d), device=dev)
    Xd, Zd = X.double(), Z.double()
    want = Zd - (Xd @ (Zd.T @ Xd)) / n           # [ref: constraints.py:186-192]
    got = c.project_onto_tangent_space(X, Z.clone(), inplace=True).double()
    err_t = float((got - want).abs().max() / want.abs().max())
    Y = (torch.randn((n, d), device=dev) * 2.0 + 0.3).contiguous()
    Yd = Y.double()
    Yc = Yd - Yd.mean(0)
    U, S, Vh = torch.linalg.svd(Yc, full_matrices=False)
    wantR = (n ** 0.5) * U @ Vh
    gotR = c.project_onto_constraint(Y.clone(), inplace=True).double()
    err_r = float((gotR - wantR).abs().max() / wantR.abs().max())
    cov = gotR.T @ gotR / n
    err_c = float((cov - torch.eye(d, device=dev, dtype=torch.float64)).abs().max())
    print("n=%d d=%d: tangent rel err %.2e | retraction rel err %.2e, |Z^T Z / n - I|_max %.2e, |col mean|_max %.2e"
          % (n, d, err_t, err_r, err_c, float(gotR.mean(0).abs().max())))


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]


n, d = 500000, 128
X = c.initialization(n, d, device=dev).contiguous()
Z = torch.randn((n, d), device=dev)
Y = X.clone()
print("config-5 shape (n=500k, d=128): tangent %.3f ms, retraction %.3f ms"
      % (timeit(lambda: c.project_onto_tangent_space(X, Z, inplace=True)), timeit(lambda: c.project_onto_constraint(Y, inplace=True))))
