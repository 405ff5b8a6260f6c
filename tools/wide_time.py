# wide_time.py -- the general-d CSR kernels at widths that are not a multiple of 128 floats (ms per evaluation)
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pymde_amd
from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate
rng = np.random.default_rng(0)
n, p = 200000, 4000000
i = rng.integers(0, n, p); j = (i + 1 + rng.integers(0, n - 1, p)) % n
edges = torch.tensor(np.stack([i, j], 1), device='cuda')
w = torch.tensor(rng.uniform(0.5, 2.0, p).astype(np.float32), device='cuda')
for d in (7, 10, 24, 50, 100):
    X = torch.randn(n, d, device='cuda')
    b = Binding(EdgePlan(n, edges), pymde_amd.penalties.Log1p(w))
    buf = torch.zeros(n * d + 1, device='cuda')
    for _ in range(3): fused_evaluate(b, X, buf[:n * d].view(n, d), buf[n * d:])
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fused_evaluate(b, X, buf[:n * d].view(n, d), buf[n * d:])
    e.record(); torch.cuda.synchronize()
    print("d=%3d: %.3f ms per evaluation" % (d, a.elapsed_time(e) / 20))
