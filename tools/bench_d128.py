"""Time the general-d kernel at BASELINE config 5 scale (n=500k, |E|=20M, d=128)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import pymde_amd
from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate
dev = torch.device("cuda", 0)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n, deg = 500_000, 40
gen = torch.Generator(device=dev); gen.manual_seed(0)
src = torch.arange(n, device=dev).repeat_interleave(deg)
dst = torch.randint(0, n - 1, (n * deg,), device=dev, generator=gen); dst += (dst >= src).long()
edges = torch.stack([torch.minimum(src, dst), torch.maximum(src, dst)], 1).contiguous()
w = 1.0 + (torch.rand(n * deg, device=dev, generator=gen) < 0.3).float()
X = torch.randn((n, d), device=dev)
for name, f in (("Quadratic", pymde_amd.penalties.Quadratic(w)), ("Log1p", pymde_amd.penalties.Log1p(w))):
    b = Binding(EdgePlan(n, edges), f)
    buf = torch.zeros(n * d + 1, device=dev)
    for _ in range(3):
        fused_evaluate(b, X, buf[:n * d].view(n, d), buf[n * d:])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fused_evaluate(b, X, buf[:n * d].view(n, d), buf[n * d:])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    p = edges.shape[0]
    gather_bytes = 2 * p * d * 4
    print("d=%d %s: %.3f ms  %.3e edges/s  gather %.2f TB/s (2p rows of %d B)  alg-roofline frac %.3f"
          % (d, name, dt * 1e3, p / dt, gather_bytes / dt / 1e12, d * 4, (12 * p + 2 * n * d * 4) / dt / 8e12))
