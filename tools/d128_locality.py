# d128_locality.py -- config-5 shape (n = 500k, |E| = 20M, d = 128) on graphs WITH locality: neighbours within a window
# of the vertex order (a k-NN graph whose vertices are sorted along a space-filling curve looks like this), against the
# uniform-random graph of the bench.  The kernel is the same (k_fused_wide4): what changes is where the 512-byte rows
# come from -- L2 / Infinity Cache instead of HBM.
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pymde_amd
from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate
dev = torch.device("cuda", 0)
d = int(os.environ.get("LOC_D", "128"))
n, deg = 500_000 * 128 // d if d > 128 else 500_000, 40
n = min(n, 500_000)
g = torch.Generator(device=dev); g.manual_seed(0)
X = torch.randn(n, d, device=dev, generator=g)
src = torch.arange(n, device=dev).repeat_interleave(deg)
print("d =", d, "n =", n)
for window in (0, 100_000, 10_000, 1_000, 100):
    if window == 0:
        dst = torch.randint(0, n - 1, (n * deg,), device=dev, generator=g); dst += (dst >= src).long()
    else:
        off = torch.randint(1, window + 1, (n * deg,), device=dev, generator=g)
        dst = (src + off) % n
    edges = torch.stack([torch.minimum(src, dst), torch.maximum(src, dst)], 1).contiguous()
    edges = torch.unique(edges, dim=0)
    if os.environ.get("LOC_SHUFFLE"):
        # the same graph under a random renumbering of its vertices: the locality is still there, the numbering hides it
        perm = torch.randperm(n, device=dev, generator=g)
        edges = perm[edges]
        edges = torch.stack([edges.min(1).values, edges.max(1).values], 1).contiguous()
    p = edges.shape[0]
    w = 1.0 + (torch.rand(p, device=dev, generator=g) < 0.3).float()
    plan = EdgePlan(n, edges)
    b = Binding(plan, pymde_amd.penalties.Log1p(w))
    buf = torch.zeros(n * d + 1, device=dev)
    for _ in range(3): fused_evaluate(b, X, buf[:n * d].view(n, d), buf[n * d:])
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fused_evaluate(b, X, buf[:n * d].view(n, d), buf[n * d:])
    e.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(e) / 10
    import time
    t0 = time.time(); info = plan.row_order(1); t1 = time.time()   # (already built by the first evaluation: a query)
    print("window %7s: %8d edges  %.3f ms per evaluation  %.2f ns per edge  row gathers %.2f TB/s   order: %s" % (
        window or "uniform", p, ms, 1e6 * ms / p, 2.0 * p * d * 4 / (ms * 1e-3) / 1e12,
        "in use (mean distance %.0f -> %.0f, %d levels)" % (info["mean_distance_before"], info["mean_distance_after"], info["levels"])
        if info["in_use"] else "none (mean distance %.0f, built: %.0f, %d levels)" % (info["mean_distance_before"], info["mean_distance_after"], info["levels"])))
    del b, edges, plan
