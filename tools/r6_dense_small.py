# r6_dense_small.py -- dense problems on few items (preserve_distances on a graph with every pair retained): evaluation
# time, layout, and time per embed() iteration; n = 2000 .. 10000 items, all n (n - 1) / 2 pairs, Quadratic loss, d = 2
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pymde_amd
from pymde_amd.average_distortion import fused_evaluate
dev = torch.device("cuda", 0)
for n in [int(a) for a in sys.argv[1:]] or [2000, 5000, 10000]:
    iu = torch.triu_indices(n, n, 1, device=dev).T.contiguous()
    p = iu.shape[0]
    g = torch.Generator(device=dev); g.manual_seed(0)
    dev_ = torch.randint(1, 40, (p,), device=dev, generator=g).float()       # hop-count-like deviations
    mde = pymde_amd.MDE(n, 2, iu, pymde_amd.losses.Quadratic(dev_), constraint=pymde_amd.Centered(), device=dev)
    mde.embed(max_iter=5); torch.cuda.synchronize()
    t0 = time.perf_counter(); mde.embed(max_iter=50, eps=0.0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    its = max(int(mde.solve_stats.iterations), 1)
    b = mde._binding(); X = mde.X.contiguous(); buf = torch.zeros(n * 2 + 1, device=dev)
    for _ in range(10): fused_evaluate(b, X, buf[:n * 2].view(n, 2), buf[n * 2:])
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): fused_evaluate(b, X, buf[:n * 2].view(n, 2), buf[n * 2:])
    e.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(e) / 50
    print("n=%6d pairs=%9d  %.4f ms per evaluation (%.3f per 1e8 half-edges)  %.4f ms per embed() iteration (%d its, %d evaluations)  layout %s stream %s" % (
        n, p, ms, ms * 1e8 / (2 * p), 1e3 * dt / its, its, -1, "ring" if int(b.struct(2).layout) == 1 else "CSR", b.stream_kind), flush=True)
    del mde, iu
