# r6_pn_scale.py -- preserve_neighbors() at sizes between the MNIST stand-in and the benchmark shape: Gaussian mixtures of
# n points in R^64, k = 15, Log1p / LogRatio, Standardized, d = 2 (and 3); ms per embed() iteration and per evaluation, the
# kernel layout the library picked.  MDE_PANEL=0 in the environment keeps the CSR kernels (the state before the mid-regime rule).
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pymde_amd
from pymde_amd.average_distortion import fused_evaluate
dev = torch.device("cuda", 0)
for n in [int(a) for a in sys.argv[1:]] or [100_000, 200_000, 400_000]:
    for d in (2, 3):
        g = torch.Generator(device=dev); g.manual_seed(0)
        nf = 64
        centers = 4.0 * torch.randn((20, nf), device=dev, generator=g)
        lab = torch.randint(0, 20, (n,), device=dev, generator=g)
        if os.environ.get("PN_SORTED"):
            lab = torch.sort(lab).values          # items sorted by class: neighbours are near in the vertex order
        data = centers[lab] + torch.randn((n, nf), device=dev, generator=g)
        t0 = time.perf_counter()
        mde = pymde_amd.preserve_neighbors(data, embedding_dim=d, n_neighbors=15, attractive_penalty=pymde_amd.penalties.Log1p,
                                           repulsive_penalty=pymde_amd.penalties.LogRatio, constraint=pymde_amd.Standardized(), device=dev)
        torch.cuda.synchronize(); build = time.perf_counter() - t0
        mde.embed(max_iter=5); torch.cuda.synchronize()
        t0 = time.perf_counter(); mde.embed(max_iter=100, eps=0.0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        its = max(int(mde.solve_stats.iterations), 1)
        b = mde._binding(); X = mde.X.contiguous(); buf = torch.zeros(n * d + 1, device=dev)
        for _ in range(20): fused_evaluate(b, X, buf[:n * d].view(n, d), buf[n * d:])
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100): fused_evaluate(b, X, buf[:n * d].view(n, d), buf[n * d:])
        e.record(); torch.cuda.synchronize()
        p = int(mde.edges.shape[0])
        print("n=%7d d=%d  edges %8d  build %.2f s  %.4f ms per embed() iteration (%d its)  %.4f ms per evaluation  layout %s  stream %s  value %.5f" % (
            n, d, p, build, 1e3 * dt / its, its, a.elapsed_time(e) / 100, "ring" if int(b.struct(d).layout) == 1 else "CSR", b.stream_kind, float(mde.value)), flush=True)
        del mde, data
