import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pymde_amd
dev = torch.device("cuda", 0)
for n in (100000, 1000000):
    edges, w, X0 = bench.make_workload(dev, n=n, deg=20, d=2)
    w = w.clone(); w[::3] = -1.0
    for m in (1, 5, 10, 16, 30, 63, 100):
        f = pymde_amd.penalties.PushAndPull(w, pymde_amd.penalties.Log1p, pymde_amd.penalties.Log)
        mde = pymde_amd.MDE(n, 2, edges, f, constraint=pymde_amd.Centered(), device=dev)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mde.embed(max_iter=5, memory_size=m); torch.cuda.synchronize()
            t0 = time.perf_counter(); mde.embed(max_iter=80, eps=0.0, memory_size=m); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        its = max(int(mde.solve_stats.iterations), 1)
        print("n=%7d memory_size=%3d  %.3f ms per iteration  distortion %.5f" % (n, m, 1e3 * dt / its, float(mde.value)), flush=True)
