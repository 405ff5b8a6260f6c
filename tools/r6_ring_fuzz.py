# r6_ring_fuzz.py [cases] [seed] -- random problems through the ring layout FORCED (MDE_PANEL=1), the ring layout with the
# round-5 row map (MDE_RING_ASSIGN=1), and auto mode, against the CSR kernels (MDE_PANEL=0) on the same tensors
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pymde_amd
from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate
dev = torch.device("cuda", 0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
pen, los = pymde_amd.penalties, pymde_amd.losses
worst = 0.0
for c in range(cases):
    n = rnd.choice([2000, 5000, 12345, 40000, 100000, 250000, 600000])
    deg = rnd.choice([3, 8, 20, 50, 120]) if n <= 100000 else rnd.choice([5, 20, 50])
    d = rnd.choice([1, 2, 2, 3, 3, 4])
    graph = rnd.choice(["uniform", "powerlaw", "clusters", "hub"]) if n >= 40000 else rnd.choice(["uniform", "clusters"])
    edges, w, X = bench.make_workload(dev, n=n, deg=deg, d=d, graph=graph)
    p = edges.shape[0]
    kind = rnd.choice(["log1p", "pushpull", "quadratic", "huber", "contw"])
    if kind == "pushpull":
        w = w.clone(); w[torch.rand(p, device=dev) < 0.4] = -1.0
        f = pen.PushAndPull(w, pen.Log1p, pen.Log)
    elif kind == "quadratic":
        f = pen.Quadratic(w)
    elif kind == "huber":
        f = los.Huber(0.5 + w, 0.5)
    elif kind == "contw":
        f = pen.Log1p(w * (0.5 + torch.rand(p, device=dev)))      # continuous weights: the fp32 parameter stream
    else:
        f = pen.Log1p(w)
    outs = {}
    for mode, env in (("csr", {"MDE_PANEL": "0"}), ("ring", {"MDE_PANEL": "1"}), ("ring+assign", {"MDE_PANEL": "1", "MDE_RING_ASSIGN": "1"}), ("auto", {})):
        for k in ("MDE_PANEL", "MDE_RING_ASSIGN"): os.environ.pop(k, None)
        os.environ.update(env)
        plan = EdgePlan(n, edges)
        b = Binding(plan, f)
        buf = torch.zeros(n * d + 1, device=dev)
        fused_evaluate(b, X, buf[:n * d].view(n, d), buf[n * d:])
        buf2 = torch.zeros(n * d + 1, device=dev)
        fused_evaluate(b, X, buf2[:n * d].view(n, d), buf2[n * d:])
        torch.cuda.synchronize()
        assert torch.equal(buf, buf2), (c, mode, "not reproducible")
        outs[mode] = (buf, plan.ring_info()["built"])
        del plan, b
    ref = outs["csr"][0]
    msg = "case %2d n=%6d deg=%3d d=%d %-9s %-9s" % (c, n, deg, d, graph, kind)
    for mode in ("ring", "ring+assign", "auto"):
        a = outs[mode][0]
        gerr = float((a[:n * d] - ref[:n * d]).abs().max() / ref[:n * d].abs().max().clamp_min(1e-30))
        lerr = abs(float(a[n * d]) - float(ref[n * d])) / max(abs(float(ref[n * d])), 1e-30)
        worst = max(worst, gerr, lerr)
        msg += "  %s[%s] %.1e/%.1e" % (mode, "R" if outs[mode][1] else "C", gerr, lerr)
        assert gerr < 5e-4 and lerr < 2e-5, msg
    print(msg, flush=True)
print("ok, worst", worst)
