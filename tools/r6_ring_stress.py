# r6_ring_stress.py -- the ring layout FORCED (MDE_PANEL=1) on graphs auto mode would not give it, against the CSR kernels
# (MDE_PANEL=0) on the same tensors: loss and gradient to rounding, no fault.  n, degree, d, graph per case.
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pymde_amd
from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate
dev = torch.device("cuda", 0)
cases = [(2_000_000, 50, 2, "clusters"), (1_000_000, 50, 3, "clusters"), (1_000_000, 100, 2, "clusters"), (500_000, 20, 2, "clusters"),
         (3_000_000, 30, 2, "powerlaw"), (1_000_000, 50, 4, "clusters"), (1_000_000, 50, 1, "clusters")]
for n, deg, d, graph in cases:
    edges, w, X = bench.make_workload(dev, n=n, deg=deg, d=d, graph=graph)
    f = pymde_amd.penalties.PushAndPull(w, pymde_amd.penalties.Log1p, pymde_amd.penalties.Log) if graph == "clusters" else pymde_amd.penalties.Log1p(w)
    outs = {}
    for mode in ("1", "0"):
        os.environ["MDE_PANEL"] = mode
        plan = EdgePlan(n, edges)
        b = Binding(plan, f)
        buf = torch.zeros(n * d + 1, device=dev)
        fused_evaluate(b, X, buf[:n * d].view(n, d), buf[n * d:])
        torch.cuda.synchronize()
        outs[mode] = (buf, plan.ring_info()["built"])
        del plan, b
    a, c = outs["1"][0], outs["0"][0]
    gerr = float((a[:n * d] - c[:n * d]).abs().max() / c[:n * d].abs().max())
    lerr = abs(float(a[n * d]) - float(c[n * d])) / abs(float(c[n * d]))
    print("n=%d deg=%d d=%d %-9s ring built %s  grad rel diff %.2e  loss rel diff %.2e" % (n, deg, d, graph, outs["1"][1], gerr, lerr), flush=True)
    assert gerr < 2e-4 and lerr < 1e-5
print("ok")
