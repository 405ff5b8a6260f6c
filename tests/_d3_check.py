# _d3_check.py (test infrastructure, not collected by pytest) -- diagnosis of the d = 3 full-size parity failure: n = 250k, out-degree 200 (50M edges), d = 3;
# function and environment from the command line; prints mismatches against the oracle and whether two
# evaluations agree bitwise
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pymde_amd
from oracle import oracle
from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate
which = sys.argv[1]
d = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
edges, w, X = bench.make_workload(dev, n=250000, deg=200, d=d)
n, p = X.shape[0], edges.shape[0]
pen = pymde_amd.penalties
if which == "pushpull":
    w = w.clone(); w[(2 * p) // 3:] = -1.0
    f = pen.PushAndPull(w, pen.Log1p, pen.Log)
    of = oracle.func("LOG1P", w.cpu().numpy(), None, (1.5,), "LOG", (1.0,))
else:
    f = pen.Log1p(w)
    of = oracle.func("LOG1P", w.cpu().numpy(), None, (1.5,))
b = Binding(EdgePlan(n, edges), f)
bufs = []
for _ in range(2):
    buf = torch.zeros(n * d + 1, device=dev)
    fused_evaluate(b, X, buf[:n * d].view(n, d), buf[n * d:])
    bufs.append(buf)
wE, wgrad = oracle.average_distortion(edges.cpu().numpy(), X.cpu().numpy(), of)
got = bufs[0][:n * d].view(n, d).double().cpu().numpy()
scale = np.abs(wgrad).max()
bad = np.abs(got - wgrad) > 1e-4 * np.abs(wgrad) + 1e-5 * scale
rows = np.nonzero(bad.any(1))[0]
print("%s d=%d env=%s layout=%d codebook=%s: loss %.8g vs %.8g, %d bad elements in %d rows, max |diff| %.3e (scale %.3e), bitwise repeat %s"
      % (which, d, {k: v for k, v in os.environ.items() if k.startswith("MDE_")}, b.struct(d).layout, b.codebook,
         float(bufs[0][n * d]), wE, int(bad.sum()), len(rows), float(np.abs(got - wgrad).max()), scale,
         bool(torch.equal(bufs[0], bufs[1]))))
if len(rows):
    deg = np.bincount(edges.cpu().numpy().reshape(-1), minlength=n)
    print("  rows", rows[:12], "degrees", deg[rows[:12]], "diff", (got - wgrad)[rows[:4]])
