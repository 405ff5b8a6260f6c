# r6_widep_check.py -- the pipelined general-d kernel (k_fused_wide4p) against a float64 torch evaluation of the same
# objective on ragged graphs (empty rows, rows of 1..3 half-edges, a hub, n not a multiple of anything), d = 128 / 256 / 512,
# and three runs bitwise equal.  MDE_WIDE_P=0 in the environment checks the old kernel the same way.
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pymde_amd
from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
worst = 0.0
for d in (5, 6, 7, 9, 10, 11, 13, 17, 31, 33, 50, 99, 101, 127, 129, 255, 257, 511, 8, 12, 16, 64, 128, 512):
    for n, p in ((1, 0), (37, 50), (1000, 3000), (20011, 400000)):
        if n < 2: continue
        X = torch.randn(n, d, device=dev, generator=g)
        i = torch.randint(0, n, (p,), device=dev, generator=g)
        j = torch.randint(0, n, (p,), device=dev, generator=g)
        # a hub and a block of isolated vertices
        i[: p // 10] = 3
        keep = (i != j) & (i % 7 != 5) & (j % 7 != 5)
        e = torch.stack([torch.minimum(i, j), torch.maximum(i, j)], 1)[keep]
        e = torch.unique(e, dim=0).contiguous()
        pp = e.shape[0]
        w = 0.5 + torch.rand(pp, device=dev, generator=g)
        plan = EdgePlan(n, e)
        if os.environ.get("CHECK_ORDER"):
            print("   order:", plan.row_order(2) if n >= 8192 else "(small)")
        b = Binding(plan, pymde_amd.penalties.Log1p(w))
        buf = torch.zeros(n * d + 1, device=dev)
        outs = []
        for _ in range(3):
            buf.zero_()
            fused_evaluate(b, X, buf[:n * d].view(n, d), buf[n * d:])
            outs.append(buf.clone())
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "not bitwise reproducible"
        Xd = X.double().requires_grad_(True)
        dist = (Xd[e[:, 0]] - Xd[e[:, 1]]).pow(2).sum(1).sqrt()
        loss = (w.double() * torch.log1p(dist.pow(1.5))).mean()
        loss.backward()
        gref = Xd.grad
        gg = outs[0][:n * d].view(n, d).double()
        err = (gg - gref).abs().max().item() / max(gref.abs().max().item(), 1e-30)
        lerr = abs(outs[0][n * d].item() - loss.item()) / abs(loss.item())
        worst = max(worst, err, lerr)
        print("d=%d n=%d p=%d  grad rel err %.2e  loss rel err %.2e" % (d, n, pp, err, lerr))
        assert err < 2e-5 and lerr < 1e-5
print("ok, worst", worst)
