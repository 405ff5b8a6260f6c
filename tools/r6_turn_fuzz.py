# r6_turn_fuzz.py -- mde_turn_* against the call-by-call loop (MDE_NO_TURN=1) across sizes, dimensions and constraints:
# iterates and recorded statistics bit for bit
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pymde_amd
dev = torch.device("cuda", 0)
bad = 0
for n, deg in ((6000, 10), (50000, 20), (150000, 20), (150000, 50)):
    for d in (2, 3, 8):
        for graph in ("uniform", "clusters"):
            edges, w, X0 = bench.make_workload(dev, n=n, deg=deg, d=d if d <= 4 else 2, graph=graph)
            if d > 4:
                X0 = torch.randn((n, d), device=dev, generator=torch.Generator(device=dev).manual_seed(0))
            w = w.clone(); w[torch.rand(w.shape[0], device=dev, generator=torch.Generator(device=dev).manual_seed(1)) < 0.3] = -1.0
            for cname in ("centered", "standardized"):
                runs = []
                for no_turn in (False, True):
                    if no_turn: os.environ["MDE_NO_TURN"] = "1"
                    else: os.environ.pop("MDE_NO_TURN", None)
                    c = pymde_amd.Centered() if cname == "centered" else pymde_amd.Standardized()
                    f = pymde_amd.penalties.PushAndPull(w, pymde_amd.penalties.Log1p, pymde_amd.penalties.Log)
                    mde = pymde_amd.MDE(n, d, edges, f, constraint=c, device=dev)
                    X = mde.embed(X=c.project_onto_constraint(X0.clone()), max_iter=40, eps=0.0)
                    st = mde.solve_stats
                    runs.append((X.clone(), list(st.average_distortions), list(st.residual_norms), list(st.step_size_percents)))
                a, b = runs
                same = torch.equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2] and a[3] == b[3]
                bad += 0 if same else 1
                print("n=%6d deg=%2d d=%d %-8s %-12s %s  (%.5f -> %.5f)" % (n, deg, d, graph, cname, "same" if same else "DIFFERENT", a[1][0], a[1][-1]), flush=True)
os.environ.pop("MDE_NO_TURN", None)
print("done,", bad, "different")
