"""round 6 debug: the row-sharded problem's constraint maps against the single-GPU kernels, element by element"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pymde_amd
from pymde_amd import distributed, optim
dev = torch.device("cuda", 0)
for d in (2, 64, 128):
    rng = np.random.default_rng(17 + d)
    n, p = 6000, 50000
    i = rng.integers(0, n, p); j = (i + 1 + rng.integers(0, n - 1, p)) % n
    e = np.unique(np.sort(np.stack([i, j], 1), 1), axis=0)
    w = rng.choice(np.array([-1.0, 1.0, 2.0], dtype=np.float32), size=len(e), p=[0.3, 0.4, 0.3])
    edges = torch.tensor(e, device=dev)
    pen = pymde_amd.penalties
    c = pymde_amd.Standardized()
    torch.manual_seed(0)
    x0 = c.initialization(n, d, device=dev)
    f = pen.PushAndPull(torch.tensor(w, device=dev), pen.Log1p, pen.Log)
    single = pymde_amd.MDE(n, d, edges, f, constraint=c)
    sh = distributed.ShardedMDE(n, d, edges, f, constraint=pymde_amd.Standardized(), rank=0, world_size=1, force_exchange=True)
    args = optim._sharded_solver_args(sh.average_distortion, sh.constraint)
    eng = optim._ShardedEngine(x0, 10, *args[1:])
    prob = optim._ShardedProblem(eng, args[0], sh.constraint)
    # projected gradient
    xs = x0.clone().requires_grad_(True)
    E = single.average_distortion(xs); E.backward()
    gs = c.project_onto_tangent_space(x0, xs.grad.clone(), inplace=True)
    prob.value_and_grad(eng.X)
    torch.cuda.synchronize()
    print("d", d, "tangent: max |diff| / max|g|", float((eng.g - gs).abs().max() / gs.abs().max()), "raw grad diff", float((xs.grad - xs.grad).abs().max()))
    # trial point
    direction = -gs
    eng.dir.copy_(direction)
    for t in (1.0, 30.0):
        want = c.project_onto_constraint(x0 + t * direction, inplace=False)
        prob.retract_step(t, eng.X_trial)
        torch.cuda.synchronize()
        Z = eng.X_trial.double()
        print("   t", t, "retract: max |diff|", float((eng.X_trial - want).abs().max()), " |Z^T Z/n - I|", float((Z.T @ Z / n - torch.eye(d, device=dev, dtype=torch.float64)).abs().max()),
              "single's", float((want.double().T @ want.double() / n - torch.eye(d, device=dev, dtype=torch.float64)).abs().max()), "mean", float(Z.mean(0).abs().max()))
    eng.close()
