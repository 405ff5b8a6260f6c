"""round 6 debug: two gloo ranks on one GPU, the row-sharded solver against the single process on the n = 20k problem"""
import os, sys, importlib.util
import numpy as np, torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pymde_amd
from pymde_amd import distributed
dist.init_process_group(backend="gloo")
rank = dist.get_rank()
here = os.path.join(ROOT, "tests")
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(here, "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
g = np.load(os.path.join(here, "golden", "trajectories_mid.npz"))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
n, e_mid, par = mg.mid_problem_arrays("neighbors")
pen = pymde_amd.penalties
def mk(): return pen.PushAndPull(torch.tensor(par, device=dev), pen.Log1p, pen.Log)
X0 = torch.tensor(g["neighbors__X0"], device=dev)
from pymde_amd import lbfgs as _host
_orig = _host.strong_wolfe
def _traced(phi, t, f, gtd, dmax, *a, **k):
    def phi2(tt):
        r = phi(tt)
        if rank == 0:
            print("      phi(%.9g) = f %.9g gtd %.9g ok %s" % (tt, r[0], r[1], r[2]))
        return r
    out = _orig(phi2, t, f, gtd, dmax, *a, **k)
    if rank == 0:
        print("   strong_wolfe(t=%.9g f=%.9g gtd=%.9g dmax=%.9g) -> f %.9g t %.9g" % (t, f, gtd, dmax, out[0], out[1]))
    return out
_host.strong_wolfe = _traced
os.environ["MDE_NO_TURN"] = "1"
single = pymde_amd.MDE(n, 2, torch.tensor(e_mid, device=dev), mk(), constraint=pymde_amd.Standardized(), device=dev)
if rank == 0:
    print("SINGLE")
    single.embed(X=X0.clone(), max_iter=3, eps=1e-12, memory_size=10)
dist.barrier()
if rank == 0:
    print("SHARDED")
for xg in ("1",):
    os.environ["MDE_SHARD_XGATHER"] = xg
    sh = distributed.ShardedMDE(n, 2, torch.tensor(e_mid, device=dev), mk(), constraint=pymde_amd.Standardized(), device=dev)
    sh.embed(X=X0.clone(), max_iter=3, eps=1e-12, memory_size=10)
    if rank == 0:
        print("xgather", xg, np.array(sh.solve_stats.average_distortions), sh.solve_stats.evaluations)
        print("   resid", np.array(sh.solve_stats.residual_norms)[:4], "steps", np.array(sh.solve_stats.step_size_percents)[:4])
if rank == 0:
    print("ref    ", g["neighbors__distortions"][0])
    print("   resid", g["neighbors__residuals"][0][:4], "steps", g["neighbors__steps"][0][:4])
dist.destroy_process_group()
