import os, sys, importlib.util
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pymde_amd
from pymde_amd import distributed
here = os.path.join(ROOT, "tests")
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(here, "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
g = np.load(os.path.join(here, "golden", "trajectories_mid.npz"))
dev = torch.device("cuda", 0)
n, e_mid, par = mg.mid_problem_arrays("neighbors")
pen = pymde_amd.penalties
def mk(): return pen.PushAndPull(torch.tensor(par, device=dev), pen.Log1p, pen.Log)
X0 = torch.tensor(g["neighbors__X0"], device=dev)
single = pymde_amd.MDE(n, 2, torch.tensor(e_mid, device=dev), mk(), constraint=pymde_amd.Standardized(), device=dev)
single.embed(X=X0.clone(), max_iter=8, eps=1e-12, memory_size=10)
print("single ", np.array(single.solve_stats.average_distortions), single.solve_stats.evaluations)
os.environ["MDE_NO_TURN"] = "1"
single.embed(X=X0.clone(), max_iter=8, eps=1e-12, memory_size=10)
print("noturn ", np.array(single.solve_stats.average_distortions), single.solve_stats.evaluations)
os.environ.pop("MDE_NO_TURN")
for W in (1, 2):
    sh = distributed.ShardedMDE(n, 2, torch.tensor(e_mid, device=dev), mk(), constraint=pymde_amd.Standardized(), device=dev, rank=0, world_size=W, force_exchange=True)
    sh.embed(X=X0.clone(), max_iter=8, eps=1e-12, memory_size=10)
    print("shard W=%d" % W, np.array(sh.solve_stats.average_distortions), sh.solve_stats.evaluations)
    print("   resid", np.array(sh.solve_stats.residual_norms)[:4], "steps", np.array(sh.solve_stats.step_size_percents)[:4])
print("ref    ", g["neighbors__distortions"][0])
print("   resid", g["neighbors__residuals"][0][:4], "steps", g["neighbors__steps"][0][:4])
print("single resid", np.array(single.solve_stats.residual_norms)[:4], "steps", np.array(single.solve_stats.step_size_percents)[:4])
