"""BASELINE configs[0] at full size: preserve_distances on a 5000-node cycle graph, Quadratic loss
(the reference: 13.4 s to build the problem, 1.44 s per iteration on 8 CPU threads)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pymde_amd
n = 5000
cyc = np.array([[i, (i + 1) % n] for i in range(n)])
def t():
    torch.cuda.synchronize(); return time.time()
t0 = t()
graph = pymde_amd.Graph.from_edges(torch.tensor(cyc), n_items=n)
mde = pymde_amd.preserve_distances(graph, embedding_dim=2, loss=pymde_amd.losses.Quadratic)
t1 = t()
torch.manual_seed(0)
mde.embed(max_iter=50)
t2 = t()
s = mde.solve_stats
print("config 1: p = %d pairs; problem built in %.2f s; embed(max_iter=50): %.3f s = %.2f ms/iteration; "
      "distortion %.4g -> %.4g" % (int(mde.p), t1 - t0, t2 - t1, 1e3 * (t2 - t1) / s.iterations,
                                   s.average_distortions[0], s.average_distortions[-1]))
X = mde.X.cpu().numpy(); r = np.linalg.norm(X - X.mean(0), axis=1)
print("radius std/mean = %.4f (a cycle embeds as a circle)" % (r.std() / r.mean()))
