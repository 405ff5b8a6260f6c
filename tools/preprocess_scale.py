"""Time the GPU edge preprocessing at the BASELINE config-4 scale against the reference's numpy path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pymde_amd import preprocess
from oracle import oracle
n, p = 1_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
rng = np.random.default_rng(0)
e = rng.integers(0, n, (p, 2)); e = e[e[:, 0] != e[:, 1]]
et = torch.tensor(e, device="cuda")
torch.cuda.synchronize(); t0 = time.time()
sim = preprocess.deduplicate_edges(et, n_items=n); torch.cuda.synchronize(); t1 = time.time()
neg = preprocess.dissimilar_edges(n, sim, seed=0); torch.cuda.synchronize(); t2 = time.time()
print("GPU: deduplicate %d -> %d edges in %.3f s; sample %d dissimilar edges in %.3f s" % (len(e), len(sim), t1 - t0, len(neg), t2 - t1))
sub = e[: p // 10]
t0 = time.time(); oracle.deduplicate_edges(sub); t1 = time.time()
print("CPU numpy (reference algorithm) deduplicate on a 10%% sample (%d rows): %.2f s" % (len(sub), t1 - t0))
