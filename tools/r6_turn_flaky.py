# r6_turn_flaky.py -- the turn engine against the call-by-call loop, many times over: which solves differ, from which
# iteration, by how much (tests/test_gpu_solver.py::test_turn_calls_follow_the_python_loop_bit_for_bit failed 2 of ~13 runs)
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pymde_amd
DEV = "cuda"
n, p = 6000, 60000
rng = np.random.default_rng(17)
i = rng.integers(0, n, p); j = (i + 1 + rng.integers(0, n - 1, p)) % n
edges = torch.tensor(np.stack([i, j], 1), device=DEV)
w = torch.tensor(np.where(rng.random(p) < 0.3, -1.0, rng.uniform(0.5, 2.0, p)).astype(np.float32), device=DEV)
X0 = torch.tensor(rng.standard_normal((n, 2)).astype(np.float32), device=DEV)
def solve(cname, no_turn, **kw):
    if no_turn: os.environ["MDE_NO_TURN"] = "1"
    else: os.environ.pop("MDE_NO_TURN", None)
    c = pymde_amd.Centered() if cname == "centered" else pymde_amd.Standardized()
    mde = pymde_amd.MDE(n, 2, edges, pymde_amd.penalties.PushAndPull(w), constraint=c)
    X = mde.embed(X=c.project_onto_constraint(X0.clone()), **kw)
    st = mde.solve_stats
    return X.clone(), list(st.average_distortions), list(st.residual_norms), list(st.step_size_percents)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
for cname in ("centered", "standardized"):
    ref = {}
    for kwname, kw in (("60", dict(max_iter=60, eps=0.0)), ("1", dict(max_iter=1)), ("2", dict(max_iter=2)), ("3", dict(max_iter=3)), ("snap", dict(max_iter=12, snapshot_every=5))):
        base = solve(cname, True, **kw)
        for r in range(reps):
            for no_turn in (False, True):
                got = solve(cname, no_turn, **kw)
                if not (torch.equal(got[0], base[0]) and got[1] == base[1] and got[2] == base[2] and got[3] == base[3]):
                    bad += 1
                    k = next((t for t in range(min(len(got[1]), len(base[1]))) if got[1][t] != base[1][t]), None)
                    k2 = next((t for t in range(min(len(got[3]), len(base[3]))) if got[3][t] != base[3][t]), None)
                    print("DIFF %s kw=%s rep=%d no_turn=%s: first differing distortion at iteration %s (%r vs %r), step size at %s, lens %d/%d, X equal %s" % (
                        cname, kwname, r, no_turn, k, got[1][k] if k is not None else None, base[1][k] if k is not None else None, k2, len(got[1]), len(base[1]), torch.equal(got[0], base[0])), flush=True)
print("done, %d differing solves" % bad)
