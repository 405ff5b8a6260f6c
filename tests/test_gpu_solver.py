"""embed(): the device-resident projected L-BFGS against the reference's recorded
trajectories (tests/golden/trajectories.npz, cycle.npz) and solver-level properties.

Tolerance tiers: first iterations of the trajectory rtol 1e-3 (fp32
summation order makes the iterates drift apart afterwards -- the reference differs from
itself by O(1) in X between 1 and 8 threads); end of solve: final average distortion within
1e-2 relative; constraint residuals at the projection tolerance; no element-wise comparison
of the final embedding."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _problem(g, name):
    import pymde_amd
    pen, los = pymde_amd.penalties, pymde_amd.losses
    n = int(g["n"])
    edges = torch.tensor(g["edges"], device=DEV)
    w_pos = torch.tensor(g["w_pos"], device=DEV)
    w_mix = torch.tensor(g["w_mix"], device=DEV)
    dev = torch.tensor(g["dev"], device=DEV)
    d = int(g[name + "__d"])
    cname = str(g[name + "__constraint"])
    f = {
        "quad_std": lambda: pen.Quadratic(w_pos),
        "log1p_centered": lambda: pen.Log1p(w_pos),
        "pushpull_std": lambda: pen.PushAndPull(w_mix, pen.Log1p, pen.Log),
        "pushpull_centered_d3": lambda: pen.PushAndPull(w_mix),
        "absolute_centered": lambda: los.Absolute(dev),
        "huber_std": lambda: los.Huber(dev, 0.5),
        "quadloss_anchored": lambda: los.Quadratic(dev),
    }[name]()
    if cname == "standardized":
        c = pymde_amd.Standardized()
    elif cname == "centered":
        c = pymde_amd.Centered()
    else:
        c = pymde_amd.Anchored(torch.tensor(g["anchors"]), torch.tensor(g["anchor_values"]))
    return pymde_amd.MDE(n, d, edges, f, constraint=c), c, cname


NAMES = ["quad_std", "log1p_centered", "pushpull_std", "pushpull_centered_d3", "absolute_centered",
         "huber_std", "quadloss_anchored"]


def _match_member(got, members, k, rtol):
    """Index of an ensemble member whose first k entries agree with `got`, or None."""
    got = np.asarray(got[:k], dtype=np.float64)
    for idx in range(members.shape[0]):
        want = members[idx, :k]
        if np.all(np.isfinite(want)) and np.allclose(got, want, rtol=rtol, atol=1e-7):
            return idx
    return None


@pytest.mark.parametrize("name", NAMES)
def test_trajectory_matches_reference(golden_trajectories, name):
    """The fixture holds an ENSEMBLE of reference runs from X0 perturbed by <= 1e-5 (relative):
    the reference's line search branches on fp32 rounding (see make_golden.py), so parity is
    "the first iterations agree with one member to rtol 1e-3 and the run stays inside the
    ensemble's envelope"."""
    g = golden_trajectories
    mde, c, cname = _problem(g, name)
    X0 = torch.tensor(g[name + "__X0"], device=DEV)
    mde.embed(X=X0, max_iter=12, eps=1e-9, memory_size=5)
    s = mde.solve_stats
    E_ref, R_ref, S_ref = g[name + "__distortions"], g[name + "__residuals"], g[name + "__steps"]
    # iteration 0 is the plain evaluation at X0: tight
    assert s.average_distortions[0] == pytest.approx(E_ref[0, 0], rel=1e-5)
    assert s.residual_norms[0] == pytest.approx(R_ref[0, 0], rel=1e-4)
    # the first 5 iterations within rtol 1e-3 of the oracle run.  "The" oracle run does
    # not exist -- six of the seven ensembles (12 reference runs from X0 perturbed by <= 1e-5)
    # spread by 1e-2 .. 4e-1 already at iteration 1 -- so the tier is stated against a member:
    # one reference run is followed for the first 5 iterations (measured on MI355X: every case
    # follows its member for all 12 recorded iterations), and the run stays inside the envelope.
    spread = (np.nanmax(E_ref, 0) - np.nanmin(E_ref, 0)) / np.abs(np.nanmean(E_ref, 0))
    k = 5
    idx = _match_member(s.average_distortions, E_ref, k, 1e-3)
    assert idx is not None, (s.average_distortions[:k], E_ref[:, :k])
    # which members (X0 perturbed by NOISE[member % 4], make_golden.py) are followed: a run that only a
    # 1e-5-perturbed reference run reproduces, while the unperturbed / 1e-7 ones do not, is NOT parity
    noise = [0.0, 1e-7, 1e-6, 1e-5]
    matched = [i for i in range(E_ref.shape[0]) if _match_member(s.average_distortions, E_ref[i:i + 1], k, 1e-3) is not None]
    print("trajectory %s: matches reference members %s (X0 noise %s)" % (name, matched, [noise[i % 4] for i in matched]))
    assert any(noise[i % 4] <= 1e-6 for i in matched), (name, matched)
    follow = k
    while follow < min(E_ref.shape[1], len(s.average_distortions)) and \
            _match_member(s.average_distortions, E_ref[idx:idx + 1], follow + 1, 1e-3) is not None:
        follow += 1
    print("trajectory %s: member %d of %d followed for %d iterations (ensemble spread > 1e-3 from iteration %d)"
          % (name, idx, E_ref.shape[0], follow, int(np.argmax(spread > 1e-3)) if (spread > 1e-3).any() else -1))
    np.testing.assert_allclose(s.residual_norms[:k], R_ref[idx, :k], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(s.step_size_percents[:k], S_ref[idx, :k], rtol=5e-3, atol=1e-5)
    m = min(E_ref.shape[1], len(s.average_distortions))
    lo = np.nanmin(E_ref[:, :m], axis=0)
    hi = np.nanmax(E_ref[:, :m], axis=0)
    got = np.array(s.average_distortions[:m])
    assert np.all(got >= lo * (1 - 2e-2) - 1e-9) and np.all(got <= hi * (1 + 2e-2) + 1e-9), (got, lo, hi)
    assert s.iterations == len(s.average_distortions) == len(s.residual_norms)
    X = mde.X.double().cpu().numpy()
    n, d = X.shape
    if cname == "standardized":
        np.testing.assert_allclose(X.T @ X / n, np.eye(d), rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(X.mean(0), 0, atol=1e-5)
    elif cname == "centered":
        np.testing.assert_allclose(X.mean(0), 0, atol=1e-5)
    else:
        np.testing.assert_array_equal(X[g["anchors"]].astype(np.float32), g["anchor_values"])


@pytest.mark.parametrize("name", NAMES)
def test_end_of_solve_value(golden_trajectories, name):
    g = golden_trajectories
    mde, _, _ = _problem(g, name)
    X0 = torch.tensor(g[name + "__X0"], device=DEV)
    mde.embed(X=X0, max_iter=150, eps=1e-6, memory_size=10)
    finals = g[name + "__final_value_150"]
    # within 1e-2 (relative) of the reference ensemble's own spread of final values
    assert finals.min() * (1 - 1e-2) - 1e-9 <= mde.value <= finals.max() * (1 + 1e-2) + 1e-9, (mde.value, finals)
    # `value` is recorded at the start of the last step: X is one step past it (optim.py:149, 165)
    assert float(mde.average_distortion(mde.X)) <= mde.value * (1 + 1e-5) + 1e-12
    assert mde.residual_norm == mde.solve_stats.residual_norms[-1]
    # monotone decrease of the recorded objective (strong-Wolfe accepts only decreases)
    E = np.array(mde.solve_stats.average_distortions)
    assert (np.diff(E) <= 1e-6 * np.abs(E[:-1]) + 1e-12).all()


def test_config1_cycle_graph_quadratic_loss(golden_cycle):
    """BASELINE config 1 (scaled to n = 300): preserve_distances on a cycle graph, Quadratic
    loss over all pairs, against the reference CPU run from the same initial point."""
    import pymde_amd
    g = golden_cycle
    n = int(g["n"])
    f = pymde_amd.losses.Quadratic(torch.tensor(g["deviations"], device=DEV))
    mde = pymde_amd.MDE(n, 2, torch.tensor(g["edges"], device=DEV), f)
    mde.embed(X=torch.tensor(g["X0"], device=DEV), max_iter=40, eps=1e-8)
    E = np.array(mde.solve_stats.average_distortions)
    assert _match_member(E, g["distortions"], 3, 1e-3) is not None, (E[:3], g["distortions"][:, :3])
    finals = g["final_value"]
    assert finals.min() * (1 - 1e-2) <= mde.value <= finals.max() * (1 + 1e-2)
    # a cycle embeds as a circle: all radii equal
    X = mde.X.cpu().numpy()
    r = np.linalg.norm(X - X.mean(0), axis=1)
    assert r.std() / r.mean() < 0.05


def test_embed_api_and_stats():
    import pymde_amd
    torch.manual_seed(0)
    n = 500
    edges = pymde_amd.all_edges(40)
    rng = np.random.default_rng(0)
    i = rng.integers(0, n, 4000)
    j = (i + 1 + rng.integers(0, n - 1, 4000)) % n
    edges = torch.tensor(np.stack([i, j], 1))
    mde = pymde_amd.MDE(n, 2, edges, pymde_amd.penalties.Quadratic(torch.ones(4000)),
                        constraint=pymde_amd.Standardized())
    assert "standardized" in str(mde) and mde.X is None
    with pytest.raises(ValueError):
        mde.average_distortion()
    X = mde.embed(max_iter=30, snapshot_every=10, verbose=False)
    assert X is mde.X and X.is_cuda and X.shape == (n, 2)
    s = mde.solve_stats
    assert len(s.snapshots) == (s.iterations + 9) // 10 and not s.snapshots[0].is_cuda
    assert len(s.times) == s.iterations and s.solve_time > 0
    assert "iterations" in str(s)
    # warm start from the solution converges immediately-ish and does not move far
    X2 = mde.embed(X=X, max_iter=5)
    assert float(mde.average_distortion(X2)) <= s.average_distortions[-1] * (1 + 1e-5)
    with pytest.raises(ValueError):
        mde.embed(memory_size=0)
    # eps reached -> early stop
    mde.embed(eps=1e9, max_iter=50)
    assert mde.solve_stats.iterations == 1


def test_custom_constraint_and_callable_take_the_generic_path():
    """A user-defined Constraint object and a plain callable still solve (callbacks in Python,
    vectors on the device)."""
    import pymde_amd
    from pymde_amd import constraints

    class Scaled(constraints.Constraint):  # rows mean zero, Frobenius norm sqrt(n)
        def name(self):
            return "scaled"

        def initialization(self, n_items, embedding_dim, device=None):
            X = torch.randn((int(n_items), int(embedding_dim)), device=device)
            return self.project_onto_constraint(X)

        def project_onto_constraint(self, Z, inplace=True):
            W = Z if inplace else Z.clone()
            W.sub_(W.mean(0))
            W.mul_((W.shape[0] ** 0.5) / W.norm())
            return W

        def project_onto_tangent_space(self, X, Z, inplace=True):
            W = Z if inplace else Z.clone()
            W.sub_(W.mean(0))
            W.sub_(X * ((W * X).sum() / (X * X).sum()))
            return W

    rng = np.random.default_rng(0)
    n, p = 300, 2500
    i = rng.integers(0, n, p)
    j = (i + 1 + rng.integers(0, n - 1, p)) % n
    w = torch.tensor(rng.uniform(0.5, 2, p).astype(np.float32), device=DEV)
    mde = pymde_amd.MDE(n, 2, np.stack([i, j], 1), lambda d: w * d.pow(2), constraint=Scaled())
    torch.manual_seed(1)
    X = mde.embed(max_iter=40)
    E = mde.solve_stats.average_distortions
    assert E[-1] < 0.7 * E[0]
    assert abs(float(X.norm()) - n ** 0.5) < 1e-2 and abs(float(X.mean())) < 1e-5


def test_sharded_evaluation_sums_to_the_full_one():
    """Two vertex-range shards evaluated on this one GPU: their [grad | loss] buffers add up
    (bitwise for the gradient) to the unsharded evaluation -- what the RCCL all-reduce does."""
    import pymde_amd
    from pymde_amd import distributed
    from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate
    rng = np.random.default_rng(4)
    n, p, d = 5000, 80000, 2
    i = rng.integers(0, n, p)
    j = (i + 1 + rng.integers(0, n - 1, p)) % n
    edges = torch.tensor(np.stack([i, j], 1), device=DEV)
    w = torch.tensor(rng.uniform(0.5, 2, p).astype(np.float32), device=DEV)
    f = pymde_amd.penalties.Log1p(w)
    X = torch.tensor(rng.standard_normal((n, d)).astype(np.float32), device=DEV)
    full = torch.zeros(n * d + 1, device=DEV)
    fused_evaluate(Binding(EdgePlan(n, edges), f), X, full[:n * d].view(n, d), full[n * d:])
    bounds = distributed.shard_bounds(n, edges, 2)
    total = torch.zeros_like(full)
    for r in range(2):
        lo, hi = distributed.shard_range(bounds, r)
        buf = torch.zeros_like(full)
        fused_evaluate(Binding(EdgePlan(n, edges, lo, hi), f), X, buf[:n * d].view(n, d), buf[n * d:])
        assert float(buf[:lo * d].abs().sum()) == 0 and float(buf[hi * d:n * d].abs().sum()) == 0
        total += buf
    assert torch.equal(total[:n * d], full[:n * d])
    assert float(total[n * d]) == pytest.approx(float(full[n * d]), rel=1e-6)
    # ShardedMDE with world_size 1 behaves like MDE
    smde = distributed.ShardedMDE(n, d, edges, f, rank=0, world_size=1)
    Xt = X.clone().requires_grad_(True)
    E = smde.average_distortion(Xt)
    E.backward()
    assert torch.equal(Xt.grad.reshape(-1), full[:n * d]) and float(E) == pytest.approx(float(full[n * d]))


def test_spectral_initialiser(golden_spectral):
    # pymde/test_quadratic.py:67-109: same subspace as the ARPACK eigenvectors
    from pymde_amd import quadratic
    g = golden_spectral
    for key, n, m in (("small", 12, 3), ("mid", 400, 2)):
        torch.manual_seed(0)
        emb = quadratic.spectral(n, m, torch.tensor(g[key + "_edges"], device=DEV),
                                 torch.tensor(g[key + "_weights"], device=DEV)).double().cpu().numpy()
        np.testing.assert_allclose(emb.T @ emb / n, np.eye(m), atol=1e-4)
        want = g[key + "_emb"].astype(np.float64)
        Q, _ = np.linalg.qr(want)
        resid = emb - Q @ (Q.T @ emb)
        assert np.linalg.norm(resid) / np.linalg.norm(emb) < 5e-3, key  # (the reference tests rtol 1e-3 per vector)
    import pymde_amd
    n, m = 400, 2
    mde = pymde_amd.MDE(n, m, torch.tensor(g["mid_edges"], device=DEV),
                        pymde_amd.penalties.Quadratic(torch.tensor(g["mid_weights"], device=DEV)),
                        constraint=pymde_amd.Standardized())
    torch.manual_seed(0)
    emb = quadratic.spectral(n, m, mde.edges, torch.tensor(g["mid_weights"], device=DEV))
    assert float(mde.average_distortion(emb)) == pytest.approx(float(g["mid_value"]), rel=1e-3)


@pytest.mark.parametrize("which", ["neighbors", "distances"])
def test_trajectory_matches_reference_at_scale(which):
    """SURVEY 8c at a realistic size: n = 20k, p ~ 300k (the config-2 stand-in: PushAndPull(Log1p, Log),
    Standardized; a preserve_distances-shaped Huber problem, Centered).  The fixture holds the reference's
    first 8 iterations from X0 and from X0 perturbed by 1e-7 / 1e-6 / 1e-5 (tests/golden/make_golden.py:
    gen_trajectories_mid); edges and parameters are regenerated here from the same numpy seeds and
    checked against the fixture's checksums.  'neighbors' is stable in the reference itself (its four
    runs agree to 1e-4): the run must follow the UNPERTURBED reference run.  'distances' is not (the
    reference's own runs differ by 10 % at iteration 1 -- the line search branches on rounding): the run
    must follow one of them, and not only the 1e-5-perturbed one."""
    import importlib.util
    import pymde_amd
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    g = np.load(os.path.join(GOLDEN, "trajectories_mid.npz"))
    n, edges, par = mg.mid_problem_arrays(which)
    np.testing.assert_array_equal(g[which + "__edge_checksum"], [int(edges[:, 0].sum()), int(edges[:, 1].sum()), len(edges)])
    assert float(np.abs(par).astype(np.float64).sum()) == float(g[which + "__param_checksum"][0])
    pt = torch.tensor(par, device=DEV)
    if which == "neighbors":
        c = pymde_amd.Standardized()
        f = pymde_amd.penalties.PushAndPull(pt, pymde_amd.penalties.Log1p, pymde_amd.penalties.Log)
    else:
        c = pymde_amd.Centered()
        f = pymde_amd.losses.Huber(pt, 0.5)
    mde = pymde_amd.MDE(n, 2, torch.tensor(edges, device=DEV), f, constraint=c)
    X0 = torch.tensor(g[which + "__X0"], device=DEV)
    mde.embed(X=X0, max_iter=8, eps=1e-12, memory_size=10)
    s = mde.solve_stats
    E_ref, R_ref, S_ref = g[which + "__distortions"], g[which + "__residuals"], g[which + "__steps"]
    noise = list(g["noise"])
    assert s.average_distortions[0] == pytest.approx(E_ref[0, 0], rel=1e-5)
    assert s.residual_norms[0] == pytest.approx(R_ref[0, 0], rel=1e-4)
    k = 5
    matched = [i for i in range(E_ref.shape[0]) if _match_member(s.average_distortions, E_ref[i:i + 1], k, 1e-3) is not None]
    follow = {}
    for i in matched:
        m = k
        while m < min(E_ref.shape[1], len(s.average_distortions)) and \
                _match_member(s.average_distortions, E_ref[i:i + 1], m + 1, 1e-3) is not None:
            m += 1
        follow[i] = m
    print("trajectory at scale (%s): follows reference runs %s (X0 noise %s) for %s iterations; reference spread at "
          "iteration 1: %.1e" % (which, matched, [noise[i] for i in matched], [follow[i] for i in matched],
                                 (E_ref[:, 1].max() - E_ref[:, 1].min()) / abs(E_ref[:, 1].mean())))
    assert matched, (s.average_distortions[:k], E_ref[:, :k])
    assert any(noise[i] <= 1e-6 for i in matched), matched
    if which == "neighbors":
        assert 0 in matched and follow[0] == min(8, len(s.average_distortions)), (matched, follow)
    idx = matched[0]
    np.testing.assert_allclose(s.residual_norms[:k], R_ref[idx, :k], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(s.step_size_percents[:k], S_ref[idx, :k], rtol=5e-3, atol=1e-5)


def test_trajectory_on_planted_clusters_matches_reference():
    """A STRUCTURED problem of config-4 kind (round 6; the uniform-random graph of SURVEY 8d has a flat Standardized
    landscape): n = 100k items in 100 planted clusters, p ~ 1.8M, PushAndPull(Log1p, Log), Standardized -- on the ring
    kernel (the table does not fit L2).  The reference's first 6 iterations from X0 and from X0 perturbed by 1e-7 / 1e-6
    are in tests/golden/trajectories_clusters.npz (make_golden.py: gen_trajectories_clusters; the reference's own runs
    part by 0.7 % at iteration 4): the first 5 iterations agree with one member to rtol 1e-3, residual norms and step
    sizes with it, and the solve goes on to separate the clusters (the loss keeps falling, items sit nearer their
    cluster's centroid than the embedding's scale)."""
    import importlib.util
    import pymde_amd
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    g = np.load(os.path.join(GOLDEN, "trajectories_clusters.npz"))
    n, edges, w = mg.cluster_problem_arrays()
    X0n = mg.cluster_X0(n)
    np.testing.assert_array_equal(g["edge_checksum"], [int(edges[:, 0].sum()), int(edges[:, 1].sum()), len(edges)])
    assert float(np.abs(w).astype(np.float64).sum()) == float(g["param_checksum"][0])
    assert float(np.abs(X0n).astype(np.float64).sum()) == pytest.approx(float(g["X0_checksum"][0]), rel=1e-9)
    pen = pymde_amd.penalties
    f = pen.PushAndPull(torch.tensor(w, device=DEV), pen.Log1p, pen.Log)
    mde = pymde_amd.MDE(n, 2, torch.tensor(edges, device=DEV), f, constraint=pymde_amd.Standardized())
    mde.embed(X=torch.tensor(X0n, device=DEV), max_iter=6, eps=1e-12, memory_size=10)
    s = mde.solve_stats
    E_ref, R_ref, S_ref = g["distortions"], g["residuals"], g["steps"]
    k = 5
    idx = _match_member(s.average_distortions, E_ref, k, 1e-3)
    print("planted clusters: distortions", s.average_distortions, "follow reference run", idx, "(X0 noise %s)" % list(g["noise"]))
    assert idx is not None, (s.average_distortions, E_ref)
    # (residual norm and step size react to the members' parting a step earlier than the loss does: the reference's own
    # runs differ by 10 % in the residual norm at iteration 4)
    np.testing.assert_allclose(s.residual_norms[:k - 1], R_ref[idx, :k - 1], rtol=5e-3, atol=1e-7)
    np.testing.assert_allclose(s.step_size_percents[:k - 1], S_ref[idx, :k - 1], rtol=2e-2, atol=1e-5)
    # ... and the solve does something: 60 more iterations pull the clusters apart
    mde.embed(X=mde.X, max_iter=60, eps=1e-12)
    Es = np.array(mde.solve_stats.average_distortions)
    assert Es[-1] < 0.8 * E_ref[0, 0] and (np.diff(Es) <= 1e-6 * np.abs(Es[:-1])).all()
    X = mde.X.cpu().numpy().astype(np.float64)
    cen = X.reshape(100, 1000, 2).mean(1, keepdims=True)
    within = np.sqrt(((X.reshape(100, 1000, 2) - cen) ** 2).sum(2).mean())
    assert within < 0.5 * np.sqrt((X ** 2).sum(1).mean()), within


@pytest.mark.parametrize("cname", ["centered", "standardized", "anchored"])
@pytest.mark.parametrize("d", [2, 3, 64])
def test_row_sharded_engine_in_a_world_of_one(cname, d):
    """optim._ShardedEngine / _ShardedProblem (round 6: the solve with its vectors sharded by rows) with ONE rank
    owning every row and no collective: mde_lbfgs_dev_stage + _finish, the per-rank record reduced by mde_rank_reduce,
    centring as column sums + shift, the Standardized maps from Gram matrices -- against the single-GPU engine on
    the same problem: same first evaluation, the same trajectory to the line search's tolerance, the constraint
    satisfied at the end."""
    import pymde_amd
    from pymde_amd import distributed, optim
    rng = np.random.default_rng(17 + d)
    n, p = 6000, 50000
    i = rng.integers(0, n, p)
    j = (i + 1 + rng.integers(0, n - 1, p)) % n
    e = np.unique(np.sort(np.stack([i, j], 1), 1), axis=0)
    w = rng.choice(np.array([-1.0, 1.0, 2.0], dtype=np.float32), size=len(e), p=[0.3, 0.4, 0.3])
    edges = torch.tensor(e, device=DEV)
    pen = pymde_amd.penalties

    def make_c():
        if cname == "centered":
            return pymde_amd.Centered()
        if cname == "standardized":
            return pymde_amd.Standardized()
        return pymde_amd.Anchored(torch.tensor([3, 77, 4000], device=DEV),
                                  torch.tensor(rng0.standard_normal((3, d)).astype(np.float32), device=DEV))
    rng0 = np.random.default_rng(5)
    c1 = make_c()
    rng0 = np.random.default_rng(5)
    c2 = make_c()
    torch.manual_seed(0)
    x0 = c1.initialization(n, d, device=DEV)
    f = pen.PushAndPull(torch.tensor(w, device=DEV), pen.Log1p, pen.Log)
    single = pymde_amd.MDE(n, d, edges, f, constraint=c1)
    sharded = distributed.ShardedMDE(n, d, edges, f, constraint=c2, rank=0, world_size=1, force_exchange=True)
    assert optim._sharded_solver_args(sharded.average_distortion, c2) is not None
    # the constraint maps of the sharded problem against the single-GPU kernels, element by element (the trajectories
    # below part wherever the line search's cubic interpolation sits on a near-zero discriminant -- lbfgs.py:31-32 --,
    # so THIS is the check that the maps are the same maps)
    args = optim._sharded_solver_args(sharded.average_distortion, c2)
    eng = optim._ShardedEngine(x0, 10, *args[1:])
    prob = optim._ShardedProblem(eng, args[0], c2)
    xs = x0.clone().requires_grad_(True)
    single.average_distortion(xs).backward()
    gs = c1.project_onto_tangent_space(x0, xs.grad.clone(), inplace=True)
    prob.value_and_grad(eng.X)
    assert float((eng.g - gs).abs().max()) <= 5e-6 * float(gs.abs().max())
    eng.dir.copy_(-gs)
    for t in (1.0, 30.0):
        want = c1.project_onto_constraint(x0 - t * gs, inplace=False)
        prob.retract_step(t, eng.X_trial)
        assert float((eng.X_trial - want).abs().max()) <= 1e-5 * float(want.abs().max()), (cname, d, t)
    eng.close()
    iters = 12 if d <= 3 else 40
    single.embed(X=x0.clone(), max_iter=iters)
    Xr = sharded.embed(X=x0.clone(), max_iter=iters)
    a, b = np.array(single.solve_stats.average_distortions), np.array(sharded.solve_stats.average_distortions)
    np.testing.assert_allclose(b[0], a[0], rtol=1e-6)
    np.testing.assert_allclose(sharded.solve_stats.residual_norms[0], single.solve_stats.residual_norms[0], rtol=1e-4)
    if d <= 3:
        np.testing.assert_allclose(b[:3], a[:3], rtol=1e-4)
    assert (np.diff(b) <= 1e-6 * np.abs(b[:-1])).all() and abs(sharded.value - single.value) <= 5e-2 * abs(single.value), (a[-1], b[-1])
    Z = Xr.double()
    if cname != "anchored":
        assert float(Z.mean(0).abs().max()) < 1e-4
    if cname == "standardized":
        assert float((Z.T @ Z / n - torch.eye(d, device=DEV, dtype=torch.float64)).abs().max()) < 2e-4
    if cname == "anchored":
        assert torch.equal(Xr[torch.tensor([3, 77, 4000], device=DEV)], c2.values.to(Xr.device))


@pytest.fixture
def lb_knobs():
    """Sets the process-wide form of mde_lbfgs_dev_step for one test and restores the defaults."""
    from pymde_amd import _lib
    lib = _lib.load()
    yield lambda unfused=0, blocks=0, spins=0, lds=0: _lib.check(lib.mde_lbfgs_debug_knobs(unfused, blocks, spins, lds))
    _lib.check(lib.mde_lbfgs_debug_knobs(0, 0, 0, 0))


@pytest.mark.parametrize("N,unfused", [(3001, "fused"), (3001, "unfused"), (70001, "fused"), (70001, "unfused"),
                                       (3001, "rescue"), (70001, "rescue")])
def test_device_driven_lbfgs_step_matches_explicit_two_loop(lb_knobs, N, unfused):
    """mde_lbfgs_dev_step (history update, acceptance test and two-loop recursion all on the
    device) against the explicit recursion of lbfgs.py:468-507 in float64, incl. the history
    wrap-around (more than 8 pairs: two kernel groups) and a rejected pair (y.s <= 1e-10).  Both
    forms: the single launch with grid-wide arrival points that small vectors take (one / many
    workgroups), and the four launches of large vectors (forced through mde_lbfgs_debug_knobs).
    "rescue": the single launch with 512 workgroups of 100 KB of LDS each -- one per CU, i.e. at most
    256 of them resident on the chip -- and a spin limit of 2000 polls: the resident workgroups wait
    for the others at the first arrival point, give up, and the rescue kernel queued behind the
    launch redoes the step.  The direction must be the same (never a silently wrong one)."""
    import ctypes
    from pymde_amd import _lib, util
    lib = _lib.load()
    if unfused == "unfused":
        lb_knobs(unfused=1)
    elif unfused == "rescue":
        lb_knobs(blocks=512, spins=2000, lds=100 * 1024)
    rng = np.random.default_rng(0)
    A = rng.standard_normal((60, N))
    diag = rng.uniform(0.5, 2.0, N)

    def grad(x):  # SPD quadratic: y.s > 0
        return diag * x + A.T @ (A @ x) / 60.0

    for hist in (3, 10, 20):   # 20: the general (64-column) form of the direction step, always four launches
        h = ctypes.c_void_p()
        _lib.check(lib.mde_lbfgs_create(N, hist, ctypes.byref(h)))
        st = _lib.stream_ptr(torch.device(DEV))
        _lib.check(lib.mde_lbfgs_dev_reset(h, st))
        work = util.work_buffer(torch.device(DEV), 2)
        work.view(torch.int32)[2 * 2304 + 3 * 512 + 2] = 0
        board = torch.zeros(64, dtype=torch.float64, device=DEV)
        x = rng.standard_normal(N)
        g_prev_np = grad(x)
        d_np = -g_prev_np
        g_prev = torch.tensor(g_prev_np, dtype=torch.float32, device=DEV)
        d = torch.tensor(d_np, dtype=torch.float32, device=DEV)
        S, Y = [], []
        Hd = 1.0
        for step in range(14 if hist < 20 else 25):
            t = 0.3
            s32 = np.float32(t) * d.cpu().numpy()
            reject = step == 11
            if not reject:
                x = x + s32.astype(np.float64)
            g_np = grad(x).astype(np.float32) if not reject else g_prev.cpu().numpy().copy()
            g = torch.tensor(g_np, device=DEV)
            y32 = g_np - g_prev.cpu().numpy()
            _lib.check(lib.mde_lbfgs_dev_step(h, _lib.ptr(g), _lib.ptr(g_prev), _lib.ptr(d), t,
                                              _lib.ptr(d), _lib.ptr(board), _lib.ptr(work), st))
            cnt, acc = ctypes.c_int32(0), ctypes.c_int32(0)
            _lib.check(lib.mde_lbfgs_dev_info(h, ctypes.byref(cnt), ctypes.byref(acc), st))
            ys = float(np.dot(y32.astype(np.float64), s32.astype(np.float64)))
            assert bool(acc.value) == (ys > 1e-10) == (not reject)
            if ys > 1e-10:
                S.append(s32.astype(np.float64))
                Y.append(y32.astype(np.float64))
                if len(S) > hist:
                    S.pop(0)
                    Y.pop(0)
                Hd = ys / float(np.dot(Y[-1], Y[-1]))
            assert cnt.value == len(S)
            # explicit two-loop recursion
            q = -g_np.astype(np.float64)
            ro = [1.0 / Y[i].dot(S[i]) for i in range(len(S))]
            al = [0.0] * len(S)
            for i in range(len(S) - 1, -1, -1):
                al[i] = S[i].dot(q) * ro[i]
                q -= al[i] * Y[i]
            r = q * Hd
            for i in range(len(S)):
                r += (al[i] - Y[i].dot(r) * ro[i]) * S[i]
            got = d.cpu().numpy().astype(np.float64)
            assert np.abs(got - r).max() <= 2e-5 * np.abs(r).max(), (hist, step)
            assert torch.equal(g_prev, g)                       # g_prev <- g
            b = board.cpu().numpy()
            assert b[0] == pytest.approx(float(np.dot(g_np.astype(np.float64), got)), rel=1e-6)   # g.d
            assert b[5] == pytest.approx(float(np.dot(got, got)), rel=1e-6)                       # d.d
        if hist < 16 and unfused != "unfused":
            # (the work buffer's flag area: 3 x 512 arrival flags, then gave-up / done / rescue count)
            rescues = int(work.view(torch.int32)[2 * 2304 + 3 * 512 + 2])
            assert (rescues > 0) == (unfused == "rescue"), rescues
        _lib.check(lib.mde_lbfgs_destroy(h))


@pytest.mark.parametrize("cname", ["centered", "standardized", "anchored"])
def test_solve_is_bitwise_reproducible(cname):
    """Two solves of the same problem from the same start give identical iterates and statistics:
    every reduction on the path (loss, vector statistics, L-BFGS inner products, projections) has a
    fixed order and the fused kernels use no atomics on data."""
    import pymde_amd
    rng = np.random.default_rng(3)
    n, p = 20000, 150000
    i = rng.integers(0, n, p)
    j = (i + 1 + rng.integers(0, n - 1, p)) % n
    e = np.unique(np.sort(np.stack([i, j], 1), 1), axis=0)
    w = rng.choice(np.array([-1.0, 1.0, 2.0], dtype=np.float32), size=len(e), p=[0.3, 0.4, 0.3])
    edges = torch.tensor(e, device="cuda")

    def make():
        if cname == "anchored":
            anchors = torch.arange(0, 50, device="cuda")
            values = torch.tensor(rng2.standard_normal((50, 2)).astype(np.float32), device="cuda")
            return pymde_amd.Anchored(anchors, values)
        return pymde_amd.Centered() if cname == "centered" else pymde_amd.Standardized()

    results = []
    for _ in range(2):
        rng2 = np.random.default_rng(9)
        c = make()
        f = pymde_amd.penalties.PushAndPull(torch.tensor(w, device="cuda"))
        mde = pymde_amd.MDE(n, 2, edges, f, constraint=c)
        X0 = c.project_onto_constraint(torch.tensor(np.random.default_rng(4).standard_normal((n, 2)).astype(np.float32),
                                                    device="cuda"))
        X = mde.embed(X=X0.clone(), max_iter=60, eps=0.0, snapshot_every=25).clone()
        results.append((X, mde.solve_stats))
    (Xe, se), (Xg, sg) = results
    assert torch.equal(Xe, Xg), float((Xe - Xg).abs().max())
    assert se.average_distortions == sg.average_distortions
    assert se.residual_norms == sg.residual_norms
    assert se.step_size_percents == sg.step_size_percents
    assert all(torch.equal(a, b) for a, b in zip(se.snapshots, sg.snapshots))


@pytest.mark.parametrize("d", [1, 2, 3, 4, 8])
def test_fused_trial_point_and_projection_entry_points(d):
    """mde_center_step / mde_std_retract_step (the step X + t dir folded into the retraction) and
    mde_std_tangent_stats (statistics folded into the tangent projection) against the calls they
    replace: mde_axpy + mde_center / mde_std_retract bit for bit, mde_std_tangent bit for bit and
    mde_vec_stats to rounding (same doubles, another order of additions)."""
    from pymde_amd import _lib, util
    lib = _lib.load()
    dev = torch.device(DEV)
    st = _lib.stream_ptr(dev)
    torch.manual_seed(d)
    n = 30011
    X = torch.randn(n, d, device=dev)
    dirv = torch.randn(n, d, device=dev) * 0.1
    t = 0.37
    work = util.work_buffer(dev, d)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    # Centered
    want = torch.empty_like(X)
    _lib.check(lib.mde_axpy(n * d, t, _lib.ptr(dirv), _lib.ptr(X), _lib.ptr(want), st))
    _lib.check(lib.mde_center(n, d, _lib.ptr(want), _lib.ptr(work), st))
    got = torch.empty_like(X)
    _lib.check(lib.mde_center_step(n, d, _lib.ptr(X), _lib.ptr(dirv), t, _lib.ptr(got), _lib.ptr(work), st))
    assert torch.equal(got, want)
    # Standardized retraction
    want = torch.empty_like(X)
    _lib.check(lib.mde_axpy(n * d, t, _lib.ptr(dirv), _lib.ptr(X), _lib.ptr(want), st))
    _lib.check(lib.mde_std_retract(n, d, _lib.ptr(want), 1, _lib.ptr(work), _lib.ptr(status), st))
    got = torch.empty_like(X)
    _lib.check(lib.mde_std_retract_step(n, d, _lib.ptr(X), _lib.ptr(dirv), t, _lib.ptr(got), 1, _lib.ptr(work),
                                        _lib.ptr(status), st))
    assert torch.equal(got, want) and int(status[0]) == 0
    # tangent projection + statistics
    Xs = want                                   # a point on the constraint set
    G = torch.randn(n, d, device=dev)
    g_want = G.clone()
    b_want = torch.zeros(64, dtype=torch.float64, device=dev)
    _lib.check(lib.mde_std_tangent(n, d, _lib.ptr(Xs), _lib.ptr(g_want), _lib.ptr(work), st))
    _lib.check(lib.mde_vec_stats(n * d, _lib.ptr(g_want), _lib.ptr(dirv), _lib.ptr(Xs), _lib.ptr(b_want),
                                 _lib.ptr(work), st))
    for with_dir in (True, False):
        g_got = G.clone()
        b_got = torch.zeros(64, dtype=torch.float64, device=dev)
        _lib.check(lib.mde_std_tangent_stats(n, d, _lib.ptr(Xs), _lib.ptr(g_got), _lib.ptr(dirv) if with_dir else None,
                                             _lib.ptr(b_got), _lib.ptr(work), st))
        assert torch.equal(g_got, g_want)
        rows = range(8) if with_dir else (1, 2, 3, 4, 7)
        for q in rows:
            assert float(b_got[q]) == pytest.approx(float(b_want[q]), rel=1e-12, abs=1e-300), q


@pytest.mark.parametrize("cname,n,p", [("centered", 6000, 60000), ("standardized", 6000, 60000),
                                       ("centered", 140000, 700000), ("standardized", 140000, 700000)])
def test_turn_calls_follow_the_python_loop_bit_for_bit(monkeypatch, cname, n, p):
    """mde_turn_enqueue / mde_turn_wait (wait, launch of the next iteration inside the library, read-back and
    first-trial strong-Wolfe test by the iteration's last kernel, the next L-BFGS step queued behind it and
    gated on that test) against the same solve driven call by call from Python (MDE_NO_TURN=1): identical
    iterates and statistics, on a problem whose line search both accepts t = 1 at once and has to bracket /
    zoom.  n = 140k: vectors beyond the one-launch L-BFGS step (the gated kernels are the four-launch form)."""
    import pymde_amd
    rng = np.random.default_rng(17)
    i = rng.integers(0, n, p)
    j = (i + 1 + rng.integers(0, n - 1, p)) % n
    edges = torch.tensor(np.stack([i, j], 1), device=DEV)
    w = torch.tensor(np.where(rng.random(p) < 0.3, -1.0, rng.uniform(0.5, 2.0, p)).astype(np.float32), device=DEV)
    X0 = torch.tensor(rng.standard_normal((n, 2)).astype(np.float32), device=DEV)
    runs = []
    for no_turn in ("", "1"):
        if no_turn:
            monkeypatch.setenv("MDE_NO_TURN", no_turn)
        else:
            monkeypatch.delenv("MDE_NO_TURN", raising=False)
        c = pymde_amd.Centered() if cname == "centered" else pymde_amd.Standardized()
        mde = pymde_amd.MDE(n, 2, edges, pymde_amd.penalties.PushAndPull(w), constraint=c)
        X = mde.embed(X=c.project_onto_constraint(X0.clone()), max_iter=60, eps=0.0)
        st = mde.solve_stats
        runs.append((X.clone(), list(st.average_distortions), list(st.residual_norms), list(st.step_size_percents)))
    a, b = runs
    assert torch.equal(a[0], b[0])
    assert a[1] == b[1] and a[2] == b[2] and a[3] == b[3]
    # the search did more than accept t = 1 every time (otherwise this test checks less than it says)
    assert len(set(round(s / max(a[3]), 3) for s in a[3])) > 3
    # short solves, a convergence stop and snapshots: the look-ahead must not run past the end of the loop
    for kwargs in (dict(max_iter=1), dict(max_iter=2), dict(max_iter=3), dict(max_iter=40, eps=float(a[2][20])),
                   dict(max_iter=12, snapshot_every=5)):
        outs = []
        for no_turn in ("", "1"):
            if no_turn:
                monkeypatch.setenv("MDE_NO_TURN", no_turn)
            else:
                monkeypatch.delenv("MDE_NO_TURN", raising=False)
            c = pymde_amd.Centered() if cname == "centered" else pymde_amd.Standardized()
            mde = pymde_amd.MDE(n, 2, edges, pymde_amd.penalties.PushAndPull(w), constraint=c)
            X = mde.embed(X=c.project_onto_constraint(X0.clone()), **kwargs)
            outs.append((X.clone(), list(mde.solve_stats.average_distortions), len(mde.solve_stats.snapshots)))
        assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1] and outs[0][2] == outs[1][2], kwargs
        if "eps" in kwargs:
            assert len(outs[0][1]) < 40


def test_turn_wait_reports_a_pending_speculative_step():
    """mde_turn_enqueue(allow_pre = 1) queues the NEXT iteration's L-BFGS step behind a gate that opens when the
    trial is accepted.  If the caller then waits with allow_next = 0 the step has run all the same: out[20] must
    say so, and the next mde_turn_enqueue must take it as done -- not run a second step on g_prev == g (round-4
    advisor finding; the Python driver never gets there, the C API has to hold on its own)."""
    import ctypes
    import pymde_amd
    from pymde_amd import _lib, optim
    lib = _lib.load()
    rng = np.random.default_rng(2)
    n, d, p = 4000, 2, 40000
    i = rng.integers(0, n, p)
    j = (i + 1 + rng.integers(0, n - 1, p)) % n
    e_np = np.unique(np.sort(np.stack([i, j], 1), 1), axis=0)
    w = torch.tensor((1.0 + (rng.random(len(e_np)) < 0.3)).astype(np.float32), device=DEV)
    c = pymde_amd.Centered()
    mde = pymde_amd.MDE(n, d, torch.tensor(e_np, device=DEV), pymde_amd.penalties.Quadratic(w), constraint=c)
    torch.manual_seed(0)
    X0 = c.initialization(n, d, device=DEV)
    e = optim._Engine(X0, 10)
    prob = optim._NativeProblem(e, mde._binding(), c)
    T = prob.turn_desc()
    assert T is not None
    # the first iteration by hand (optim.lbfgs: evaluate, d = -g, t = min(1, 1 / |g|_1), accepted point = X + t d)
    prob.value_and_grad(e.X)
    e.stats(e.g, None, e.X)
    vals, loss0 = e.read_board(8)
    e.reset_memory()
    e.axpy(-2.0, e.g, e.g, e.dir)
    e.axpy(0.0, e.g, e.g, e.g_prev)
    t0 = float(min(1.0, 1.0 / vals[optim._G1]))
    prob.retract_step(t0, e.X_trial)
    e.X, e.X_trial = e.X_trial, e.X
    prob.value_and_grad(e.X)
    _, loss1 = e.read_board(8)
    bufs = (T.X[0], T.X[1])
    cur = 0 if e.X.data_ptr() == bufs[0] else 1
    out = np.zeros(24)
    outp = ctypes.c_void_p(out.ctypes.data)
    cnt, acc = ctypes.c_int32(0), ctypes.c_int32(0)
    pending_seen = False
    f = loss1
    t_prev = t0
    _lib.check(lib.mde_lbfgs_dev_info(e.lbfgs, ctypes.byref(cnt), ctypes.byref(acc), e._stream))
    expect = cnt.value                            # pairs in the history
    own_step = True                               # the first enqueue runs its own L-BFGS step
    for _ in range(6):
        _lib.check(lib.mde_turn_enqueue(ctypes.byref(T), cur, float(t_prev), float(f), 1e-4, 0.9, 1, e._stream))
        _lib.check(lib.mde_turn_wait(ctypes.byref(T), cur, float(f), 0, 1e-4, 0.9, 0.0, outp, e._stream))
        assert out[2] == 0.0                      # allow_next = 0: nothing launched
        expect = min(expect + (1 if own_step else 0), 10)
        _lib.check(lib.mde_lbfgs_dev_info(e.lbfgs, ctypes.byref(cnt), ctypes.byref(acc), e._stream))
        if out[1] == 0.0:
            assert out[20] == 0.0 and cnt.value == expect   # a rejected trial: the gate stayed shut
            break
        assert out[20] == 1.0, "accepted trial + allow_pre: the speculative step has run and must be reported"
        pending_seen = True
        # the gated step pushed ITS pair -- and an enqueue that found a pending step did not push another
        expect = min(expect + 1, 10)
        assert cnt.value == expect, (cnt.value, expect, own_step)
        assert torch.equal(e.g_prev, e.g)         # g_prev <- g happened in the speculative step
        cur = 1 - cur                             # the accepted trial becomes the iterate
        f, t_prev, own_step = float(out[0]), 1.0, False
    assert pending_seen
    torch.cuda.synchronize()
    e.close()
