"""Small and awkward inputs end to end: nothing here is about speed."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_two_items_one_edge():
    import pymde_amd
    mde = pymde_amd.MDE(2, 1, torch.tensor([[0, 1]]), pymde_amd.losses.Absolute(torch.tensor([1.5])),
                        constraint=pymde_amd.Centered(), device=DEV)
    X = mde.embed(max_iter=100)
    assert X.shape == (2, 1) and torch.isfinite(X).all()
    assert abs(float(X[0] - X[1])) == pytest.approx(1.5, rel=1e-3)
    assert float(X.sum()) == pytest.approx(0.0, abs=1e-5)


def test_cpu_inputs_are_copied_to_the_gpu():
    import pymde_amd
    rng = np.random.default_rng(0)
    n = 300
    e = np.unique(np.sort(rng.integers(0, n, (2000, 2)), 1), axis=0)
    e = e[e[:, 0] != e[:, 1]]
    w = torch.tensor(rng.uniform(0.5, 2.0, len(e)).astype(np.float32))          # CPU tensors
    mde = pymde_amd.MDE(n, 2, torch.tensor(e), pymde_amd.penalties.Log1p(w), device=DEV)
    X0 = torch.randn(n, 2)                                                         # CPU start point
    X = mde.embed(X=X0, max_iter=30)
    assert X.is_cuda and X.shape == (n, 2) and mde.edges.is_cuda
    assert float(mde.average_distortion(X)) < float(mde.average_distortion(X0.to(DEV) - X0.to(DEV).mean(0)))


@pytest.mark.parametrize("d", [5, 64])
def test_wide_embedding_dimensions(d):
    import pymde_amd
    from oracle import oracle
    rng = np.random.default_rng(d)
    n = 500
    e = np.unique(np.sort(rng.integers(0, n - 7, (6000, 2)), 1), axis=0)         # last 7 items isolated
    e = e[e[:, 0] != e[:, 1]]
    w = rng.uniform(0.5, 2.0, len(e)).astype(np.float32)
    X = rng.standard_normal((n, d)).astype(np.float32)
    mde = pymde_amd.MDE(n, d, torch.tensor(e, device=DEV), pymde_amd.penalties.Cubic(torch.tensor(w, device=DEV)),
                        constraint=pymde_amd.Standardized())
    Xt = torch.tensor(X, device=DEV, requires_grad=True)
    E = mde.average_distortion(Xt)
    E.backward()
    wE, wg = oracle.average_distortion(e, X, oracle.func("CUBIC", w))
    assert float(E) == pytest.approx(wE, rel=1e-5)
    np.testing.assert_allclose(Xt.grad.cpu().numpy(), wg, rtol=1e-4, atol=1e-5 * np.abs(wg).max())
    assert float(Xt.grad[n - 7:].abs().sum()) == 0.0                               # isolated items
    Xe = mde.embed(max_iter=20)
    G = (Xe.T @ Xe / n).cpu().numpy()
    np.testing.assert_allclose(G, np.eye(d), atol=2e-3)


def test_preserve_distances_on_a_graph_with_sampling():
    import pymde_amd
    rng = np.random.default_rng(2)
    n = 400
    e = np.unique(np.sort(rng.integers(0, n, (1200, 2)), 1), axis=0)
    e = e[e[:, 0] != e[:, 1]]
    g = pymde_amd.Graph.from_edges(torch.tensor(e), n_items=n, device=DEV)
    full = pymde_amd.preserve_distances(g, embedding_dim=2)
    part = pymde_amd.preserve_distances(g, embedding_dim=2, max_distances=5000)
    assert part.edges.shape[0] <= full.edges.shape[0]
    assert 3500 <= part.edges.shape[0] <= 6500                                     # Bernoulli retention
    # the retained pairs are pairs of the full problem with the same deviations
    key = lambda E: (E[:, 0] * n + E[:, 1]).cpu().numpy()
    fk, pk = key(full.edges), key(part.edges)
    pos = np.searchsorted(fk, pk)
    assert (fk[pos] == pk).all()
    np.testing.assert_array_equal(full.distortion_function.deviations.cpu().numpy()[pos],
                                  part.distortion_function.deviations.cpu().numpy())
    X = part.embed(max_iter=50)
    assert torch.isfinite(X).all()


def test_errors_match_the_reference():
    import pymde_amd
    with pytest.raises(ValueError):
        pymde_amd.MDE(3, 2, torch.tensor([[0, 1], [0, 2], [1, 2], [0, 1]]),       # more than C(3, 2) edges
                      pymde_amd.penalties.Quadratic(torch.ones(4)), device=DEV)
    with pytest.raises(ValueError):
        pymde_amd.MDE(3, 2, torch.tensor([[0, 1, 2]]), pymde_amd.penalties.Quadratic(torch.ones(1)), device=DEV)
    with pytest.raises((ValueError, RuntimeError)):
        pymde_amd.MDE(3, 2, torch.tensor([[0, 1]]), pymde_amd.penalties.Quadratic(torch.ones(1)), device="cpu")


def test_graph_from_an_adjacency_matrix():
    """pymde.Graph(adjacency_matrix) [ref: preprocess/graph.py:90-112]: dense numpy / torch and
    scipy sparse inputs, inf = no edge, the upper triangle defines the edges, a non-zero diagonal
    is an error with the reference's message."""
    import scipy.sparse as sp
    import pymde_amd
    rng = np.random.default_rng(4)
    n = 60
    A = np.triu(rng.uniform(0.5, 3.0, (n, n)) * (rng.random((n, n)) < 0.15), 1)
    A = A + A.T
    A[3, 7] = A[7, 3] = np.inf                       # unreachable: dropped
    want = pymde_amd.Graph.from_edges(torch.tensor(np.argwhere(np.triu(np.where(np.isinf(A), 0, A), 1) > 0)),
                                      torch.tensor(A[np.triu(np.where(np.isinf(A), 0, A), 1) > 0].astype(np.float32)),
                                      n_items=n, device=DEV)
    for M in (A.copy(), torch.tensor(A), sp.csr_matrix(np.where(np.isinf(A), 0, A)), sp.coo_matrix(np.where(np.isinf(A), 0, A))):
        g = pymde_amd.Graph(M, device=DEV)
        assert g.n_items == n and g.n_edges == want.n_edges
        assert torch.equal(g.edges, want.edges)
        np.testing.assert_allclose(g.distances.cpu().numpy(), want.distances.cpu().numpy(), rtol=1e-6)
        assert (g.edges[:, 0] < g.edges[:, 1]).all()
    # a graph built this way goes through the recipes like one built from edges
    mde = pymde_amd.preserve_distances(pymde_amd.Graph(A, device=DEV), embedding_dim=2)
    assert mde.n_items == n
    B = A.copy()
    B[5, 5] = 1.0
    with pytest.raises(ValueError, match="Adjacency matrices must not contain self edges"):
        pymde_amd.Graph(np.where(np.isinf(B), 0, B), device=DEV)


def test_default_n_neighbors_is_the_references(monkeypatch):
    """recipes.py:318-321: about 1 % of all pairs, within [5, 15] -- k = 5 at n = 1000 (the 2 %-of-n
    rule this package used before gave 15), 15 from n = 3001 on."""
    import pymde_amd
    from pymde_amd import preprocess
    seen = []
    real = preprocess.k_nearest_neighbors

    def spy(data, k, **kw):
        seen.append(int(k))
        return real(data, k, **kw)
    monkeypatch.setattr(preprocess, "k_nearest_neighbors", spy)
    for n, want in ((1000, 5), (2001, 10), (4000, 15)):
        data = torch.randn((n, 6), device=DEV)
        pymde_amd.preserve_neighbors(data, embedding_dim=2, init="random")
        assert seen[-1] == want, (n, seen[-1], want)
        assert seen[-1] == int(max(min(15, (n * (n - 1) / 2) * 0.01 / n), 5))


def test_push_and_pull_with_arbitrary_penalties_and_solver_limits():
    import pymde_amd
    from pymde_amd.functions.function import Function

    class Sqrt(Function):          # no closed form in the kernels: a plain torch callable
        def __init__(self, weights):
            super(Sqrt, self).__init__()
            self.weights = weights

        def forward(self, distances):
            return self.weights * torch.sqrt(distances + 1.0)

    rng = np.random.default_rng(5)
    n, p = 200, 1500
    e = np.unique(np.sort(rng.integers(0, n, (p, 2)), 1), axis=0)
    e = e[e[:, 0] != e[:, 1]]
    w = torch.tensor(np.where(rng.random(len(e)) < 0.3, -1.0, 1.0).astype(np.float32), device=DEV)
    f = pymde_amd.penalties.PushAndPull(w, Sqrt, pymde_amd.penalties.Log)
    d = torch.tensor(rng.uniform(0.2, 2.0, len(e)).astype(np.float32), device=DEV)
    got = f(d)
    pos = w >= 0
    np.testing.assert_allclose(got[pos].cpu().numpy(), (w[pos] * torch.sqrt(d[pos] + 1.0)).cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(got[~pos].cpu().numpy(),
                               pymde_amd.penalties.Log(w[~pos])(d[~pos]).cpu().numpy(), rtol=1e-6)
    # the MDE takes the unfused path with it
    mde = pymde_amd.MDE(n, 2, torch.tensor(e, device=DEV), f)
    X = torch.randn((n, 2), device=DEV, requires_grad=True)
    E = mde.average_distortion(X)
    E.backward()
    assert torch.isfinite(E) and torch.isfinite(X.grad).all()
    # the device-resident L-BFGS holds at most 63 pairs; the reference has no cap (optim.py:69-82): a larger
    # memory_size is taken as 63 with a warning (round 6; rounds 3-5 raised), and the solve goes on
    with pytest.warns(UserWarning, match="63 L-BFGS pairs"):
        mde.embed(max_iter=2, memory_size=64)
    assert torch.isfinite(mde.X).all()
    # a sharded problem evaluates an arbitrary callable too (round 5): distances and f replicated, the scatter
    # over the owned rows, rank 0 carrying the mean -- here rank 0 of 2 without a process group: its own rows
    # of the gradient, and the full value
    from pymde_amd import distributed
    sh = distributed.ShardedMDE(n, 2, torch.tensor(e, device=DEV), f, device=DEV, rank=0, world_size=2)
    Xs = X.detach().clone().requires_grad_(True)
    Es = sh.average_distortion(Xs)
    Es.backward()
    lo, hi = sh._layout.ranges[0][0]
    assert float(Es) == pytest.approx(float(E), rel=1e-6)
    assert torch.equal(Xs.grad[lo:hi], X.grad[lo:hi])
