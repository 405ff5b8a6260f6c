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
