"""The reference's own recipe tests (pymde/test_recipes.py:12-160), run against this package on the
GPU: same inputs, same assertions."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_k_nearest_neighbors():                      # test_recipes.py:12-27
    from pymde_amd import preprocess
    data_matrix = np.array([[0.0], [1.0], [1.5], [1.75]], dtype=np.float32)
    edges, weights = preprocess.k_nearest_neighbors(torch.tensor(data_matrix, device=DEV), k=2)
    got = set(tuple(e) for e in edges.cpu().numpy().tolist())
    assert got == {(0, 1), (0, 2), (1, 2), (1, 3), (2, 3)}
    np.testing.assert_allclose(np.array([1.0, 1.0, 2.0, 2.0, 2.0]), weights.cpu().numpy())


def test_laplacian_embedding():                      # test_recipes.py:30-52
    import pymde_amd
    from pymde_amd import penalties, recipes, util
    torch.manual_seed(0)
    data_matrix = torch.randn(100, 10, device=DEV)
    util.seed(0)
    laplacian_emb = recipes.laplacian_embedding(data_matrix, device=DEV).embed()
    util.seed(0)
    also_laplacian = recipes.preserve_neighbors(data_matrix, attractive_penalty=penalties.Quadratic,
                                                repulsive_penalty=None, device=DEV).embed()
    also_laplacian = pymde_amd.align(source=also_laplacian, target=laplacian_emb)
    np.testing.assert_allclose(laplacian_emb.cpu().numpy(), also_laplacian.cpu().numpy(), rtol=1e-4, atol=1e-4)


def test_anchor_initialization():                    # test_recipes.py:55-87
    from pymde_amd import constraints, recipes, util
    n_items = 10
    util.seed(0)
    data_matrix = torch.randn(n_items, 5, device=DEV)
    anchors = torch.tensor([0, 1, 3], device=DEV)
    values = torch.tensor([2.0, 1.0, 3.0], device=DEV).reshape(3, 1)
    constraint = constraints.Anchored(anchors, values)
    for init in ("random", "quadratic"):
        mde = recipes.preserve_neighbors(data_matrix, embedding_dim=1, constraint=constraint, init=init,
                                         device=DEV)
        np.testing.assert_allclose(mde._X_init[anchors].cpu().numpy(), values.cpu().numpy())


def test_no_anchor_anchor_edges():                   # test_recipes.py:90-112
    from pymde_amd import constraints, recipes, util
    util.seed(0)
    data_matrix = torch.randn(3, 2, device=DEV)
    anchors = torch.tensor([0, 1], device=DEV)
    values = torch.tensor([2.0, 3.0], device=DEV).reshape(2, 1)
    constraint = constraints.Anchored(anchors, values)
    expected_edges = np.array([[0, 2], [1, 2]])
    mde = recipes.preserve_distances(data_matrix, embedding_dim=1, constraint=constraint, device=DEV)
    np.testing.assert_array_equal(expected_edges, mde.edges.cpu().numpy())
    mde = recipes.preserve_neighbors(data_matrix, embedding_dim=1, constraint=constraint, device=DEV)
    np.testing.assert_array_equal(expected_edges, mde.edges.cpu().numpy())


@pytest.mark.parametrize("n_items", [36, 1001])
def test_neighbor_reproducibility(n_items):          # test_recipes.py:115-135
    from pymde_amd import recipes, util
    torch.manual_seed(0)
    Y = torch.rand((n_items, 128), device=DEV)
    prev = None
    for _ in range(3):
        util.seed(0)
        mde = recipes.preserve_neighbors(Y, device=DEV)
        cur = (mde.edges.clone(), mde.distortion_function.weights.clone())
        if prev is not None:
            assert torch.equal(cur[0], prev[0]) and torch.equal(cur[1], prev[1])
        prev = cur


@pytest.mark.parametrize("n_items", [36, 1001])
def test_distances_reproducibility(n_items):         # test_recipes.py:138-160
    from pymde_amd import recipes, util
    torch.manual_seed(0)
    Y = torch.rand((n_items, 128), device=DEV)
    prev = None
    for _ in range(3):
        util.seed(0)
        mde = recipes.preserve_distances(Y, max_distances=1e5, device=DEV)
        cur = (mde.edges.clone(), mde.distortion_function.deviations.clone())
        if prev is not None:
            assert torch.equal(cur[0], prev[0]) and torch.equal(cur[1], prev[1])
        prev = cur


def _assert_close_up_to_sign(x, y, rtol=1e-4, atol=1e-5):
    try:
        np.testing.assert_allclose(x, y, rtol=rtol, atol=atol)
    except AssertionError:
        np.testing.assert_allclose(-x, y, rtol=rtol, atol=atol)


def test_pca():                                      # test_quadratic.py:10-64
    import pymde_amd
    torch.random.manual_seed(0)
    np.random.seed(0)
    for n, k in ((5, 5), (5, 4), (4, 5)):
        Y = np.random.randn(n, k).astype(np.float32)
        Y -= Y.mean(axis=0)
        top = min(n, k) - 1 if n <= k else k - 1      # the centred matrix has rank <= n - 1
        for m in range(1, top + 1):
            X = pymde_amd.pca(torch.tensor(Y, device=DEV), m).cpu().numpy()
            np.testing.assert_allclose(1.0 / n * X.T @ X, np.eye(m), rtol=1e-4, atol=2e-5)
            U, _, _ = np.linalg.svd(Y)
            X_unscaled = 1.0 / np.sqrt(n) * X
            for col in range(m):
                _assert_close_up_to_sign(X_unscaled[:, col], U[:, col], rtol=1e-3, atol=1e-4)
        with pytest.raises(ValueError, match=r"Embedding dimension must be at most.*"):
            pymde_amd.pca(torch.tensor(Y, device=DEV), min(n, k) + 1)


def test_rng():                                      # test_util.py:74-83
    from pymde_amd import util
    util.seed(0)
    tensor = torch.randn((10, 5))
    from_random_state = np.random.randn(10, 5)
    from_rng = util.np_rng().standard_normal((10, 5))
    util.seed(0)
    assert torch.equal(tensor, torch.randn((10, 5)))
    np.testing.assert_array_equal(from_random_state, np.random.randn(10, 5))
    np.testing.assert_array_equal(from_rng, util.np_rng().standard_normal((10, 5)))


def test_fit_spectral():                             # test_optim.py:122-154 (skipped upstream as flaky on macOS)
    import pymde_amd
    from pymde_amd import penalties, quadratic, util
    util.seed(0)
    n, m, max_iter = 200, 3, 1000
    edges = util.all_edges(n).to(DEV)
    weights = torch.ones(edges.shape[0], device=DEV)
    mde = pymde_amd.MDE(n, m, edges=edges, distortion_function=penalties.Quadratic(weights),
                        constraint=pymde_amd.Standardized(), device=DEV)
    X = mde.embed(max_iter=max_iter, eps=1e-10, memory_size=10)
    assert X is mde.X
    X_spectral = quadratic.spectral(n, m, edges=edges, weights=weights, device=DEV)
    np.testing.assert_allclose(float(mde.average_distortion(X)), float(mde.average_distortion(X_spectral)),
                               atol=1e-4)
