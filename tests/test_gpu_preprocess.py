"""Edge-list preprocessing on the GPU: exact de-duplication, sampler invariants."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_deduplicate_matches_reference_bit_exact():
    from pymde_amd import preprocess
    g = load_golden("preprocess")
    got = preprocess.deduplicate_edges(torch.tensor(g["edges"]), n_items=int(g["n"]))
    assert got.is_cuda
    np.testing.assert_array_equal(got.cpu().numpy(), g["dedup"])
    # larger: 3M rows with heavy duplication, against the numpy restatement
    rng = np.random.default_rng(0)
    n = 20000
    e = rng.integers(0, n, (3_000_000, 2))
    e = e[e[:, 0] != e[:, 1]]
    got = preprocess.deduplicate_edges(torch.tensor(e, device=DEV), n_items=n).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.deduplicate_edges(e))
    assert preprocess.deduplicate_edges(torch.zeros((0, 2), dtype=torch.int64)).shape == (0, 2)


def test_sample_edges_invariants_and_uniformity():
    from pymde_amd import preprocess
    g = load_golden("preprocess")
    n = int(g["n"])
    s = preprocess.sample_edges(n, 5000, exclude=torch.tensor(g["exclude"]), seed=3)
    # the reference draws 5000 and drops the excluded ones (4922 survive here); the GPU sampler tops
    # the draw up to the requested count
    assert s.is_cuda and s.shape == (5000, 2) and int(g["ref_sample_count"]) <= 5000
    oracle.check_sampled_edges(n, s.cpu().numpy(), g["exclude"])
    # same seed -> same edges; different seed -> different edges
    s2 = preprocess.sample_edges(n, 5000, exclude=torch.tensor(g["exclude"]), seed=3)
    s3 = preprocess.sample_edges(n, 5000, exclude=torch.tensor(g["exclude"]), seed=4)
    assert torch.equal(s, s2) and not torch.equal(s, s3)
    # uniform over the n (n-1)/2 pairs: chi-square on the row index against its exact law
    n2, m = 2000, 400_000
    e = preprocess.sample_edges(n2, m, seed=11).cpu().numpy()
    oracle.check_sampled_edges(n2, e)
    assert len(e) == m
    counts = np.bincount(e[:, 0] // 100, minlength=20).astype(np.float64)
    rows = np.arange(n2)
    per_row = (n2 - 1 - rows).astype(np.float64)
    expect = np.add.reduceat(per_row, np.arange(0, n2, 100)) / per_row.sum() * m
    chi2 = ((counts - expect) ** 2 / expect).sum()
    assert chi2 < 60, chi2                                         # 19 dof: P(chi2 > 60) ~ 1e-6
    # dense exclusion: sample the whole complement but 50 edges
    allp = np.stack(np.triu_indices(60, 1), 1)
    excl = allp[:1000]
    s = preprocess.sample_edges(60, len(allp) - 1000 - 50, exclude=torch.tensor(excl), seed=1)
    oracle.check_sampled_edges(60, s.cpu().numpy(), excl)
    assert len(s) >= 0.9 * (len(allp) - 1050)
    with pytest.raises(ValueError, match="Cannot sample more than"):
        preprocess.sample_edges(10, 46)
    d = preprocess.dissimilar_edges(n, torch.tensor(g["exclude"]), seed=0)
    assert len(d) == len(g["exclude"])
    oracle.check_sampled_edges(n, d.cpu().numpy(), g["exclude"])


def test_negative_sampling_feeds_an_mde_problem():
    """The preserve_neighbors pattern (recipes.py:402-447): similar edges + as many sampled dissimilar
    edges, weights +1 / -1, PushAndPull -- built and solved entirely on the GPU."""
    import pymde_amd
    from pymde_amd import preprocess
    rng = np.random.default_rng(5)
    n = 20000
    src = np.repeat(np.arange(n), 10)
    dst = (src + rng.integers(1, 50, n * 10)) % n
    sim = preprocess.deduplicate_edges(torch.tensor(np.stack([src, dst], 1), device=DEV), n_items=n)
    dis = preprocess.dissimilar_edges(n, sim, seed=0)
    edges = torch.cat([sim, dis])
    w = torch.cat([torch.ones(len(sim), device=DEV), -torch.ones(len(dis), device=DEV)])
    pen = pymde_amd.penalties
    mde = pymde_amd.MDE(n, 2, edges, pen.PushAndPull(w, pen.Log1p, pen.Log), constraint=pymde_amd.Standardized())
    torch.manual_seed(0)
    mde.embed(max_iter=30)
    E = mde.solve_stats.average_distortions
    assert E[-1] < E[0]


def test_preserve_distances_recipe_matches_reference():
    """recipes.preserve_distances on a data matrix (rows f1 + f4): with every pair retained the
    problem is deterministic -- same edges and deviations as the reference's recipe."""
    import pymde_amd
    g = load_golden("preprocess")
    data = torch.tensor(g["pd_data"])
    for cname, c in (("centered", None), ("standardized", pymde_amd.Standardized())):
        mde = pymde_amd.preserve_distances(data, embedding_dim=2, loss=pymde_amd.losses.Absolute, constraint=c)
        np.testing.assert_array_equal(mde.edges.cpu().numpy(), g["pd_edges_" + cname])
        np.testing.assert_allclose(mde.distortion_function.deviations.cpu().numpy(),
                                   g["pd_deviations_" + cname], rtol=1e-6, atol=1e-7)
        torch.manual_seed(0)
        mde.embed(max_iter=50)
        assert mde.value < mde.solve_stats.average_distortions[0]
    # sampled pairs on a larger matrix: distances are exact for the sampled edges
    rng = np.random.default_rng(0)
    big = rng.standard_normal((5000, 64)).astype(np.float32)
    mde = pymde_amd.preserve_distances(torch.tensor(big), max_distances=200_000, seed=1)
    e = mde.edges.cpu().numpy()
    assert len(e) == 200_000
    oracle.check_sampled_edges(5000, e)
    want = np.linalg.norm(big[e[:, 0]].astype(np.float64) - big[e[:, 1]], axis=1)
    np.testing.assert_allclose(mde.distortion_function.deviations.cpu().numpy(), want, rtol=1e-5)


def test_knn_graph_matches_reference_and_oracle():
    """Row f2: exact k-NN graph (f32 MFMA Gram tiles + top-k) -- bit-identical edges and weights to
    the reference's sklearn brute-force branch, and to the oracle at sizes / widths that exercise
    partial tiles (n, nf not multiples of 64 / 32)."""
    from pymde_amd import preprocess
    g = load_golden("preprocess")
    e, w = preprocess.k_nearest_neighbors(torch.tensor(g["knn_data"]), 15)
    assert e.is_cuda
    np.testing.assert_array_equal(e.cpu().numpy(), g["knn_edges"])
    np.testing.assert_array_equal(w.cpu().numpy(), g["knn_weights"])
    rng = np.random.default_rng(3)
    for n, nf, k in ((1000, 37, 5), (3001, 784, 15), (130, 3, 64), (65, 130, 7)):
        X = rng.standard_normal((n, nf)).astype(np.float32)
        e, w = preprocess.k_nearest_neighbors(torch.tensor(X, device=DEV), k)
        oe, ow = oracle.knn_graph(X, k)
        np.testing.assert_array_equal(e.cpu().numpy(), oe)
        np.testing.assert_array_equal(w.cpu().numpy(), ow)


def test_preserve_neighbors_recipe_end_to_end():
    """recipes.preserve_neighbors on clustered data: k-NN graph, spectral init, negative sampling,
    PushAndPull, embed -- all on the GPU; the embedding keeps the clusters apart."""
    import pymde_amd
    rng = np.random.default_rng(0)
    centers = rng.standard_normal((6, 50)) * 6
    labels = rng.integers(0, 6, 6000)
    data = (centers[labels] + rng.standard_normal((6000, 50))).astype(np.float32)
    torch.manual_seed(0)
    mde = pymde_amd.preserve_neighbors(torch.tensor(data), embedding_dim=2, constraint=pymde_amd.Standardized(),
                                       seed=0)
    assert mde._X_init is not None and mde._X_init.shape == (6000, 2)
    w = mde.distortion_function.weights
    assert set(np.unique(w.cpu().numpy()).tolist()) <= {-1.0, 1.0, 2.0}
    X = mde.embed(max_iter=100).cpu().numpy()
    np.testing.assert_allclose(X.T @ X / 6000, np.eye(2), atol=1e-3)
    cent = np.stack([X[labels == c].mean(0) for c in range(6)])
    within = np.mean([np.linalg.norm(X[labels == c] - cent[c], axis=1).mean() for c in range(6)])
    between = np.linalg.norm(cent[:, None] - cent[None], axis=2)[np.triu_indices(6, 1)].mean()
    assert between > 3 * within, (between, within)
    # default constraint / random init / no repulsion variants construct as well
    m2 = pymde_amd.preserve_neighbors(torch.tensor(data[:500]), init="random", repulsive_penalty=None)
    assert isinstance(m2.constraint, pymde_amd.constraints._Standardized)
    with pytest.raises(ValueError):
        pymde_amd.preserve_neighbors(torch.tensor(data[:500]), init="bogus")
