"""Shortest paths on graphs and the graph inputs of preserve_distances
(BASELINE configs 1 and 3 through the actual recipe)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cycle(n):
    return np.array([[i, (i + 1) % n] for i in range(n)])


def test_shortest_paths_exact_on_small_graphs():
    import pymde_amd
    from pymde_amd import graph as G
    rng = np.random.default_rng(0)
    # unweighted: cycle (long diameter), random sparse graph with several components
    for n, edges in ((257, _cycle(257)),
                     (400, np.unique(np.sort(rng.integers(0, 400, (500, 2)), 1), axis=0))):
        edges = edges[edges[:, 0] != edges[:, 1]]
        g = pymde_amd.Graph.from_edges(torch.tensor(edges), n_items=n)
        sp = G.shortest_paths(g)
        we, wd = oracle.shortest_path_pairs(n, g.edges.cpu().numpy())
        np.testing.assert_array_equal(sp.edges.cpu().numpy(), we)
        np.testing.assert_array_equal(sp.distances.cpu().numpy(), wd.astype(np.float32))
        # max_length cut (graph.py: distances beyond it are dropped)
        spm = G.shortest_paths(g, max_length=3)
        we, wd = oracle.shortest_path_pairs(n, g.edges.cpu().numpy(), max_length=3)
        np.testing.assert_array_equal(spm.edges.cpu().numpy(), we)
        np.testing.assert_array_equal(spm.distances.cpu().numpy(), wd.astype(np.float32))
    # weighted (Dijkstra branch of the reference)
    n = 300
    e = np.unique(np.sort(rng.integers(0, n, (1500, 2)), 1), axis=0)
    e = e[e[:, 0] != e[:, 1]]
    w = rng.uniform(0.5, 3.0, len(e)).astype(np.float32)
    g = pymde_amd.Graph.from_edges(torch.tensor(e), torch.tensor(w), n_items=n)
    sp = G.shortest_paths(g)
    we, wd = oracle.shortest_path_pairs(n, e, w)
    np.testing.assert_array_equal(sp.edges.cpu().numpy(), we)
    np.testing.assert_allclose(sp.distances.cpu().numpy(), wd, rtol=1e-5)
    # from_edges sums the values of repeated edges (graph.py:51-72)
    g2 = pymde_amd.Graph.from_edges(torch.tensor([[0, 1], [1, 0], [2, 1]]), torch.tensor([1.0, 2.0, 5.0]))
    assert g2.edges.cpu().tolist() == [[0, 1], [1, 2]] and g2.weights.cpu().tolist() == [3.0, 5.0]
    assert sorted(g2.neighbors(1).cpu().tolist()) == [0, 2]


def test_shortest_paths_retain_fraction_is_a_bernoulli_sample():
    import pymde_amd
    from pymde_amd import graph as G
    n = 1500
    g = pymde_amd.Graph.from_edges(torch.tensor(_cycle(n)), n_items=n)
    full_e, full_d = oracle.shortest_path_pairs(n, g.edges.cpu().numpy())
    sp = G.shortest_paths(g, retain_fraction=0.2, seed=5)
    e, d = sp.edges.cpu().numpy(), sp.distances.cpu().numpy()
    keys = oracle.check_sampled_edges(n, e)
    total = n * (n - 1) // 2
    assert abs(len(e) - 0.2 * total) < 6 * np.sqrt(0.2 * 0.8 * total)
    # the kept pairs carry the exact distances
    fk = full_e[:, 0] * n + full_e[:, 1]
    pos = np.searchsorted(fk, keys)
    np.testing.assert_array_equal(d, full_d[pos].astype(np.float32))
    # reproducible for a seed, different for another
    sp2 = G.shortest_paths(g, retain_fraction=0.2, seed=5)
    sp3 = G.shortest_paths(g, retain_fraction=0.2, seed=6)
    assert torch.equal(sp.edges, sp2.edges) and not torch.equal(sp.edges[:1000], sp3.edges[:1000])


def test_config1_preserve_distances_on_cycle_graph_via_recipe(golden_cycle):
    """BASELINE configs[0] through the recipe (scaled to n = 300 like the fixture): the cycle Graph,
    all-pairs shortest paths, Quadratic loss -- the same edges and deviations as the reference's
    preserve_distances, then the solve of tests/test_gpu_solver.py."""
    import pymde_amd
    g = golden_cycle
    n = int(g["n"])
    graph = pymde_amd.Graph.from_edges(torch.tensor(_cycle(n)), n_items=n)
    mde = pymde_amd.preserve_distances(graph, embedding_dim=2, loss=pymde_amd.losses.Quadratic)
    np.testing.assert_array_equal(mde.edges.cpu().numpy(), g["edges"])
    np.testing.assert_array_equal(mde.distortion_function.deviations.cpu().numpy(), g["deviations"])
    mde.embed(X=torch.tensor(g["X0"], device=DEV), max_iter=40, eps=1e-8)
    finals = g["final_value"]
    assert finals.min() * (1 - 1e-2) <= mde.value <= finals.max() * (1 + 1e-2)


def test_config3_sparse_graph_preserve_distances_huber_via_recipe():
    """BASELINE configs[2] stand-in (no Google Scholar offline): a 20k-node preferential-attachment
    graph, sampled shortest-path distances, Huber loss."""
    import functools
    import pymde_amd
    rng = np.random.default_rng(0)
    n, m = 20000, 4
    targets = list(range(m))
    repeated = []
    edges = []
    for v in range(m, n):                       # Barabasi-Albert
        for t in set(targets):
            edges.append((t, v))
        repeated.extend(targets)
        repeated.extend([v] * m)
        targets = [repeated[i] for i in rng.integers(0, len(repeated), m)]
    graph = pymde_amd.Graph.from_edges(torch.tensor(np.array(edges)), n_items=n)
    mde = pymde_amd.preserve_distances(graph, loss=functools.partial(pymde_amd.losses.Huber, threshold=1.0),
                                       max_distances=5e6, seed=1)
    p = int(mde.p)
    assert abs(p - 5e6) < 6 * np.sqrt(5e6)
    e = mde.edges.cpu().numpy()
    oracle.check_sampled_edges(n, e)
    dev = mde.distortion_function.deviations.cpu().numpy()
    assert dev.min() >= 1 and dev.max() < 30 and np.all(dev == np.round(dev))
    # spot-check 2000 of the sampled pairs against scipy BFS from their sources
    import scipy.sparse as sp
    import scipy.sparse.csgraph as cs
    ge = graph.edges.cpu().numpy()
    A = sp.coo_matrix((np.ones(len(ge)), (ge[:, 0], ge[:, 1])), shape=(n, n))
    A = (A + A.T).tocsr()
    pick = rng.choice(len(e), 2000, replace=False)
    srcs = np.unique(e[pick, 0])[:40]
    D = cs.shortest_path(A, directed=False, unweighted=True, indices=srcs)
    for si, s in enumerate(srcs):
        rows = np.nonzero(e[:, 0] == s)[0][:200]
        np.testing.assert_array_equal(dev[rows], D[si, e[rows, 1]].astype(np.float32))
    torch.manual_seed(0)
    mde.embed(max_iter=30)
    E = mde.solve_stats.average_distortions
    assert E[-1] < 0.5 * E[0]
