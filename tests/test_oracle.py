"""The oracle against the reference's golden vectors and known-answer tests (CPU)."""
import numpy as np
import pytest

from conftest import LOSS_RTOL, assert_grad_close, func_from_golden
from oracle import oracle


def test_known_answer_62_over_3(golden_functions):
    # pymde/test_optim.py:75-93
    e = np.array([[0, 1], [0, 2], [1, 2]])
    X = np.array([[0, 0], [1, 1], [3, 3]], dtype=np.float32)
    E, _ = oracle.average_distortion(e, X, oracle.func("QUADRATIC", [1.0, 2.0, 3.0]))
    assert E == pytest.approx(62.0 / 3, rel=1e-6)
    assert float(golden_functions["kat_62_3"]) == pytest.approx(62.0 / 3, rel=1e-6)


def test_gradient_matches_incidence_formula():
    # pymde/test_optim.py:97-118 / util.py:425-451: grad = A diag(g) A^T X
    rng = np.random.default_rng(0)
    e = np.array([[0, 1], [0, 2], [1, 2]])
    w = np.array([1.0, 2.0, 3.0], dtype=np.float32)
    X = rng.standard_normal((3, 2)).astype(np.float32)
    _, grad = oracle.average_distortion(e, X, oracle.func("QUADRATIC", w))
    A = np.array([[1, 1, 0], [-1, 0, 1], [0, -1, -1]], dtype=np.float64)
    g = 2.0 * w / 3.0  # f'(d)/(p d) for w d^2
    want = A @ (np.diag(g) @ (A.T @ X.astype(np.float64)))
    np.testing.assert_allclose(grad, want, rtol=1e-5, atol=1e-6)


def test_functions_against_reference(golden_functions):
    g = golden_functions
    edges = g["edges"]
    for name in g["names"]:
        fd = func_from_golden(g, str(name))
        for tag in ("d1", "d2", "d3", "d8", "zero"):
            X = g["X_zero"] if tag == "zero" else g["X_" + tag]
            E, grad = oracle.average_distortion(edges, X, fd)
            want_E = float(g["%s__%s__loss" % (name, tag)])
            want_grad = g["%s__%s__grad" % (name, tag)]
            if np.isfinite(want_E):
                assert E == pytest.approx(want_E, rel=LOSS_RTOL, abs=1e-7), (name, tag)
            else:
                assert not np.isfinite(E) or abs(E) > 1e30, (name, tag)
            assert_grad_close(grad, want_grad)
            dist = oracle.distances(edges, X)
            got = oracle.distortions(dist, fd)
            want = g["%s__%s__distortions" % (name, tag)]
            fin = np.isfinite(want)
            np.testing.assert_allclose(got[fin], want[fin], rtol=2e-5, atol=1e-6, err_msg=str((name, tag)))
            assert np.all(~np.isfinite(got[~fin]) | (np.abs(got[~fin]) > 1e30))


def test_distances_and_differences(golden_functions):
    g = golden_functions
    for tag in ("d1", "d2", "d3", "d8", "zero"):
        X = g["X_zero"] if tag == "zero" else g["X_" + tag]
        np.testing.assert_allclose(oracle.distances(g["edges"], X), g[tag + "__distances"], rtol=1e-6,
                                   atol=1e-7)
        np.testing.assert_array_equal(oracle.differences(g["edges"], X), g[tag + "__differences"])


def test_norm_grad_zero(golden_functions):
    # pymde/test_optim.py:57-71: the sub-gradient of the distance at coincident points is 0
    X = np.ones((3, 3), dtype=np.float32)
    got = oracle.distances_backward(np.array([[0, 1]]), X, np.ones(1))
    np.testing.assert_array_equal(got, golden_functions["norm_grad_zero"])


def test_constraints_against_reference(golden_constraints):
    g = golden_constraints
    for n, d in g["shapes"]:
        tag = "%dx%d" % (n, d)
        X, Z = g["X_" + tag], g["Z_" + tag]
        P = oracle.proj_standardized(X, demean=True)
        np.testing.assert_allclose(P, g["std_retract_" + tag], rtol=1e-3, atol=2e-4)
        np.testing.assert_allclose(P.T @ P / n, np.eye(d), atol=1e-8)
        T = oracle.std_tangent(g["std_retract_" + tag], Z)
        np.testing.assert_allclose(T, g["std_tangent_" + tag], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(oracle.center(X), g["centered_" + tag], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(
        oracle.anchor_tangent(g["anchor_Z"], g["anchors"]).astype(np.float32), g["anchor_tangent"])
    np.testing.assert_array_equal(
        oracle.anchor_retract(g["anchor_Z"], g["anchors"], g["anchor_values"]).astype(np.float32),
        g["anchor_retract"])


def test_plan_csr_restatement():
    rng = np.random.default_rng(3)
    n = 23
    edges = np.array([[i, j] for i in range(n) for j in range(i + 1, n) if rng.random() < 0.2])
    edges = np.concatenate([edges, edges[:5]])  # duplicates are allowed (only the count is checked)
    rowptr, nbr, eid = oracle.plan_csr(n, edges)
    assert rowptr[0] == 0 and rowptr[-1] == 2 * len(edges)
    for v in range(n):
        want = [(k, (j if i == v else i)) for k, (i, j) in enumerate(edges) if v in (i, j)]
        got = list(zip(eid[rowptr[v]:rowptr[v + 1]], nbr[rowptr[v]:rowptr[v + 1]]))
        assert got == want
    # shards tile the full plan
    b = oracle.shard_bounds(n, edges, 3)
    assert b[0] == 0 and b[-1] == n and b == sorted(b)
    parts = [oracle.plan_csr(n, edges, b[r], b[r + 1]) for r in range(3)]
    np.testing.assert_array_equal(np.concatenate([p[1] for p in parts]), nbr)
    np.testing.assert_array_equal(np.concatenate([p[2] for p in parts]), eid)


def test_spectral_against_reference(golden_spectral):
    g = golden_spectral
    for key, n, m in (("small", 12, 3), ("mid", 400, 2)):
        emb = oracle.spectral(n, m, g[key + "_edges"], g[key + "_weights"])
        want = g[key + "_emb"].astype(np.float64)
        # same subspace (eigenvectors are defined up to sign / rotation within clusters)
        Q, _ = np.linalg.qr(want)
        resid = emb - Q @ (Q.T @ emb)
        assert np.linalg.norm(resid) / np.linalg.norm(emb) < 5e-3


def test_edge_preprocessing_against_reference():
    # deduplicate_edges is exact; sample_edges is pinned through its invariants
    from conftest import load_golden
    g = load_golden("preprocess")
    np.testing.assert_array_equal(oracle.deduplicate_edges(g["edges"]), g["dedup"])
    keys = oracle.check_sampled_edges(int(g["n"]), g["ref_sample"], g["exclude"])
    # the reference draws 5000 and then drops the excluded ones: at most 5000 survive
    assert len(keys) == int(g["ref_sample_count"]) <= 5000


def test_knn_graph_against_reference():
    # the oracle's exact k-NN graph equals the reference's (sklearn brute force)
    from conftest import load_golden
    g = load_golden("preprocess")
    e, w = oracle.knn_graph(g["knn_data"], 15)
    np.testing.assert_array_equal(e, g["knn_edges"])
    np.testing.assert_array_equal(w, g["knn_weights"])


def _orient(A, B):
    """Flip the columns of A to the orientation of B (singular vectors are sign-free)."""
    return A * np.sign((A * B).sum(0))[None, :]


def test_api_helpers_against_reference():
    # pca / procrustes / align / rotate restatements vs the reference's outputs (api.npz)
    from conftest import load_golden
    g = load_golden("api")
    np.testing.assert_allclose(_orient(oracle.pca(g["pca_Y"], 3), g["pca_out"]), g["pca_out"], atol=5e-6)
    np.testing.assert_allclose(oracle.procrustes(g["align_source"], g["align_target"]),
                               g["procrustes_out"], atol=1e-6)
    np.testing.assert_allclose(oracle.align(g["align_source"], g["align_target"]), g["align_out"], atol=2e-5)
    np.testing.assert_allclose(oracle.rotate(g["rot_X2"], 30.0), g["rot2_out"], atol=1e-6)
    np.testing.assert_allclose(oracle.rotate(g["rot_X3"], [10.0, 20.0, 30.0]), g["rot3_out"], atol=1e-6)


def test_neighbor_graphs_against_reference():
    # k-NN with a radius (data matrix), graph k-NN (shortest-path / direct), and the graphs the
    # recipes laplacian_embedding / preserve_neighbors(Graph) build in the reference
    from conftest import load_golden
    g = load_golden("api")
    e, w = oracle.knn_graph(g["knnr_data"], 4, max_distance=1.6)
    np.testing.assert_array_equal(e, g["knnr_edges"])
    np.testing.assert_array_equal(w, g["knnr_weights"])
    n = int(g["g_n"])
    for tag, kw in (("sp", {}), ("spr", {"max_distance": 1.5}), ("direct", {"direct": True})):
        e, w = oracle.graph_knn(n, g["g_edges"], g["g_lengths"], k=3, **kw)
        np.testing.assert_array_equal(e, g["gknn_%s_edges" % tag])
        np.testing.assert_array_equal(w, g["gknn_%s_weights" % tag])
    # direct neighbours with a radius: the reference still hands out neighbours beyond the radius to
    # nodes with fewer than k inside it (argsort over +inf); the restatement keeps the documented
    # meaning, so its graph is the reference's minus those extra edges
    e, w = oracle.graph_knn(n, g["g_edges"], g["g_lengths"], k=3, direct=True, max_distance=1.2)
    ref = set(map(tuple, g["gknn_directr_edges"].tolist()))
    assert set(map(tuple, e.tolist())) <= ref and len(e) < len(ref)
    e, w = oracle.knn_graph(g["knnr_data"], 5)
    np.testing.assert_array_equal(e, g["lap_edges"])
    np.testing.assert_array_equal(w, g["lap_weights"])
    assert str(g["lap_constraint"]) == "_Standardized"
    e, w = oracle.graph_knn(n, g["g_edges"], g["g_lengths"], k=3)
    np.testing.assert_array_equal(e, g["png_edges_pos"])
    np.testing.assert_array_equal(w, g["png_weights_pos"])
