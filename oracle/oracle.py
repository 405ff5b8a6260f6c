"""CPU oracle: plain restatements of the reference algorithm for the hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; nothing under ``pymde_amd/`` does.  The HIP
product path never routes through it.

Contents
  * ctypes wrapper of ``mde_oracle.c`` (edge-order forward / scatter-add backward for every
    penalty and loss)                      [ref: pymde/average_distortion.py:62-106,
                                                 pymde/functions/penalties.py, losses.py]
  * numpy restatements of the constraint maps
                                           [ref: pymde/constraints.py:94-200, util.py:129-171]
  * numpy restatement of the edge-plan layout (integer work: compared bit-exactly)
  * scipy restatement of the spectral initialiser
                                           [ref: pymde/quadratic.py:47-179]

Parity pin: ``tests/test_oracle.py`` checks every function here against golden vectors
produced by importing the reference itself (``tests/golden/make_golden.py``) and against the
reference's own known-answer tests.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmde_oracle.so")
_lib = None

KIND = dict(
    NONE=0, LINEAR=1, QUADRATIC=2, CUBIC=3, POWER=4, HUBER=5, LOGISTIC=6, SIGMOID=7, HINGE=8,
    LOG1P=9, LOG=10, INVPOWER=11, LOGRATIO=12, DEADZONE_QUADRATIC=13, DEADZONE_CUBIC=14,
    CLIPPED_QUADRATIC=15, L_QUADRATIC=32, L_WEIGHTED_QUADRATIC=33, L_HUBER=34, L_CUBIC=35,
    L_POWER=36, L_WEIGHTED_POWER=37, L_ABSOLUTE=38, L_LOGISTIC=39, L_FRACTIONAL=40,
    L_SOFT_FRACTIONAL=41, L_CLIPPED_QUADRATIC=42, L_LOG1P=43)


class _OracleFunc(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("kind_neg", ctypes.c_int32),
                ("a0", ctypes.c_void_p), ("a1", ctypes.c_void_p),
                ("a0_scalar", ctypes.c_int32), ("a1_scalar", ctypes.c_int32),
                ("s0", ctypes.c_float), ("s1", ctypes.c_float), ("s2", ctypes.c_float),
                ("n0", ctypes.c_float), ("n1", ctypes.c_float), ("n2", ctypes.c_float)]


def build(force=False):
    """Compile mde_oracle.c with gcc (OpenMP)."""
    src = os.path.join(_HERE, "mde_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmde_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.oracle_average_distortion.restype = ctypes.c_double
        L.oracle_average_distortion.argtypes = [
            ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
            ctypes.POINTER(_OracleFunc), ctypes.c_float, ctypes.c_void_p]
        L.oracle_distances.restype = None
        L.oracle_distances.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
        L.oracle_distortions.restype = None
        L.oracle_distortions.argtypes = [ctypes.c_int64, ctypes.c_void_p,
                                         ctypes.POINTER(_OracleFunc), ctypes.c_void_p]
        L.oracle_num_threads.restype = ctypes.c_int
        L.oracle_set_num_threads.argtypes = [ctypes.c_int]
        _lib = L
    return _lib


def func(kind, a0, a1=None, scalars=(0.0, 0.0, 0.0), kind_neg="NONE", scalars_neg=(0.0, 0.0, 0.0)):
    """A function descriptor: kind names as in KIND, per-edge arrays in EDGE order."""
    return dict(kind=kind, kind_neg=kind_neg, a0=np.ascontiguousarray(a0, dtype=np.float32).ravel(),
                a1=None if a1 is None else np.ascontiguousarray(a1, dtype=np.float32).ravel(),
                scalars=tuple(float(s) for s in scalars) + (0.0,) * (3 - len(scalars)),
                scalars_neg=tuple(float(s) for s in scalars_neg) + (0.0,) * (3 - len(scalars_neg)))


def _struct(fd):
    f = _OracleFunc()
    f.kind = KIND[fd["kind"]] if isinstance(fd["kind"], str) else int(fd["kind"])
    kn = fd.get("kind_neg", "NONE")
    f.kind_neg = KIND[kn] if isinstance(kn, str) else int(kn)
    f.a0 = fd["a0"].ctypes.data
    f.a0_scalar = 1 if fd["a0"].size == 1 else 0
    if fd.get("a1") is not None:
        f.a1 = fd["a1"].ctypes.data
        f.a1_scalar = 1 if fd["a1"].size == 1 else 0
    f.s0, f.s1, f.s2 = fd["scalars"]
    f.n0, f.n1, f.n2 = fd["scalars_neg"]
    return f


def average_distortion(edges, X, fd, grad_output=1.0, want_grad=True):
    """(E, grad) of the average distortion, edge-order scatter-add
    [ref: average_distortion.py:62-106]."""
    edges = np.ascontiguousarray(edges, dtype=np.int64)
    X = np.ascontiguousarray(X, dtype=np.float32)
    n, d = X.shape
    grad = np.zeros_like(X) if want_grad else None
    f = _struct(fd)
    E = lib().oracle_average_distortion(n, edges.shape[0], edges.ctypes.data, X.ctypes.data, d,
                                        ctypes.byref(f), float(grad_output),
                                        grad.ctypes.data if want_grad else None)
    return float(E), grad


def distances(edges, X):
    edges = np.ascontiguousarray(edges, dtype=np.int64)
    X = np.ascontiguousarray(X, dtype=np.float32)
    out = np.empty(edges.shape[0], dtype=np.float32)
    lib().oracle_distances(X.shape[0], edges.shape[0], edges.ctypes.data, X.ctypes.data, X.shape[1],
                           out.ctypes.data)
    return out


def distortions(dist, fd):
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    out = np.empty_like(dist)
    f = _struct(fd)
    lib().oracle_distortions(dist.size, dist.ctypes.data, ctypes.byref(f), out.ctypes.data)
    return out


def differences(edges, X):
    """X[i] - X[j] per edge [ref: problem.py:246-250]."""
    X = np.asarray(X, dtype=np.float32)
    return X[edges[:, 0]] - X[edges[:, 1]]


def distances_backward(edges, X, gout):
    """Backward of the 2-norm of the differences, NaN -> 0 [ref: average_distortion.py:46-52]."""
    X = np.asarray(X, dtype=np.float64)
    diff = X[edges[:, 0]] - X[edges[:, 1]]
    nrm = np.sqrt((diff ** 2).sum(1))
    with np.errstate(invalid="ignore", divide="ignore"):
        gi = diff * np.asarray(gout, dtype=np.float64)[:, None] / nrm[:, None]
    gi[np.isnan(gi)] = 0.0
    out = np.zeros_like(X)
    np.add.at(out, edges[:, 0], gi)
    np.add.at(out, edges[:, 1], -gi)
    return out.astype(np.float32)


# ---------------------------------------------------------------- constraints (numpy, float64)
def center(Z):
    """Z - column mean [ref: constraints.py:106-111]."""
    Z = np.asarray(Z, dtype=np.float64)
    return Z - Z.mean(axis=0)


def proj_standardized(X, demean=False):
    """sqrt(n) U V^T of the thin SVD [ref: util.py:129-161]."""
    X = np.asarray(X, dtype=np.float64)
    if demean:
        X = X - X.mean(axis=0)
    U, _, Vh = np.linalg.svd(X, full_matrices=False)
    return np.sqrt(X.shape[0]) * U @ Vh


def std_tangent(X, Z):
    """Z - (1/n) X (Z^T X) [ref: constraints.py:186-192]."""
    X = np.asarray(X, dtype=np.float64)
    Z = np.asarray(Z, dtype=np.float64)
    return Z - (1.0 / X.shape[0]) * X @ (Z.T @ X)


def anchor_tangent(Z, anchors):
    Z = np.array(Z, dtype=np.float64)
    Z[anchors, :] = 0.0  # constraints.py:143-150
    return Z


def anchor_retract(Z, anchors, values):
    Z = np.array(Z, dtype=np.float64)
    Z[anchors, :] = values  # constraints.py:152-164
    return Z


def sphere_tangent(X, Z, radius):
    """Z - (1/radius) diag(Z X^T) X, row by row [ref: constraints.py:214-223 -- the reference's scale is
    1/radius, not 1/radius^2]."""
    X = np.asarray(X, dtype=np.float64)
    Z = np.asarray(Z, dtype=np.float64)
    return Z - (1.0 / radius) * (Z * X).sum(axis=1)[:, None] * X


def sphere_retract(Z, radius):
    """radius Z / |Z|, row by row [ref: constraints.py:225-231]."""
    Z = np.asarray(Z, dtype=np.float64)
    return radius * Z / np.linalg.norm(Z, axis=1)[:, None]


# ---------------------------------------------------------------- edge plan (integer, bit-exact)
def plan_csr(n, edges, row_lo=0, row_hi=None):
    """The symmetrised incidence CSR the HIP plan must produce: rows v in [row_lo, row_hi),
    half-edges of a row ordered by original edge id then side (stable)."""
    edges = np.asarray(edges, dtype=np.int64)
    row_hi = n if row_hi is None else row_hi
    p = edges.shape[0]
    rows = np.empty(2 * p, dtype=np.int64)
    nbrs = np.empty(2 * p, dtype=np.int64)
    eids = np.repeat(np.arange(p, dtype=np.int64), 2)
    rows[0::2], nbrs[0::2] = edges[:, 0], edges[:, 1]
    rows[1::2], nbrs[1::2] = edges[:, 1], edges[:, 0]
    keep = (rows >= row_lo) & (rows < row_hi)
    rows, nbrs, eids = rows[keep], nbrs[keep], eids[keep]
    order = np.argsort(rows, kind="stable")
    rows, nbrs, eids = rows[order], nbrs[order], eids[order]
    rowptr = np.searchsorted(rows, np.arange(row_lo, row_hi + 1), side="left")
    return rowptr.astype(np.int32), nbrs.astype(np.int32), eids.astype(np.int32)


def shard_bounds(n, edges, world):
    """Vertex-range boundaries balancing the half-edge count (mde_shard_bounds)."""
    edges = np.asarray(edges, dtype=np.int64)
    deg = np.bincount(edges.ravel(), minlength=n).astype(np.int64)
    cum = np.cumsum(deg)
    total = int(cum[-1]) if n > 0 else 0
    bounds = [0]
    for r in range(1, world):
        target = (total * r) // world
        lo = int(np.searchsorted(cum, target, side="left"))
        bounds.append(min(lo + 1, n))
    bounds.append(n)
    for r in range(1, world + 1):
        bounds[r] = max(bounds[r], bounds[r - 1])
    return bounds


# ---------------------------------------------------------------- edge preprocessing (row f1)
def deduplicate_edges(edges):
    """Rows put in (min, max) order, unique rows in lexicographic order
    [ref: preprocess/preprocess.py:116-129]."""
    e = np.array(edges, dtype=np.int64, copy=True)
    flip = e[:, 0] > e[:, 1]
    e[flip] = e[flip][:, ::-1]
    return np.unique(e, axis=0)


def _neighbor_lists_to_graph(nbr, keep):
    """Directed neighbour lists [n, k] (+ a mask of the entries that count) -> unique edges
    i < j (sorted) with weights 1 / 2 (mutual) [ref: data_matrix.py:147-178, graph.py:578-587]."""
    n, k = nbr.shape
    items = np.repeat(np.arange(n), k)
    e = np.stack([items, nbr.ravel()], 1)[keep.ravel()]
    e = np.stack([e.min(1), e.max(1)], 1)
    uniq, counts = np.unique(e, axis=0, return_counts=True)
    return uniq.astype(np.int64), counts.astype(np.float32)


def knn_graph(data, k, max_distance=None):
    """Exact k-NN graph of a data matrix: unique edges i < j (sorted) and weights 1 / 2 (mutual);
    neighbours farther than max_distance do not count
    [ref: preprocess/data_matrix.py:91-178 with the sklearn brute-force branch]."""
    X = np.asarray(data, dtype=np.float64)
    sq = (X ** 2).sum(1)
    D = sq[:, None] + sq[None, :] - 2.0 * X @ X.T
    np.fill_diagonal(D, np.inf)
    nbr = np.argsort(D, axis=1, kind="stable")[:, :k]
    keep = np.ones(nbr.shape, dtype=bool)
    if max_distance is not None:
        keep = np.take_along_axis(D, nbr, 1) <= float(max_distance) ** 2
    return _neighbor_lists_to_graph(nbr, keep)


def graph_knn(n, edges, lengths=None, k=1, max_distance=None, direct=False):
    """k-NN graph of the nodes of a graph under its shortest-path metric (or, `direct`, among a
    node's own neighbours by edge length), ties to the smaller index; targets must be at finite
    positive distance <= max_distance [ref: preprocess/graph.py:502-587]."""
    import scipy.sparse as sp
    import scipy.sparse.csgraph as csgraph
    e = np.asarray(edges)
    w = np.ones(len(e)) if lengths is None else np.asarray(lengths, dtype=np.float64)
    D = np.full((n, n), np.inf)
    if direct:
        np.minimum.at(D, (e[:, 0], e[:, 1]), w)
        np.minimum.at(D, (e[:, 1], e[:, 0]), w)
    else:
        A = sp.coo_matrix((w, (e[:, 0], e[:, 1])), shape=(n, n))
        D = csgraph.shortest_path(A.maximum(A.T).tocsr(), directed=False)
    np.fill_diagonal(D, np.inf)
    D[~(D > 0)] = np.inf
    if max_distance is not None:
        D[D > max_distance] = np.inf
    nbr = np.argsort(D, axis=1, kind="stable")[:, :k]
    return _neighbor_lists_to_graph(nbr, np.isfinite(np.take_along_axis(D, nbr, 1)))


def pca(Y, m):
    """sqrt(n) * top-m left singular vectors of the column-centred Y [ref: quadratic.py:16-44]."""
    Y = np.asarray(Y, dtype=np.float64)
    U, _, _ = np.linalg.svd(Y - Y.mean(0)[None, :], full_matrices=False)
    return np.sqrt(float(Y.shape[0])) * U[:, :m]


def procrustes(X_source, X_target):
    """argmin_Q ||X_source Q - X_target||_F, Q orthogonal [ref: util.py:201-205]."""
    U, _, Vh = np.linalg.svd(np.asarray(X_target, np.float64).T @ np.asarray(X_source, np.float64),
                             full_matrices=False)
    return Vh.T @ U.T


def align(source, target):
    """[ref: util.py:302-331]"""
    S = np.asarray(source, dtype=np.float64)
    T = np.asarray(target, dtype=np.float64)
    mean = S.mean(0)
    S = S - mean
    norms = np.linalg.norm(S, axis=0)
    S = S / norms
    T = T - T.mean(0)
    T = T / np.linalg.norm(T, axis=0)
    return (S @ procrustes(S, T)) * norms + mean


def rotate(X, degrees):
    """[ref: util.py:208-299]"""
    X = np.asarray(X, dtype=np.float64)
    deg = np.atleast_1d(np.asarray(degrees, dtype=np.float64))
    if X.shape[1] == 2:
        t = np.deg2rad(deg[0])
        return X @ np.array([[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]])
    a, b, g = np.deg2rad(deg)
    rx = np.array([[1, 0, 0], [0, np.cos(a), np.sin(a)], [0, -np.sin(a), np.cos(a)]])
    ry = np.array([[np.cos(b), 0, -np.sin(b)], [0, 1, 0], [np.sin(b), 0, np.cos(b)]])
    rz = np.array([[np.cos(g), np.sin(g), 0], [-np.sin(g), np.cos(g), 0], [0, 0, 1]])
    return X @ (rx @ ry @ rz)


def shortest_path_pairs(n, edges, weights=None, max_length=None):
    """All pairs i < j at finite positive shortest-path distance (<= max_length), sorted by (i, j),
    with their distances [ref: preprocess/graph.py:345-474 with retain_fraction = 1]."""
    import scipy.sparse as sp
    import scipy.sparse.csgraph as csgraph
    e = np.asarray(edges)
    w = np.ones(len(e)) if weights is None else np.asarray(weights, dtype=np.float64)
    A = sp.coo_matrix((w, (e[:, 0], e[:, 1])), shape=(n, n))
    A = (A + A.T).tocsr()
    D = csgraph.shortest_path(A, directed=False)
    iu, ju = np.triu_indices(n, 1)
    d = D[iu, ju]
    ok = np.isfinite(d) & (d > 0)
    if max_length is not None:
        ok &= d <= max_length
    return np.stack([iu[ok], ju[ok]], 1).astype(np.int64), d[ok]


def check_sampled_edges(n, sampled, exclude=None):
    """Invariants every output of sample_edges satisfies [ref: preprocess/preprocess.py:11-80]:
    i < j, in range, no duplicates, disjoint from `exclude`.  Returns the canonical keys."""
    s = np.asarray(sampled, dtype=np.int64)
    assert s.ndim == 2 and s.shape[1] == 2
    assert (s[:, 0] < s[:, 1]).all() and s.min(initial=0) >= 0 and s.max(initial=0) < n
    keys = s[:, 0] * n + s[:, 1]
    assert len(np.unique(keys)) == len(keys)
    if exclude is not None and len(exclude):
        ex = np.asarray(exclude, dtype=np.int64)
        ek = np.minimum(ex[:, 0], ex[:, 1]) * n + np.maximum(ex[:, 0], ex[:, 1])
        assert not np.isin(keys, ek).any()
    return keys


# ---------------------------------------------------------------- spectral initialiser
def spectral(n, m, edges, weights):
    """Bottom m non-trivial Laplacian eigenvectors via ARPACK, centred and standardized
    [ref: quadratic.py:47-118 (cg=False), :173-179]."""
    import scipy.sparse
    import scipy.sparse.linalg
    edges = np.asarray(edges)
    w = np.asarray(weights, dtype=np.float32)
    A = scipy.sparse.coo_matrix((w, (edges[:, 0], edges[:, 1])), shape=(n, n), dtype=np.float32)
    A = (A + A.T).tocoo()
    L = -A
    L.setdiag(np.asarray(A.sum(axis=1)).squeeze())
    k = m + 1
    ncv = max(2 * k + 1, int(np.sqrt(n)))
    vals, vecs = scipy.sparse.linalg.eigsh(L, k, which="SM", ncv=ncv, tol=1e-4, maxiter=n * 5)
    order = np.argsort(vals)[1:k]
    emb = vecs[:, order]
    emb = emb - emb.mean(axis=0)
    return proj_standardized(emb)
