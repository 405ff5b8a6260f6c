"""Edge-list preprocessing on the GPU: de-duplication and negative-edge sampling.

Same functions as the reference's ``pymde/preprocess/preprocess.py`` [ref: preprocess.py:11-138]
(``sample_edges``, ``dissimilar_edges``, ``deduplicate_edges``, ``scale``), running on
``csrc/mde_edges.hip`` (64-bit edge keys, rocPRIM radix sort / unique / select).  Differences:
  * results live on the GPU and are sorted by (i, j) (the reference returns sampled edges in draw
    order; order carries no meaning);
  * the sampler draws with a counter-based generator, so a given ``seed`` reproduces the same
    edges here but not the reference's NumPy stream -- parity is distributional (uniform over the
    complement, no duplicates, no excluded edge, requested count).
"""
import ctypes

import torch

from pymde_amd import _lib
from pymde_amd import util


def _edges_on_device(edges, device=None):
    if not isinstance(edges, torch.Tensor):
        edges = torch.as_tensor(edges)
    if device is None:
        device = edges.device if edges.is_cuda else util.get_default_device()
    device = util.require_cuda_device(device)
    return edges.to(device=device, dtype=torch.int64).contiguous(), device


def deduplicate_edges(edges, n_items=None):
    """Unique edges with ``e[0] <= e[1]``, sorted lexicographically [ref: preprocess.py:116-129]."""
    edges, device = _edges_on_device(edges)
    p = edges.shape[0]
    if p == 0:
        return edges
    n = int(n_items) if n_items is not None else int(edges.max().item()) + 1
    n = max(n, 2)
    out = torch.empty_like(edges)
    count = ctypes.c_int64(0)
    lib = _lib.load()
    with torch.cuda.device(device):
        _lib.check(lib.mde_edges_deduplicate(n, p, _lib.ptr(edges), _lib.ptr(out), ctypes.byref(count),
                                             _lib.stream_ptr(device)))
    return out[:count.value]


def sample_edges(n, num_edges, exclude=None, seed=None, device=None):
    """Randomly sample ``num_edges`` distinct edges (i < j), none of them in ``exclude``
    [ref: preprocess.py:11-80].  The result may hold fewer rows when the complement of
    ``exclude`` is almost exhausted."""
    n, num_edges = int(n), int(num_edges)
    ex = None
    if exclude is not None:
        ex, device = _edges_on_device(exclude, device)
    elif device is None:
        device = util.get_default_device()
    device = util.require_cuda_device(device)
    n_excluded = 0 if ex is None else int(ex.shape[0])
    n_all = n * (n - 1) // 2
    if num_edges > n_all - n_excluded:
        raise ValueError(
            f"Cannot sample more than ({n} choose 2) - {n_excluded} ="
            f"{n_all - n_excluded} edges. (requested: {num_edges} edges)")
    if seed is None:
        seed = int(util.np_rng().integers(0, 2 ** 63 - 1))
    out = torch.empty((max(num_edges, 1), 2), dtype=torch.int64, device=device)
    count = ctypes.c_int64(0)
    lib = _lib.load()
    with torch.cuda.device(device):
        _lib.check(lib.mde_sample_edges(n, num_edges, ctypes.c_uint64(int(seed) & (2 ** 64 - 1)),
                                        _lib.ptr(ex), n_excluded, _lib.ptr(out), ctypes.byref(count),
                                        _lib.stream_ptr(device)))
    return out[:count.value]


def dissimilar_edges(n_items, similar_edges, num_edges=None, seed=None):
    """Sample edges not in ``similar_edges`` (as many as ``similar_edges`` by default)
    [ref: preprocess.py:83-113]."""
    if num_edges is None:
        num_edges = similar_edges.shape[0]
    return sample_edges(n_items, num_edges, exclude=similar_edges, seed=seed)


def _neighbor_lists_to_graph(n, k, idx, values, max_value, device):
    """Directed neighbour lists [n, k] -> unique undirected edges with weights 1 / 2."""
    lib = _lib.load()
    pairs = torch.empty((n * k, 2), dtype=torch.int64, device=device)
    edges = torch.empty_like(pairs)
    weights = torch.empty(pairs.shape[0], dtype=torch.float32, device=device)
    count = ctypes.c_int64(0)
    with torch.cuda.device(device):
        _lib.check(lib.mde_knn_pairs(n, k, _lib.ptr(idx), _lib.ptr(values) if max_value is not None else None,
                                     float(max_value) if max_value is not None else 0.0,
                                     _lib.ptr(pairs), _lib.stream_ptr(device)))
        _lib.check(lib.mde_edges_count_unique(n, pairs.shape[0], _lib.ptr(pairs), _lib.ptr(edges),
                                              _lib.ptr(weights), ctypes.byref(count),
                                              _lib.stream_ptr(device)))
    return edges[:count.value], weights[:count.value]


def k_nearest_neighbors(data, k, max_distance=None, device=None, graph_distances=True):
    """Exact k-nearest-neighbour graph of the rows of a data matrix (Euclidean distance)
    [ref: preprocess/data_matrix.py:91-178] or of the nodes of a ``Graph`` (shortest-path metric,
    ``pymde_amd.graph.k_nearest_neighbors``) [ref: preprocess/generic.py dispatch].

    Returns ``(edges, weights)`` on the GPU: unique edges i < j sorted by (i, j); the weight is 2
    when i and j are neighbours of each other, 1 when only one is a neighbour of the other.
    Neighbours farther than ``max_distance`` do not count (an edge whose both directions are too
    long disappears).  The reference is exact (sklearn brute force) below 10 000 items and
    approximate (pynndescent) above; this search is exact at every size.  Self matches are
    excluded by index, so duplicated rows become ordinary zero-distance neighbours."""
    if hasattr(data, "edges") and hasattr(data, "n_items") and not isinstance(data, torch.Tensor):
        from pymde_amd import graph as _graph
        return _graph.k_nearest_neighbors(data, k, graph_distances=graph_distances,
                                          max_distance=max_distance)
    if not isinstance(data, torch.Tensor):
        data = torch.as_tensor(data)
    if device is None:
        device = data.device if data.is_cuda else util.get_default_device()
    device = util.require_cuda_device(device)
    data = data.to(device=device, dtype=torch.float32).contiguous()
    n, nf = int(data.shape[0]), int(data.shape[1])
    k = int(k)
    if k > n - 1:
        k = n - 1
    if k < 1:
        raise ValueError("k must be at least 1")
    lib = _lib.load()
    idx = torch.empty((n, k), dtype=torch.int32, device=device)
    d2 = torch.empty((n, k), dtype=torch.float32, device=device)
    sqn = torch.empty(n, dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(lib.mde_knn(n, nf, _lib.ptr(data), k, _lib.ptr(idx), _lib.ptr(d2), _lib.ptr(sqn),
                               _lib.stream_ptr(device)))
    max_d2 = None if max_distance is None else float(max_distance) ** 2
    return _neighbor_lists_to_graph(n, k, idx, d2, max_d2, device)


def _rms(distances):
    return distances.pow(2).mean().sqrt()


def scale(distances, natural_length):
    """Rescale distances so that their RMS equals ``natural_length`` [ref: preprocess.py:132-138]."""
    return (natural_length / _rms(distances)) * distances
