"""Graphs as MDE inputs: the ``Graph`` container and shortest-path distances on the GPU.

``Graph`` keeps the role of the reference's ``pymde.Graph`` [ref: pymde/preprocess/graph.py:75-256]
for what the recipes need: unique undirected edges ``i < j`` with one positive value each (called
``weights`` or ``distances`` depending on the use), ``n_items``, ``from_edges`` (duplicate edges
have their values summed).  It lives on the GPU; its adjacency is the symmetrised CSR of an
``EdgePlan``.  ``shortest_paths`` [ref: graph.py:345-474] runs batched Bellman-Ford / BFS sweeps
(``csrc/mde_graph.hip``) instead of one scipy Dijkstra or Cython BFS per node in a process pool.
"""
import ctypes
import math

import torch

from pymde_amd import _lib
from pymde_amd import average_distortion as _ad
from pymde_amd import util


class Graph(object):
    """An undirected weighted graph.

    ``Graph(adjacency_matrix)`` takes what the reference's constructor takes [ref:
    graph.py:90-112]: a dense ``numpy`` / ``torch`` matrix or a scipy sparse matrix; ``inf``
    entries mean "no edge", the upper triangle defines the (undirected) edges, and a non-zero
    diagonal is an error.  Internally the graph is its unique edges ``i < j`` with one value each,
    on the GPU (``Graph._from_unique_edges``, ``Graph.from_edges``)."""

    def __init__(self, adjacency_matrix, device=None):
        import numpy as np
        try:
            import scipy.sparse as sp
        except ImportError:  # pragma: no cover
            sp = None
        A = adjacency_matrix
        if sp is not None and sp.issparse(A):
            A = A.tocoo()
            rows, cols, vals = np.asarray(A.row), np.asarray(A.col), np.asarray(A.data, dtype=np.float64)
            n = int(A.shape[0])
        else:
            if isinstance(A, torch.Tensor):
                A = A.detach().cpu().numpy()
            A = np.asarray(A)
            if A.ndim != 2 or A.shape[0] != A.shape[1]:
                raise ValueError("An adjacency matrix must be square; got shape %s" % (tuple(A.shape),))
            n = int(A.shape[0])
            rows, cols = np.nonzero(A)
            vals = np.asarray(A[rows, cols], dtype=np.float64)
        keep = (vals != 0) & ~np.isinf(vals)  # inf = unreachable = no edge (graph.py:104-106)
        rows, cols, vals = rows[keep], cols[keep], vals[keep]
        diag = rows == cols
        if (vals[diag] > 0).any():
            raise ValueError("Adjacency matrices must not contain self edges; "
                             "the following nodes were found to have self edges: ",
                             np.unique(rows[diag & (vals > 0)]))
        upper = rows < cols  # the lower triangle is redundant (graph.py:27-28)
        if device is None:
            device = util.get_default_device()
        device = util.require_cuda_device(device)
        e = torch.as_tensor(np.stack([rows[upper], cols[upper]], axis=1).astype(np.int64), device=device)
        g = Graph.from_edges(e.reshape(-1, 2), torch.as_tensor(vals[upper].astype(np.float32), device=device),
                             n_items=n, device=device)
        self.edges, self._values, self.n_items, self._plan = g.edges, g._values, n, None

    @classmethod
    def _from_unique_edges(cls, edges, values, n_items):
        """Unique edges ``i < j`` (sorted) with one value each, already on the GPU."""
        g = cls.__new__(cls)
        g.edges = edges
        g._values = values
        g.n_items = int(n_items)
        g._plan = None
        return g

    @staticmethod
    def from_edges(edges, weights=None, n_items=None, device=None):
        """Build a graph from an edge list; edges are put in ``i < j`` order, sorted, and the
        values of repeated edges are summed [ref: graph.py:51-72, :118-140]."""
        if not isinstance(edges, torch.Tensor):
            edges = torch.as_tensor(edges)
        if device is None:
            device = edges.device if edges.is_cuda else util.get_default_device()
        device = util.require_cuda_device(device)
        edges = edges.to(device=device, dtype=torch.int64).contiguous()
        if (edges[:, 0] == edges[:, 1]).any():
            raise ValueError("Adjacency matrices must not contain self edges")
        if n_items is None:
            n_items = int(edges.max().item()) + 1
        n_items = int(n_items)
        if weights is None:
            weights = torch.ones(edges.shape[0], dtype=torch.float32, device=device)
        else:
            weights = torch.as_tensor(weights).to(device=device, dtype=torch.float32).contiguous()
        lo = torch.minimum(edges[:, 0], edges[:, 1])
        hi = torch.maximum(edges[:, 0], edges[:, 1])
        key = lo * n_items + hi
        uniq, inverse = torch.unique(key, sorted=True, return_inverse=True)
        vals = torch.zeros(uniq.shape[0], dtype=torch.float32, device=device)
        vals.index_add_(0, inverse, weights)
        e = torch.stack([uniq // n_items, uniq % n_items], dim=1).contiguous()
        return Graph._from_unique_edges(e, vals, n_items)

    @property
    def weights(self):
        return self._values

    @property
    def distances(self):
        return self._values

    @property
    def n_edges(self):
        return int(self.edges.shape[0])

    @property
    def n_all_edges(self):
        return self.n_items * (self.n_items - 1) // 2

    def plan(self):
        if self._plan is None:
            self._plan = _ad.EdgePlan(self.n_items, self.edges)
        return self._plan

    def neighbors(self, node):
        rowptr, nbr, _ = self.plan().csr()
        return nbr[int(rowptr[node]):int(rowptr[node + 1])].to(torch.int64)


def shortest_paths(graph, retain_fraction=1.0, max_length=None, seed=0, verbose=False):
    """Shortest-path distances between pairs of nodes [ref: graph.py:345-474].

    Every pair at finite positive distance (``<= max_length`` if given) is kept with probability
    ``retain_fraction``; returns a ``Graph`` whose edges are the kept pairs and whose distances are
    the path lengths.  Unit-length graphs take the BFS path of the same kernel."""
    if not isinstance(graph, Graph):
        raise ValueError("`graph` must be a pymde_amd.Graph instance.")
    lib = _lib.load()
    plan = graph.plan()
    device = plan.device
    n = graph.n_items
    unweighted = bool((graph.distances == 1.0).all())
    w = None if unweighted else plan.expand(graph.distances, 0)
    frac = min(max(float(retain_fraction), 0.0), 1.0)
    expected = frac * graph.n_all_edges
    capacity = int(min(graph.n_all_edges, expected + 8.0 * math.sqrt(max(expected, 1.0)) + 1024))
    edges = torch.empty((max(capacity, 1), 2), dtype=torch.int64, device=device)
    dist = torch.empty(max(capacity, 1), dtype=torch.float32, device=device)
    count = ctypes.c_int64(0)
    with torch.cuda.device(device):
        _lib.check(lib.mde_graph_shortest_paths(
            plan.handle, _lib.ptr(w), float(max_length) if max_length is not None else 0.0,
            ctypes.c_double(frac if retain_fraction < 1.0 else 1.0),
            ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), capacity, _lib.ptr(edges), _lib.ptr(dist),
            ctypes.byref(count), _lib.stream_ptr(device)))
    m = count.value
    return Graph._from_unique_edges(edges[:m], dist[:m], n)


def k_nearest_neighbors(graph, k, graph_distances=False, max_distance=None, verbose=False):
    """k-nearest-neighbour graph of the nodes of ``graph`` [ref: graph.py:502-587]: neighbourhoods
    among a node's own graph neighbours by edge length (the default, as in the reference, and
    whenever the graph is already complete) or under the shortest-path metric
    (``graph_distances=True``, what the recipes ask for [ref: generic.py:97-107]), optionally
    restricted to radius ``max_distance``.  Returns ``(edges, weights)``: weight 2 for mutual
    neighbours, 1 otherwise.  Ties in distance go to the smaller node index (the reference's
    ``argsort`` leaves them unspecified)."""
    from pymde_amd import preprocess
    if not isinstance(graph, Graph):
        raise ValueError("`graph` must be a pymde_amd.Graph instance.")
    lib = _lib.load()
    plan = graph.plan()
    device = plan.device
    n = graph.n_items
    k = min(int(k), n - 1)
    if k < 1:
        raise ValueError("k must be at least 1")
    if graph.n_edges == graph.n_all_edges:
        graph_distances = False  # already a full distance matrix
    unweighted = bool((graph.distances == 1.0).all())
    w = None if unweighted else plan.expand(graph.distances, 0)
    idx = torch.empty((n, k), dtype=torch.int32, device=device)
    dist = torch.empty((n, k), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(lib.mde_graph_knn(plan.handle, _lib.ptr(w),
                                     float(max_distance) if max_distance is not None else 0.0,
                                     0 if graph_distances else 1, k, _lib.ptr(idx), _lib.ptr(dist),
                                     _lib.stream_ptr(device)))
    return preprocess._neighbor_lists_to_graph(n, k, idx, dist, None, device)
