"""Recipes that build MDE problems from data (the callers either side of the hot path).

``preserve_distances`` for data matrices [ref: pymde/recipes.py:103-218,
pymde/preprocess/data_matrix.py:11-88] is built from the GPU pieces of this package: the edge
sampler (``preprocess.sample_edges``, row f1) and the edge-order distance kernel with
``d = n_features`` (``mde_distances``, row f4); graph inputs use the batched shortest-path kernel (``graph.shortest_paths``,
row f3).  ``preserve_neighbors`` [ref: recipes.py:221-448] adds the exact GPU k-NN graph
(``preprocess.k_nearest_neighbors`` for data matrices, row f2; ``graph.k_nearest_neighbors`` under the
shortest-path metric for graphs, row f3) and the spectral initialiser; ``laplacian_embedding``
[ref: recipes.py:451-503] is the quadratic, standardized special case.
"""
import torch

from pymde_amd import _lib
from pymde_amd import constraints
from pymde_amd import graph as _graph
from pymde_amd import preprocess
from pymde_amd import problem
from pymde_amd import util
from pymde_amd import quadratic
from pymde_amd.functions import losses, penalties


class EdgeGraph(object):
    """Edges with one value per edge (the role ``pymde.Graph`` plays for recipe outputs)."""

    def __init__(self, edges, values, n_items):
        self.edges = edges
        self.distances = values
        self.weights = values
        self.n_items = int(n_items)


def distances(data, retain_fraction=1.0, seed=None, device=None):
    """Euclidean distances between (a sample of) the pairs of rows of a data matrix
    [ref: preprocess/data_matrix.py:11-88].  All ``n (n-1)/2`` pairs when ``retain_fraction >= 1``,
    otherwise a uniform sample of that fraction."""
    if not isinstance(data, torch.Tensor):
        data = torch.as_tensor(data)
    if device is None:
        device = data.device if data.is_cuda else util.get_default_device()
    device = util.require_cuda_device(device)
    data = data.to(device=device, dtype=torch.float32).contiguous()
    n, nf = int(data.shape[0]), int(data.shape[1])
    all_edges = n * (n - 1) // 2
    max_distances = int(retain_fraction * all_edges)
    if max_distances <= 0:
        raise ValueError("max_distances must be positive")
    if max_distances >= all_edges:
        edges = util.all_edges(n).to(device).contiguous()
    else:
        edges = preprocess.sample_edges(n, max_distances, seed=seed, device=device)
    lib = _lib.load()
    delta = torch.empty(edges.shape[0], dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(lib.mde_distances(n, edges.shape[0], _lib.ptr(edges), _lib.ptr(data), nf,
                                     _lib.ptr(delta), _lib.stream_ptr(device)))
    return EdgeGraph(edges, delta, n)


def _remove_anchor_anchor_edges(edges, data, anchors):
    """Drop edges whose two endpoints are both anchors: they are pinned by the constraint
    [ref: recipes.py:15-100]."""
    anchors = torch.as_tensor(anchors).to(edges.device)
    if anchors.numel() == 0:
        return edges, data
    both = torch.isin(edges[:, 0], anchors) & torch.isin(edges[:, 1], anchors)
    return edges[~both].contiguous(), data[~both].contiguous()


def preserve_distances(data, embedding_dim=2, loss=losses.Absolute, constraint=None,
                       max_distances=5e7, device=None, verbose=False, seed=None):
    """An MDE problem that preserves the pairwise Euclidean distances of a data matrix
    (rows = items) [ref: recipes.py:103-218].  At most ``max_distances`` pairs are used, sampled
    uniformly; with ``Standardized()`` the distances are rescaled to the constraint's natural
    length.  Call ``.embed()`` on the result."""
    is_graph = isinstance(data, _graph.Graph)
    if not is_graph and not isinstance(data, torch.Tensor) and not hasattr(data, "shape"):
        raise ValueError("`data` must be a np.ndarray/torch.Tensor data matrix, or a pymde_amd.Graph.")
    n_items = data.n_items if is_graph else int(data.shape[0])
    n_all_edges = n_items * (n_items - 1) / 2
    retain_fraction = max_distances / n_all_edges
    if verbose:
        problem.LOGGER.info(f"Computing {int(min(max_distances, n_all_edges))} distances")
    if is_graph:
        # original distance = length of the shortest path between the two nodes (row f3)
        graph = _graph.shortest_paths(data, retain_fraction=retain_fraction,
                                      seed=0 if seed is None else seed)
    else:
        graph = distances(data, retain_fraction=retain_fraction, seed=seed, device=device)
    edges, deviations = graph.edges, graph.distances
    if constraint is None:
        constraint = constraints.Centered()
    elif isinstance(constraint, constraints._Standardized):
        deviations = preprocess.scale(deviations,
                                      constraint.natural_length(n_items, embedding_dim).to(deviations.device))
    elif isinstance(constraint, constraints.Anchored):
        edges, deviations = _remove_anchor_anchor_edges(edges, deviations, constraint.anchors)
    return problem.MDE(n_items=n_items, embedding_dim=embedding_dim, edges=edges,
                       distortion_function=loss(deviations), constraint=constraint,
                       device=edges.device)


def preserve_neighbors(data, embedding_dim=2, attractive_penalty=penalties.Log1p,
                       repulsive_penalty=penalties.Log, constraint=None, n_neighbors=None,
                       repulsive_fraction=None, max_distance=None, init="quadratic", device=None,
                       verbose=False, seed=None):
    """An MDE problem that preserves the k-nearest-neighbour structure of a data matrix
    (rows = items) [ref: recipes.py:221-448]: k-NN graph (weights 1 / 2), optional spectral
    initialisation, uniformly sampled repulsive edges (weight -1), ``PushAndPull`` of the two
    penalties.  ``data`` may also be a ``Graph``: neighbourhoods are then taken under its
    shortest-path metric.  Every stage runs on the GPU (rows f2, f3, a10, f1 of SURVEY section 8)."""
    is_graph = isinstance(data, _graph.Graph)
    if not is_graph and not isinstance(data, torch.Tensor):
        data = torch.as_tensor(data)
    if device is None:
        if is_graph:
            device = data.edges.device
        else:
            device = data.device if data.is_cuda else util.get_default_device()
    device = util.require_cuda_device(device)
    n = int(data.n_items) if is_graph else int(data.shape[0])
    if n_neighbors is None:
        # the reference's default (recipes.py:318-321): about 1 % of all pairs as edges, within [5, 15]
        n_choose_2 = n * (n - 1) / 2
        n_neighbors = int(max(min(15, n_choose_2 * 0.01 / n), 5))
    if n_neighbors > n:
        problem.LOGGER.warning(
            "Requested n_neighbors {0} > number of items {1}. Setting n_neighbors to {2}".format(
                n_neighbors, n, n - 1))
        n_neighbors = n - 1
    if constraint is None and repulsive_penalty is not None:
        constraint = constraints.Centered()
    elif constraint is None and repulsive_penalty is None:
        constraint = constraints.Standardized()
    if is_graph and max_distance is None:
        # the reference bounds neighbourhoods on graphs (recipes.py:335-339)
        max_distance = (3 * torch.quantile(data.distances.float().cpu(), 0.75)).item()
    if verbose:
        problem.LOGGER.info(f"Computing {n_neighbors}-nearest neighbors, with max_distance={max_distance}")
    if is_graph:
        edges, weights = _graph.k_nearest_neighbors(data, k=n_neighbors, graph_distances=True,
                                                          max_distance=max_distance, verbose=verbose)
    else:
        edges, weights = preprocess.k_nearest_neighbors(data, k=n_neighbors, max_distance=max_distance,
                                                        device=device)
    if isinstance(constraint, constraints.Anchored):
        edges, weights = _remove_anchor_anchor_edges(edges, weights, constraint.anchors)
    if init == "quadratic":
        if verbose:
            problem.LOGGER.info(f"Computing {init} initialization.")
        X_init = quadratic.spectral(n, embedding_dim, edges, weights, max_iter=1000, device=device, cg=True)
        if not isinstance(constraint, (constraints._Centered, constraints._Standardized)):
            constraint.project_onto_constraint(X_init, inplace=True)
    elif init == "random":
        X_init = constraint.initialization(n, embedding_dim, device)
    else:
        raise ValueError(f"Unsupported value '{init}' for keyword argument `init`; "
                         "the supported values are 'quadratic' and 'random'.")
    if repulsive_penalty is not None:
        if repulsive_fraction is None:
            # the standardization constraint already spreads the points: use a lower repulsion
            repulsive_fraction = 0.5 if isinstance(constraint, constraints._Standardized) else 1
        n_choose_2 = n * (n - 1) // 2
        n_repulsive = min(int(repulsive_fraction * edges.shape[0]), n_choose_2 - edges.shape[0])
        negative_edges = preprocess.sample_edges(n, n_repulsive, exclude=edges, seed=seed, device=device)
        negative_weights = -torch.ones(negative_edges.shape[0], dtype=torch.float32, device=device)
        if isinstance(constraint, constraints.Anchored):
            negative_edges, negative_weights = _remove_anchor_anchor_edges(
                negative_edges, negative_weights, constraint.anchors)
        edges = torch.cat([edges, negative_edges])
        weights = torch.cat([weights, negative_weights])
        f = penalties.PushAndPull(weights, attractive_penalty=attractive_penalty,
                                  repulsive_penalty=repulsive_penalty)
    else:
        f = attractive_penalty(weights)
    mde = problem.MDE(n_items=n, embedding_dim=embedding_dim, edges=edges, distortion_function=f,
                      constraint=constraint, device=device)
    mde._X_init = X_init
    # overlapping points make the average distortion non-differentiable: perturb them apart
    if bool((mde.distances(mde._X_init) == 0).any()):
        mde._X_init = mde._X_init + 1e-4 * torch.randn(mde._X_init.shape, device=device,
                                                       dtype=mde._X_init.dtype)
    return mde


def laplacian_embedding(data, embedding_dim=2, n_neighbors=None, max_distance=None, init="quadratic",
                        device=None, verbose=False):
    """An MDE problem whose solution is a Laplacian embedding [ref: recipes.py:451-503]: the k-NN
    graph of ``preserve_neighbors`` with quadratic penalties, no repulsion and the standardization
    constraint."""
    return preserve_neighbors(data, embedding_dim=embedding_dim, attractive_penalty=penalties.Quadratic,
                              repulsive_penalty=None, n_neighbors=n_neighbors, max_distance=max_distance,
                              init=init, device=device, verbose=verbose)
