"""Minimum-Distortion Embedding problems.

``MDE`` keeps the object API of the reference's ``pymde.problem.MDE``
[ref: pymde/problem.py:36-527]: same constructor arguments and validation, the registered
buffers (``n_items, embedding_dim, edges, p, X, _X_init``), ``differences / distances /
distortions / average_distortion / high_distortion_pairs`` and ``embed`` with its
``solve_stats``.  Everything that touches the embedding runs on the HIP kernels; the MDE
owns an ``EdgePlan`` built once from ``edges`` in place of the reference's ``_lhs/_rhs``
gather-index buffers.

Device semantics (this package is GPU-only): ``device=None`` places the problem on the
current GPU (CPU inputs are copied there); asking for ``device='cpu'`` raises.
"""
import copy
import logging
import sys
import typing as tp

import torch

from pymde_amd import average_distortion as _ad
from pymde_amd import constraints
from pymde_amd import optim
from pymde_amd import util

LOGGER = logging.getLogger("__pymde_amd__")
LOGGER.propagate = False
LOGGER.setLevel(logging.INFO)
if not LOGGER.handlers:
    _h = logging.StreamHandler(sys.stdout)
    _h.setLevel(logging.INFO)
    _h.setFormatter(logging.Formatter(fmt="%(asctime)s: %(message)s", datefmt="%b %d %I:%M:%S %p"))
    LOGGER.addHandler(_h)


def _module_device(module):
    bufs = list(module.buffers())
    if not bufs:
        return None
    dev = str(bufs[0].device)
    return dev if all(str(b.device) == dev for b in bufs) else None


_WARNED_F64 = False


class MDE(torch.nn.Module):
    """An MDE problem: ``n_items`` items embedded in ``R^embedding_dim``, a list of ``edges``
    (pairs i != j), a vector distortion function mapping the p embedding distances to p
    distortions, and a constraint (default: centered).

    After ``embed()``: ``X`` (the embedding), ``solve_stats``, ``value`` (average distortion)
    and ``residual_norm``.
    """

    def __init__(self, n_items: int, embedding_dim: int, edges: torch.Tensor,
                 distortion_function: tp.Callable,
                 constraint: tp.Optional[constraints.Constraint] = None,
                 device: tp.Optional[str] = None):
        super(MDE, self).__init__()
        if device is None:
            if (isinstance(edges, torch.Tensor) and edges.is_cuda
                    and isinstance(distortion_function, torch.nn.Module)
                    and str(edges.device) == str(_module_device(distortion_function))):
                device = edges.device
            else:
                device = util.get_default_device()
        self.device = util.require_cuda_device(device)

        n_items = torch.as_tensor(n_items).to(self.device)
        embedding_dim = torch.as_tensor(embedding_dim).to(self.device)
        self.register_buffer("n_items", n_items)
        self.register_buffer("embedding_dim", embedding_dim)
        self._n = int(n_items)
        self._d = int(embedding_dim)

        if edges is None:
            raise ValueError(
                "edges must be given: stochastic MDE problems are not supported "
                "(unreachable in the reference as well, see SURVEY section 2 #7).")
        if not isinstance(edges, torch.Tensor):
            edges = torch.tensor(edges, dtype=torch.int64, device=self.device)
        if edges.dim() != 2 or edges.shape[1] != 2:
            raise ValueError("edges must have shape (num_edges, 2)")
        if str(edges.device) != str(self.device):
            if device is not None and edges.is_cuda:
                LOGGER.warning("edges.device (%s) does not match requested device (%s); copying "
                               "edges to requested device." % (edges.device, self.device))
            edges = edges.to(self.device)
        edges = edges.to(torch.int64).contiguous()
        p = torch.tensor(edges.shape[0], device=self.device)

        # validates (self edges raise ValueError as in problem.py:134-140) and builds the plan
        self._plan = self._make_plan(edges)

        complete_graph_edges = (self._n * (self._n - 1)) // 2
        if int(p) > complete_graph_edges:
            raise ValueError(
                "Your graph has more than (n_items choose 2) edges."
                "(p: {0}, n_items choose 2: {1})".format(int(p), complete_graph_edges))

        self.register_buffer("edges", edges)
        self.register_buffer("p", p)
        self.register_buffer("_complete_graph_edges", torch.tensor(complete_graph_edges,
                                                                   device=self.device))

        if isinstance(distortion_function, torch.nn.Module):
            f_device = _module_device(distortion_function)
            if f_device is None or str(f_device) != str(self.device):
                if f_device is not None and f_device.startswith("cuda"):
                    LOGGER.warning(
                        "distortion_function device (%s) does not match requested device (%s); "
                        "making a copy of distortion_function" % (str(f_device), self.device))
                distortion_function = copy.deepcopy(distortion_function)
                distortion_function.to(self.device)
        self.distortion_function = distortion_function

        if constraint is None:
            constraint = constraints.Centered()
        self.constraint = constraint

        self.register_buffer("X", None)
        self.register_buffer("_X_init", None)
        self.solve_stats = None
        self.value = None
        self.residual_norm = None
        self.__binding = None
        self._reducer = None  # set by pymde_amd.distributed for multi-GPU solves

    # ------------------------------------------------------------------ plumbing
    def _make_plan(self, edges):
        return _ad.EdgePlan(self._n, edges)

    def _binding(self):
        """The distortion function bound to the plan (parameters permuted once)."""
        if self.__binding is None or self.__binding.function is not self.distortion_function:
            self.__binding = _ad.Binding(self._plan, self.distortion_function)
        return self.__binding

    def to(self, device):
        """Move the problem to another GPU."""
        device = util.require_cuda_device(device)
        super(MDE, self).to(device)
        if isinstance(self.distortion_function, torch.nn.Module):
            self.distortion_function.to(device)
        self.device = device
        self._plan = self._make_plan(self.edges)
        self.__binding = None

    def __str__(self):
        f = self.distortion_function
        func_name = f.__name__ if hasattr(f, "__name__") else type(f).__name__
        return ("MDE problem:\n\tn (number of items) {0}\n\tm (embedding dimension) {1}\n"
                "\tp (number of edges) {2}\n\tfraction of total edges {3:.1e}\n"
                "\t{4} distortion functions\n\tconstraint {5}\n\tdevice {6}".format(
                    self._n, self._d, int(self.p),
                    float(self.p) / max(float(self._complete_graph_edges), 1.0), func_name,
                    self.constraint.name(), self.device))

    def _repr_pretty_(self, p, cycle):
        del cycle
        p.text(self.__str__())

    def _embedding_arg(self, X):
        if X is None:
            X = self.X
        if X is None:
            raise ValueError(
                "Call this function after running the `embed` method, or "
                "provide a value for the embedding argument `X`")
        if X.device != self.device:
            X = X.to(self.device)
        if X.dtype == torch.float64:
            # the reference follows the dtype of X (average_distortion.py:96-98); this package
            # computes in float32: cast (differentiably) and say so once
            global _WARNED_F64
            if not _WARNED_F64:
                LOGGER.warning("pymde_amd computes in float32: float64 embeddings are cast to float32 "
                               "(gradients flow back through the cast)")
                _WARNED_F64 = True
            X = X.to(torch.float32)
        return X

    # ------------------------------------------------------------------ evaluators
    def differences(self, X):
        """``X[i] - X[j]`` for each edge (i, j); shape (n_edges, embedding_dim)."""
        return _ad.differences(self._embedding_arg(X), self._plan)

    def distances(self, X=None):
        """Embedding distances, one per edge (differentiable w.r.t. ``X``)."""
        return _ad.distances(self._embedding_arg(X), self._plan)

    def distortions(self, X=None):
        """Distortions, one per edge: ``distortion_function(distances(X))``."""
        return self.distortion_function(self.distances(self._embedding_arg(X)))

    def average_distortion(self, X=None):
        """The average distortion of ``X`` (a 0-dim tensor; differentiable w.r.t. ``X``)."""
        return _ad.average_distortion(self._embedding_arg(X), self._binding())

    def high_distortion_pairs(self, X=None):
        """Edges and their distortions sorted from the highest distortion to the lowest."""
        distortions = self.distortions(X).detach()
        order = torch.argsort(distortions, descending=True)
        return self.edges[order], distortions[order]

    # ------------------------------------------------------------------ the solve
    def embed(self, X=None, eps=1e-5, max_iter=300, memory_size=10, verbose=False,
              print_every=None, snapshot_every=None):
        """Compute an embedding; stores it in ``self.X`` and returns it.

        ``X``: optional initial iterate satisfying the constraint (default: ``_X_init`` set by a
        recipe, else ``constraint.initialization``).  ``eps``: residual-norm stopping
        threshold.  ``memory_size``: quasi-Newton memory.  ``snapshot_every``: keep CPU copies
        of the iterate in ``solve_stats.snapshots``.
        """
        if X is None and self._X_init is not None:
            X = self._X_init.detach().clone()
        elif X is None:
            X = self.constraint.initialization(self.n_items, self.embedding_dim, self.device)
        else:
            X = X.detach().clone()
        if X.device != self.device:
            if X.is_cuda:
                LOGGER.warning(
                    f"The initial iterate's device ({X.device}) does not match the requested "
                    f"device ({self.device}). Copying the iterate to {self.device}.")
            X = X.to(self.device)
        if max_iter < 0:
            raise ValueError("`max_iter` must be greater than 0")
        if memory_size <= 0:
            raise ValueError("`memory_size` must be greater than 0")
        if X.dtype != torch.float32:
            X = X.to(torch.float32)

        if verbose:
            LOGGER.info(f"Fitting a {self.constraint.name()} embedding into "
                        f"R^{self._d}, for a graph with {self._n} items and {int(self.p)} edges.")
            LOGGER.info(f"`embed` method parameters: eps={eps:.1e}, "
                        f"max_iter={max_iter}, memory_size={memory_size}")
        if print_every is None:
            print_every = max(1, max_iter // 10)
        # scalar parameters that live on the GPU are remembered between evaluations (functions/function.py
        # read_scalars); a solve starts from what the tensors hold now, however they were written
        from pymde_amd.functions import function as _function
        _function.invalidate_scalars(self.distortion_function)

        X_star, solve_stats = optim.lbfgs(
            X=X, objective_fn=self.average_distortion, constraint=self.constraint, eps=eps,
            max_iter=max_iter, memory_size=memory_size, use_line_search=True,
            use_cached_loss=True, verbose=verbose, print_every=print_every,
            snapshot_every=snapshot_every, logger=LOGGER)

        self.X = X_star
        self.solve_stats = solve_stats
        if solve_stats.iterations > 0:
            self.value = solve_stats.average_distortions[-1]
            self.residual_norm = solve_stats.residual_norms[-1]
        if verbose:
            LOGGER.info(f"Finished fitting in {solve_stats.solve_time:.3f} seconds "
                        f"and {solve_stats.iterations} iterations.")
            if solve_stats.iterations > 0:
                LOGGER.info(f"average distortion {self.value:.3g} | "
                            f"residual norm {self.residual_norm:.1e}")
        return self.X

    forward = embed
