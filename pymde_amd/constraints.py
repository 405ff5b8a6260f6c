"""Constraints on the embedding: Centered, Anchored, Standardized.

The ``Constraint`` protocol -- ``name / initialization / project_onto_constraint /
project_onto_tangent_space`` with ``inplace`` flags -- is the reference's
[ref: pymde/constraints.py:7-91]; the built-in constraints run on the HIP kernels of
``csrc/mde_vec.hip`` (column-mean reduction, d x d Gram matrices -- f32 MFMA when d is a
multiple of 32 --, d x d inverse square root, row-tile right-multiply).

Differences to the reference, both exact in exact arithmetic:
  * ``Standardized`` retracts with sqrt(n) Z C^{-1/2}, C = Z^T Z, instead of a thin SVD
    (sqrt(n) U V^T) [ref: util.py:129-161]; the reference notes this alternative at
    util.py:144-147.
  * ``Standardized().initialization`` draws randn on the GPU and applies that same
    retraction (the reference multiplies by Q diag(lambda^-1/2), which differs by a rotation).
"""
import abc

import torch

from pymde_amd import _lib
from pymde_amd import util


class Constraint(abc.ABC):
    """A generic constraint.  Subclass and implement the four methods to define your own;
    custom constraints run through the generic (Python-callback) solver path."""

    @abc.abstractmethod
    def name(self) -> str:
        raise NotImplementedError

    @abc.abstractmethod
    def initialization(self, n_items: int, embedding_dim: int, device=None) -> torch.Tensor:
        """A random embedding of shape (n_items, embedding_dim) in the constraint set."""
        raise NotImplementedError

    @abc.abstractmethod
    def project_onto_constraint(self, Z: torch.Tensor, inplace=True) -> torch.Tensor:
        """Project (retract) ``Z`` onto the constraint set."""
        raise NotImplementedError

    @abc.abstractmethod
    def project_onto_tangent_space(self, X: torch.Tensor, Z: torch.Tensor,
                                   inplace=True) -> torch.Tensor:
        """Euclidean projection of ``Z`` onto the tangent space of the set at ``X``."""
        raise NotImplementedError


def _randn(n_items, embedding_dim, device):
    device = util.require_cuda_device(device if device is not None else util.get_default_device())
    return torch.randn((int(n_items), int(embedding_dim)), device=device, dtype=torch.float32)


def _prepare(Z, inplace):
    """(work tensor, device): a contiguous float32 CUDA tensor the kernels may overwrite."""
    device = util.require_cuda_device(Z.device)
    if Z.dtype != torch.float32:
        raise ValueError("pymde_amd constraints expect float32 embeddings, got %s" % Z.dtype)
    if inplace:
        if not Z.is_contiguous():
            raise ValueError("in-place projection needs a contiguous tensor")
        return Z, device
    return Z.detach().clone().contiguous(), device


class _Centered(Constraint):
    def name(self):
        return "centered"

    def initialization(self, n_items, embedding_dim, device=None):
        X = _randn(n_items, embedding_dim, device)
        return self.project_onto_constraint(X, inplace=True)

    def project_onto_tangent_space(self, X, Z, inplace=True):
        del X  # the tangent space of the centring "constraint" is everything (constraints.py:102-104)
        return Z

    def project_onto_constraint(self, Z, inplace=True):
        W, device = _prepare(Z, inplace)
        lib = _lib.load()
        n, d = W.shape
        with torch.no_grad(), torch.cuda.device(device):
            _lib.check(lib.mde_center(n, d, _lib.ptr(W), _lib.ptr(util.work_buffer(device, d)),
                                      _lib.stream_ptr(device)))
        return W


class Anchored(Constraint):
    """Pin some embedding vectors (the anchors) to given values."""

    def __init__(self, anchors, values):
        super(Anchored, self).__init__()
        self.anchors = anchors
        self.values = values
        self._cache = None

    def name(self):
        return "anchored"

    def _device_args(self, device):
        c = self._cache
        ver = (getattr(self.anchors, "_version", None), getattr(self.values, "_version", None))
        if (c is None or c[0] != str(device) or c[3] is not self.anchors or c[4] is not self.values
                or c[5] != ver):  # (values edited in place are picked up through the version counter)
            a = torch.as_tensor(self.anchors).to(device=device, dtype=torch.int64).contiguous()
            v = torch.as_tensor(self.values).to(device=device, dtype=torch.float32).contiguous()
            self._cache = c = (str(device), a, v, self.anchors, self.values, ver)
        return c[1], c[2]

    def _write(self, W, device, with_values):
        a, v = self._device_args(device)
        lib = _lib.load()
        with torch.no_grad(), torch.cuda.device(device):
            _lib.check(lib.mde_anchor_rows(a.numel(), W.shape[1], _lib.ptr(a),
                                           _lib.ptr(v) if with_values else None, _lib.ptr(W),
                                           _lib.stream_ptr(device)))
        return W

    def initialization(self, n_items, embedding_dim, device=None):
        X = _randn(n_items, embedding_dim, device)
        return self._write(X, X.device, True)

    def project_onto_tangent_space(self, X, Z, inplace=True):
        del X
        W, device = _prepare(Z, inplace)
        return self._write(W, device, False)  # zero the anchor rows (constraints.py:143-150)

    def project_onto_constraint(self, Z, inplace=True):
        W, device = _prepare(Z, inplace)
        return self._write(W, device, True)  # rewrite the anchor rows (constraints.py:152-164)


class _Standardized(Constraint):
    """Centered, with (1/n) X^T X = I."""

    def name(self):
        return "standardized"

    def initialization(self, n_items, embedding_dim, device=None):
        X = _randn(n_items, embedding_dim, device)
        return self.project_onto_constraint(X, inplace=True)

    def project_onto_tangent_space(self, X, Z, inplace=True):
        # Z - (1/n) X (Z^T X)   [ref: constraints.py:186-192; note Z^T X, not X^T Z]
        W, device = _prepare(Z, inplace)
        if X.dtype != torch.float32 or not X.is_contiguous() or X.device != W.device:
            X = X.detach().to(device=W.device, dtype=torch.float32).contiguous()
        lib = _lib.load()
        n, d = W.shape
        with torch.no_grad(), torch.cuda.device(device):
            _lib.check(lib.mde_std_tangent(n, d, _lib.ptr(X.detach()), _lib.ptr(W),
                                           _lib.ptr(util.work_buffer(device, d)),
                                           _lib.stream_ptr(device)))
        return W

    def project_onto_constraint(self, Z, inplace=True):
        return util.proj_standardized(Z, demean=True, inplace=inplace)

    def natural_length(self, n_items, embedding_dim):
        n, m = float(n_items), float(embedding_dim)
        return torch.tensor(2.0 * n * m / (n - 1.0)).sqrt()


class _Sphere(Constraint):
    """Rows on a sphere of the given radius (private in the reference, unused by the recipes
    [ref: constraints.py:203-231]).  Both projections are one row-wise HIP kernel (``mde_sphere_rows``); the
    solver reaches them through its generic (callback) path."""

    def __init__(self, radius):
        self.radius = radius
        super(_Sphere, self).__init__()

    def name(self):
        return "sphere"

    def _rows(self, W, device, X):
        lib = _lib.load()
        n, d = W.shape
        with torch.no_grad(), torch.cuda.device(device):
            _lib.check(lib.mde_sphere_rows(n, d, _lib.ptr(X) if X is not None else None, _lib.ptr(W),
                                           float(self.radius), _lib.stream_ptr(device)))
        return W

    def initialization(self, n_items, embedding_dim, device=None):
        X = _randn(n_items, embedding_dim, device)
        return self.project_onto_constraint(X, inplace=True)

    def project_onto_tangent_space(self, X, Z, inplace=True):
        # Z - (1/radius) diag(Z X^T) X   [ref: constraints.py:214-223]
        W, device = _prepare(Z, inplace)
        if X.dtype != torch.float32 or not X.is_contiguous() or X.device != W.device:
            X = X.detach().to(device=W.device, dtype=torch.float32).contiguous()
        return self._rows(W, device, X.detach())

    def project_onto_constraint(self, Z, inplace=True):
        # radius Z / |Z| row by row   [ref: constraints.py:225-231]
        W, device = _prepare(Z, inplace)
        return self._rows(W, device, None)


__Centered = _Centered()
__Standardized = _Standardized()


def Centered():
    """Centering constraint (the default): embedding vectors have mean zero."""
    return __Centered


def Standardized():
    """Standardization constraint: mean zero and (1/n) X^T X = I."""
    return __Standardized
