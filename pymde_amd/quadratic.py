"""Spectral (Laplacian-eigenmap) initialisation on the GPU.

``spectral(n_items, embedding_dim, edges, weights, ...)`` solves the quadratic MDE problem
    minimise sum_k w_k ||x_i - x_j||^2   s.t. (1/n) X^T X = I, X^T 1 = 0
i.e. the bottom non-trivial eigenvectors of the graph Laplacian L = D - A
[ref: pymde/quadratic.py:47-179].  The reference calls ARPACK (``eigsh(which='SM')``) on the
CPU or ``torch.lobpcg`` on CUDA; here a block LOBPCG runs on the device with
  * L V applied by the SAME fused edge kernel as the solve: for the Quadratic penalty
    dE/dV = (2/p) L V, so L V = (p/2) grad (``mde_average_distortion`` with grad_scale = p/2),
  * every Rayleigh-Ritz Gram matrix ([X R P]^T [X R P] and [X R P]^T L [X R P]) formed by
    ``mde_gram`` (f32 MFMA tiles when the block width is a multiple of 32),
  * the (<= 3(d+1))-sized dense eigenproblem solved on the host in float64.
The constant vector is deflated by centring the block, so only non-trivial eigenvectors are
iterated.  The result is centred and projected onto the standardization constraint, as in
quadratic.py:173-179.
"""
import numpy as np
import scipy.linalg
import torch

from pymde_amd import _lib
from pymde_amd import average_distortion as _ad
from pymde_amd import util
from pymde_amd.functions import penalties


class _Laplacian(object):
    """y = L V through the fused edge kernel."""

    def __init__(self, n, edges, weights, device):
        self.n = int(n)
        self.device = device
        edges = torch.as_tensor(edges).to(device=device, dtype=torch.int64).contiguous()
        weights = torch.as_tensor(weights).to(device=device, dtype=torch.float32).contiguous()
        self.plan = _ad.EdgePlan(self.n, edges)
        self.binding = _ad.Binding(self.plan, penalties.Quadratic(weights))
        self.p = int(edges.shape[0])
        self.loss = torch.empty(1, dtype=torch.float32, device=device)
        deg = torch.zeros(self.n, dtype=torch.float32, device=device)
        deg.index_add_(0, edges[:, 0], weights)
        deg.index_add_(0, edges[:, 1], weights)
        self.inv_degree = 1.0 / torch.clamp(deg, min=1e-12)

    def apply(self, V):
        out = torch.empty_like(V)
        _ad.fused_evaluate(self.binding, V.contiguous(), out, self.loss, grad_scale=0.5 * self.p)
        return out


def _gram(A, B, work):
    lib = _lib.load()
    n, da = A.shape
    db = B.shape[1]
    out = torch.empty((da, db), dtype=torch.float64, device=A.device)
    with torch.cuda.device(A.device):
        _lib.check(lib.mde_gram(n, da, db, _lib.ptr(A), _lib.ptr(B), _lib.ptr(out), _lib.ptr(work),
                                _lib.stream_ptr(A.device)))
    return out


def _rmul(A, M):
    """A @ M with M a small host float64 matrix."""
    lib = _lib.load()
    n, d = A.shape
    Md = torch.as_tensor(np.ascontiguousarray(M), dtype=torch.float64, device=A.device)
    out = torch.empty((n, Md.shape[1]), dtype=torch.float32, device=A.device)
    with torch.cuda.device(A.device):
        _lib.check(lib.mde_right_multiply(n, d, Md.shape[1], _lib.ptr(A), _lib.ptr(Md), _lib.ptr(out),
                                          _lib.stream_ptr(A.device)))
    return out


def _orthonormalise(V, work):
    """Return V C^{-1/2}-like orthonormal basis via Cholesky of the Gram matrix (host, tiny)."""
    G = _gram(V, V, work).cpu().numpy()
    G = 0.5 * (G + G.T)
    w, Q = np.linalg.eigh(G)
    keep = w > max(w.max(), 1e-300) * 1e-10
    M = Q[:, keep] / np.sqrt(w[keep])
    return _rmul(V, M)


def _lobpcg(lap, k, max_iter, tol, device, seed_block=None):
    n = lap.n
    work = util.work_buffer(device, max(3 * k, 4))
    X = seed_block if seed_block is not None else torch.randn((n, k), device=device,
                                                              dtype=torch.float32)
    X = X - X.mean(dim=0, keepdim=True)
    X = _orthonormalise(X, work)
    P = None
    theta = None
    for it in range(max_iter):
        LX = lap.apply(X)
        if theta is None:
            A = _gram(X, LX, work).cpu().numpy()
            theta, C = np.linalg.eigh(0.5 * (A + A.T))
            X = _rmul(X, C)
            LX = _rmul(LX, C)
        R = LX - X * torch.as_tensor(theta, dtype=torch.float32, device=device)[None, :]
        rnorm = R.norm(dim=0).cpu().numpy()
        if np.all(rnorm <= tol * np.maximum(np.abs(theta), 1e-12) + 1e-30):
            break
        W = R * lap.inv_degree[:, None]           # Jacobi preconditioner
        W = W - W.mean(dim=0, keepdim=True)       # stay orthogonal to the constant vector
        blocks = [X, W] if P is None else [X, W, P]
        S = _orthonormalise(torch.cat(blocks, dim=1), work)
        LS = lap.apply(S)
        A = _gram(S, LS, work).cpu().numpy()
        evals, evecs = scipy.linalg.eigh(0.5 * (A + A.T))
        C = evecs[:, :k]
        X_new = _rmul(S, C)
        # implicit P: the part of the new iterate outside span(X)
        P = X_new - X @ (X.T @ X_new)
        X = X_new
        theta = evals[:k]
        # re-orthonormalise X against rounding drift
        X = X - X.mean(dim=0, keepdim=True)
        X = _orthonormalise(X, work)
        if X.shape[1] < k:
            raise util.SolverError("spectral: the iteration block lost rank")
    return theta, X


def spectral(n_items, embedding_dim, edges, weights, cg=False, max_iter=40, device=None):
    """Spectral embedding: the ``embedding_dim`` bottom non-trivial Laplacian eigenvectors,
    centred and standardized.

    ``cg`` is accepted for signature compatibility (quadratic.py:122-124): both settings run the
    device LOBPCG; ``cg=False`` iterates to a tight tolerance (the reference's Lanczos branch),
    ``cg=True`` stops after ``max_iter`` iterations at most.
    """
    if device is None:
        device = edges.device if isinstance(edges, torch.Tensor) and edges.is_cuda else \
            util.get_default_device()
    device = util.require_cuda_device(device)
    n, m = int(n_items), int(embedding_dim)
    lap = _Laplacian(n, edges, weights, device)
    iters = max(int(max_iter), 1) if cg else max(5 * n, 200)
    tol = 1e-3 if cg else 1e-5
    with torch.no_grad(), torch.cuda.device(device):
        _, V = _lobpcg(lap, m, iters, tol, device)
        V = V - V.mean(dim=0, keepdim=True)
        return util.proj_standardized(V.contiguous(), demean=False)
