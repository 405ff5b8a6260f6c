"""Spectral (Laplacian-eigenmap) initialisation on the GPU.

``spectral(n_items, embedding_dim, edges, weights, ...)`` solves the quadratic MDE problem
    minimise sum_k w_k ||x_i - x_j||^2   s.t. (1/n) X^T X = I, X^T 1 = 0
i.e. the bottom non-trivial eigenvectors of the graph Laplacian L = D - A
[ref: pymde/quadratic.py:47-179].  The reference calls ARPACK (``eigsh(which='SM')``) on the
CPU or ``torch.lobpcg`` on CUDA; here a block LOBPCG runs on the device, built only from the
library's kernels:
  * L V is applied by the SAME fused edge kernel as the solve: for the Quadratic penalty
    dE/dV = (2/p) L V, so L V = (p/2) grad (``mde_average_distortion`` with grad_scale = p/2);
  * every Rayleigh-Ritz Gram block ([X W P]^T [X W P] and [X W P]^T L [X W P]) is formed by
    ``mde_gram`` (f32 MFMA tiles when the block widths are multiples of 32);
  * block updates are ``mde_right_multiply_add`` (n x k times k x k), the Jacobi preconditioner
    is ``mde_row_scale`` with the Laplacian diagonal from ``mde_weighted_degree``, centring is
    ``mde_center``;
  * only the (<= 3 d)-sized dense eigenproblems are solved on the host, in float64.
The constant vector is deflated by centring every block, so only non-trivial eigenvectors are
iterated.  The result is centred and projected onto the standardization constraint, as in
quadratic.py:173-179.  torch is used for the random start block and buffer allocation only.
"""
import numpy as np
import scipy.linalg
import torch

from pymde_amd import _lib
from pymde_amd import average_distortion as _ad
from pymde_amd import util
from pymde_amd.functions import penalties


class _Ops(object):
    """Thin wrappers over the C ABI for [n, k] float32 blocks on one device.

    Small matrices cross the PCIe bus through pinned staging buffers: a Gram matrix costs one
    asynchronous copy + one stream synchronisation, a right-multiplication none (its matrix is
    staged in a ring of pinned buffers; a ring entry is reused only after a later
    synchronisation has retired the copy that read it)."""

    _RING = 16

    def __init__(self, n, device, kmax):
        self.lib = _lib.load()
        self.n = int(n)
        self.device = device
        self.work = util.work_buffer(device, max(kmax, 4))
        self._stream_obj = torch.cuda.current_stream(device)
        self._stream_ptr = _lib.stream_ptr(device)
        self._host_out = torch.empty(max(kmax, 4) ** 2, dtype=torch.float64).pin_memory()
        self._ring_cap = min(max(kmax, 4), 64) ** 2   # larger matrices take the synchronous path
        self._ring = [torch.empty(self._ring_cap, dtype=torch.float64).pin_memory()
                      for _ in range(self._RING)]
        self._ring_next = 0      # next ring entry to use
        self._ring_used = 0      # entries handed out since the last synchronisation

    def _stream(self):
        return self._stream_ptr

    def gram(self, A, B):
        da, db = A.shape[1], B.shape[1]
        out = torch.empty(da * db, dtype=torch.float64, device=self.device)
        _lib.check(self.lib.mde_gram(self.n, da, db, _lib.ptr(A), _lib.ptr(B),
                                     _lib.ptr(out), _lib.ptr(self.work), self._stream()))
        host = self._host_out[:da * db]
        host.copy_(out, non_blocking=True)
        self._stream_obj.synchronize()
        self._ring_used = 0
        return host.numpy().reshape(da, db).copy()

    def _upload(self, M):
        """Small host float64 matrix -> device, without waiting for the copy."""
        if M.size > self._ring_cap:
            Md = torch.from_numpy(M).to(self.device)
            self._stream_obj.synchronize()   # the pageable source must outlive the copy
            return Md.reshape(-1)
        if self._ring_used >= self._RING:
            self._stream_obj.synchronize()
            self._ring_used = 0
        buf = self._ring[self._ring_next][:M.size]
        self._ring_next = (self._ring_next + 1) % self._RING
        self._ring_used += 1
        buf.numpy()[:] = M.reshape(-1)
        Md = torch.empty(M.size, dtype=torch.float64, device=self.device)
        Md.copy_(buf, non_blocking=True)
        return Md

    def rmul(self, A, M, alpha=1.0, base=None, out=None):
        """out = base + alpha * A @ M  (M: small host float64 matrix)."""
        M = np.ascontiguousarray(M, dtype=np.float64)
        Md = self._upload(M)
        if out is None:
            out = torch.empty((self.n, M.shape[1]), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mde_right_multiply_add(self.n, A.shape[1], M.shape[1], _lib.ptr(A),
                                                   _lib.ptr(Md), float(alpha), _lib.ptr(base),
                                                   _lib.ptr(out), self._stream()))
        return out

    def center(self, Z):
        _lib.check(self.lib.mde_center(self.n, Z.shape[1], _lib.ptr(Z), _lib.ptr(self.work),
                                       self._stream()))
        return Z

    def row_scale(self, Z, scale):
        _lib.check(self.lib.mde_row_scale(self.n, Z.shape[1], _lib.ptr(scale), _lib.ptr(Z),
                                          self._stream()))
        return Z


class _Laplacian(object):
    """y = L V through the fused edge kernel; diag(L) through mde_weighted_degree."""

    def __init__(self, n, edges, weights, device):
        lib = _lib.load()
        self.n = int(n)
        self.device = device
        edges = torch.as_tensor(edges).to(device=device, dtype=torch.int64).contiguous()
        weights = torch.as_tensor(weights).to(device=device, dtype=torch.float32).contiguous()
        self.plan = _ad.EdgePlan(self.n, edges)
        self.binding = _ad.Binding(self.plan, penalties.Quadratic(weights))
        self.p = int(edges.shape[0])
        self.loss = torch.empty(1, dtype=torch.float32, device=device)
        w_csr = self.plan.expand(weights, 0)
        deg = torch.zeros(self.n, dtype=torch.float32, device=device)
        _lib.check(lib.mde_weighted_degree(self.plan.handle, _lib.ptr(w_csr), _lib.ptr(deg),
                                           _lib.stream_ptr(device)))
        # Jacobi preconditioner 1 / L_vv (isolated vertices: 1)
        self.inv_degree = torch.where(deg > 0, 1.0 / deg, torch.ones_like(deg)).contiguous()

    def apply(self, V):
        out = torch.empty_like(V)
        _ad.fused_evaluate(self.binding, V, out, self.loss, grad_scale=0.5 * self.p)
        return out


def _orthonormalizer(G, rel_tol=1e-9):
    """Host: M with (S M)^T (S M) = I for the Gram matrix G = S^T S (drops null directions)."""
    G = 0.5 * (G + G.T)
    w, Q = np.linalg.eigh(G)
    keep = w > max(float(w.max()), 1e-300) * rel_tol
    return Q[:, keep] / np.sqrt(w[keep])


def _sym(A):
    return 0.5 * (A + A.T)


def _lobpcg(lap, k, max_iter, tol, device, scale_by_operator=False):
    """Block LOBPCG for the k smallest non-trivial eigenpairs of the Laplacian.

    Convergence: ``||r_i|| <= tol |theta_i|`` per Ritz pair, or, with ``scale_by_operator``, the
    test of ``torch.lobpcg`` that the reference's GPU branch runs (quadratic.py:109-116):
    ``||r_i|| <= tol (||L X_0||_F + |theta_i| ||X_0||_F)`` -- residuals measured against the scale
    of the operator rather than against the (small) eigenvalue itself.

    Every iteration builds an explicitly orthonormal basis Q of span[X, W, P] and applies L to
    it afresh, so the Rayleigh-Ritz matrices always belong to the vectors actually held (no
    drift from recombining stored L-images in fp32)."""
    n = lap.n
    ops = _Ops(n, device, 3 * k)

    def orthonormal(blocks):
        """Orthonormal basis of the span of the given [n, *] blocks (explicit vectors): one Gram
        matrix of the concatenated block, one right-multiplication, then one re-orthonormalisation
        pass that removes the fp32 error of the first."""
        S = blocks[0] if len(blocks) == 1 else torch.cat(blocks, dim=1).contiguous()
        out = ops.rmul(S, _orthonormalizer(_sym(ops.gram(S, S))))
        return ops.rmul(out, _orthonormalizer(ops.gram(out, out)))

    X = orthonormal([ops.center(torch.randn((n, k), device=device, dtype=torch.float32))])
    if X.shape[1] < k:
        raise util.SolverError("spectral: the random start block is rank deficient")
    P = None
    theta = None
    op_norm = None
    for _ in range(max_iter):
        LX = lap.apply(X)
        if scale_by_operator and op_norm is None:
            op_norm = float(np.sqrt(max(np.trace(ops.gram(LX, LX)), 0.0)))
        A = _sym(ops.gram(X, LX))
        theta, C = np.linalg.eigh(A)
        X, LX = ops.rmul(X, C), ops.rmul(LX, C)       # Ritz vectors of the current block
        R = ops.rmul(X, np.diag(theta), alpha=-1.0, base=LX)   # residual L X - X diag(theta)
        rnorm = np.sqrt(np.maximum(np.diag(ops.gram(R, R)), 0.0))
        if scale_by_operator:
            bound = tol * (op_norm + np.abs(theta) * np.sqrt(float(k)))
        else:
            bound = tol * np.maximum(np.abs(theta), 1e-12)
        if np.all(rnorm <= bound):
            break
        W = ops.center(ops.row_scale(R, lap.inv_degree))  # Jacobi-preconditioned, orthogonal to 1
        Q = orthonormal([X, W] if P is None else [X, W, P])
        LQ = lap.apply(Q)
        evals, evecs = scipy.linalg.eigh(_sym(ops.gram(Q, LQ)))
        X_new = ops.rmul(Q, evecs[:, :k])
        # conjugate direction: the part of the new iterate outside the old one
        P = ops.rmul(X, ops.gram(X, X_new), alpha=-1.0, base=X_new)
        X = orthonormal([X_new])
        if X.shape[1] < k:
            raise util.SolverError("spectral: the iteration block lost rank")
        theta = evals[:k]
    return theta, X


def spectral(n_items, embedding_dim, edges, weights, cg=False, max_iter=40, device=None):
    """Spectral embedding: the ``embedding_dim`` bottom non-trivial Laplacian eigenvectors,
    centred and standardized.

    Both settings of ``cg`` run the device LOBPCG.  ``cg=False`` iterates to a tight tolerance
    relative to the eigenvalues (the accuracy of the reference's Lanczos branch,
    quadratic.py:84-92); ``cg=True`` is the reference's GPU branch (quadratic.py:109-116): at most
    ``max_iter`` iterations with ``torch.lobpcg``'s default test, residuals below sqrt(float32 eps)
    relative to the scale of the operator.
    """
    if device is None:
        device = edges.device if isinstance(edges, torch.Tensor) and edges.is_cuda else \
            util.get_default_device()
    device = util.require_cuda_device(device)
    n, m = int(n_items), int(embedding_dim)
    iters = max(int(max_iter), 1) if cg else max(5 * n, 200)
    tol = float(np.sqrt(np.finfo(np.float32).eps)) if cg else 1e-5
    with torch.no_grad(), torch.cuda.device(device):
        lap = _Laplacian(n, edges, weights, device)
        _, V = _lobpcg(lap, m, iters, tol, device, scale_by_operator=bool(cg))
        _Ops(n, device, m).center(V)
        return util.proj_standardized(V.contiguous(), demean=False)


def pca(Y, embedding_dim, device=None):
    """PCA embedding of a data matrix: the top ``embedding_dim`` left singular vectors of the
    column-centred ``Y``, scaled by sqrt(n) [ref: pymde/quadratic.py:16-44].

    The reference takes a full SVD of ``Y`` on the CPU.  Here the k x k Gram matrix of the centred
    data is formed on the GPU (``mde_center`` + ``mde_gram``), its eigendecomposition (k x k,
    float64) is done on the host, and ``U = Y V S^-1`` is one ``mde_right_multiply``.  Singular
    vectors are defined up to sign; each column is oriented so that its largest-magnitude entry
    of the right singular vector is positive."""
    if not isinstance(Y, torch.Tensor):
        Y = torch.as_tensor(Y)
    if device is None:
        device = Y.device if Y.is_cuda else util.get_default_device()
    device = util.require_cuda_device(device)
    n, k = int(Y.shape[0]), int(Y.shape[1])
    m = int(embedding_dim)
    if m > min(n, k):
        raise ValueError("Embedding dimension must be at most minimum dimension of Y")
    with torch.no_grad(), torch.cuda.device(device):
        Yc = Y.detach().to(device=device, dtype=torch.float32).contiguous().clone()
        ops = _Ops(n, device, k)
        ops.center(Yc)
        G = _sym(ops.gram(Yc, Yc))
        evals, evecs = np.linalg.eigh(G)
        order = np.argsort(evals)[::-1][:m]
        sing = np.sqrt(np.maximum(evals[order], 0.0))
        if np.any(sing <= 1e-12 * max(float(sing.max()), 1e-300)):
            raise util.SolverError("pca: the centred data matrix has rank below embedding_dim")
        V = evecs[:, order]
        V = V * np.sign(V[np.abs(V).argmax(axis=0), np.arange(m)])[None, :]
        return ops.rmul(Yc, V * (np.sqrt(float(n)) / sing)[None, :])
