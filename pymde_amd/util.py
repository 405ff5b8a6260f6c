"""Small host-side utilities shared by the package.

Only what the hot path needs from the reference's ``pymde/util.py`` is provided:
``to_tensor`` [ref: util.py:59-79], device helpers [ref: util.py:20-48], ``SolverError``
[ref: util.py:16], ``seed``/``np_rng`` [ref: util.py:394-408], ``all_edges`` [ref: util.py:101-112],
``natural_length`` [ref: util.py:115-117] and ``proj_standardized`` [ref: util.py:129-171],
the last one running on the HIP kernels (Gram + d x d inverse square root) instead of a
thin SVD.
"""
import numbers

import numpy as np
import torch

from pymde_amd import _lib

_DEVICE = None
_NP_RNG = np.random.default_rng()


class SolverError(Exception):
    pass


def _canonical_device(device):
    if isinstance(device, str):
        device = torch.device(device)
    elif not isinstance(device, torch.device):
        raise ValueError("device must be a str or a torch.device object.")
    if device.type == "cuda" and device.index is None:
        idx = torch.cuda.current_device() if torch.cuda.is_available() else 0
        device = torch.device("cuda", idx)
    return device


def get_default_device():
    """The device MDE problems are placed on when none is given: the current GPU."""
    if _DEVICE is not None:
        return str(_DEVICE)
    return str(_canonical_device("cuda"))


def set_default_device(device):
    global _DEVICE
    _DEVICE = _canonical_device(device)


def require_cuda_device(device):
    """This package has no CPU path: fail loudly when asked for one."""
    device = _canonical_device(device)
    if device.type != "cuda":
        raise RuntimeError(
            "pymde_amd is MI355X-native: device must be a 'cuda' (ROCm) device, got %r. "
            "There is no CPU path; use the reference pymde for CPU runs." % str(device))
    _lib.require_gpu()
    return device


def _is_numeric(arg):
    return isinstance(arg, (numbers.Number, np.ndarray, np.generic, torch.Tensor))


def to_tensor(args, device=None):
    """Convert a number / array (or a list of them) to torch tensors; float64 arrays become
    float32 (the kernels compute in fp32)."""
    single = not isinstance(args, (list, tuple))
    items = [args] if single else list(args)
    out = []
    for a in items:
        if isinstance(a, torch.Tensor):
            out.append(a if device is None else a.to(device))
        elif _is_numeric(a):
            if isinstance(a, np.ndarray) and a.dtype == np.float64:
                out.append(torch.tensor(a, dtype=torch.float32, device=device))
            else:
                out.append(torch.tensor(a, device=device))
        else:
            raise ValueError("Received non-numeric argument ", a)
    return out[0] if single else out


def all_edges(n):
    """All ``n choose 2`` edges (i, j), i < j, in row-major order."""
    return torch.triu_indices(n, n, 1).T


def natural_length(n, m):
    return torch.tensor(2.0 * float(n) * float(m) / (float(n) - 1.0)).sqrt()


def np_rng():
    return _NP_RNG


def seed(seed: int):
    """Seed torch, numpy's legacy global state and the package's private Generator."""
    global _NP_RNG
    torch.manual_seed(seed)
    np.random.seed(seed)
    _NP_RNG = np.random.default_rng(seed)


def center(X):
    """Return X with its column means removed."""
    from pymde_amd import constraints
    return constraints.Centered().project_onto_constraint(X, inplace=False)


_WORK = {}          # (device, stream handle) -> buffer, least recently used first
_WORK_KEEP = 8      # streams whose scratch is kept (34 MB each at d <= 4): a program that solves on many
                    # short-lived streams must not accumulate one buffer per stream it ever used


def work_buffer(device, d):
    """Scratch (doubles) for the reduction / Gram kernels, one buffer per (device, current stream):
    the kernels that finish their reduction in the last workgroup keep arrival counters in it, so
    two streams must never share one.  The buffers of the `_WORK_KEEP` most recently used streams are
    kept; an evicted one stays alive as long as a solver object still holds it."""
    lib = _lib.load()
    need = int(lib.mde_work_doubles(int(d)))
    key = (str(device), int(torch.cuda.current_stream(device).cuda_stream))
    buf = _WORK.pop(key, None)
    if buf is None or buf.numel() < need:
        # zeroed once: the small area holds the arrival counters of the kernels that finish their
        # reduction in the last workgroup (they return to zero after every launch).  The allocation happens
        # with the KEY stream current (torch.cuda.current_stream above is the stream the caller launches on),
        # which is what makes an eviction safe: the caching allocator hands a freed block to another stream
        # only after the work queued on the allocating stream has passed the free.
        buf = torch.zeros(need, dtype=torch.float64, device=device)
    _WORK[key] = buf  # (re-inserted: most recently used last)
    while len(_WORK) > _WORK_KEEP:
        old_key = next(iter(_WORK))
        old = _WORK.pop(old_key)
        # an evicted buffer may still be read by kernels in flight on ITS stream (callers pass the pointer of a
        # temporary to asynchronous launches): tell the allocator, whatever stream is current at the free
        # (the stream handle belongs to the device of the EVICTED buffer, old_key[0], not to this call's device;
        # ExternalStream cannot tell a destroyed handle from a live one, so a buffer of another device -- whose
        # stream this process may have dropped -- is simply released: its block returns to that device's allocator
        # and is reused on the allocating stream's order only)
        if old_key[0] == str(device):
            try:
                old.record_stream(torch.cuda.ExternalStream(old_key[1], device=old.device))
            except Exception:  # (a stream that no longer exists: nothing is in flight on it)
                pass
    return buf


def reset_work_tickets(buf):
    """Zero the arrival counters of a work buffer (the last 32 doubles of its 4096-double small
    area): a launch that faulted may have left one non-zero, after which no workgroup would ever see
    itself as the last one."""
    buf[4096 - 32:4096].zero_()


def proj_standardized(X, demean=False, inplace=False):
    """Project X onto {X : (1/n) X^T X = I} (and centre it first when ``demean``).

    Equals sqrt(n) U V^T of the thin SVD X = U S V^T [ref: util.py:129-171], computed on the
    GPU as sqrt(n) X C^{-1/2} with C = X^T X.
    """
    device = require_cuda_device(X.device)
    if X.dtype != torch.float32:
        raise ValueError("proj_standardized expects a float32 tensor")
    lib = _lib.load()
    Z = X if (inplace and X.is_contiguous()) else X.detach().clone().contiguous()
    n, d = Z.shape
    work = work_buffer(device, d)
    status = torch.zeros(1, dtype=torch.int32, device=device)
    with torch.no_grad(), torch.cuda.device(device):
        _lib.check(lib.mde_std_retract(n, d, _lib.ptr(Z), 1 if demean else 0, _lib.ptr(work),
                                       _lib.ptr(status), _lib.stream_ptr(device)))
    if int(status.item()) != 0:
        raise SolverError("proj_standardized: X^T X is numerically singular")
    if inplace and Z is not X:
        X.copy_(Z)
        return X
    return Z


# ---------------------------------------------------------------- aligning / rotating embeddings
# [ref: pymde/util.py:201-331].  The n-sized work (column sums, Gram matrices, X @ M) runs in the
# library's kernels (mde_gram / mde_center / mde_right_multiply / mde_shift_rows); only d x d
# matrices visit the host.
def _blocks(X):
    from pymde_amd import _lib
    X = X.detach()
    device = require_cuda_device(X.device)
    X = X.to(dtype=torch.float32).contiguous()
    return _lib, _lib.load(), device, X, int(X.shape[0]), int(X.shape[1])


def _gram(lib_mod, lib, A, B, work):
    out = torch.empty((A.shape[1], B.shape[1]), dtype=torch.float64, device=A.device)
    lib_mod.check(lib.mde_gram(A.shape[0], A.shape[1], B.shape[1], lib_mod.ptr(A), lib_mod.ptr(B),
                               lib_mod.ptr(out), lib_mod.ptr(work), lib_mod.stream_ptr(A.device)))
    return out.cpu().numpy()


def _right_multiply(lib_mod, lib, A, M):
    """A @ M for a small host float64 matrix M."""
    import numpy as np
    M = np.ascontiguousarray(M, dtype=np.float64)
    Md = torch.from_numpy(M).to(A.device)
    torch.cuda.current_stream(A.device).synchronize()
    out = torch.empty((A.shape[0], M.shape[1]), dtype=torch.float32, device=A.device)
    lib_mod.check(lib.mde_right_multiply(A.shape[0], A.shape[1], M.shape[1], lib_mod.ptr(A),
                                         lib_mod.ptr(Md), lib_mod.ptr(out),
                                         lib_mod.stream_ptr(A.device)))
    return out


def _centered_unit_columns(lib_mod, lib, X, work):
    """(X - mean) / column norm, the column means and the column norms."""
    import numpy as np
    n, d = X.shape
    ones = torch.ones((n, 1), dtype=torch.float32, device=X.device)
    mean = _gram(lib_mod, lib, ones, X, work)[0] / float(n)
    Xc = X.clone()
    lib_mod.check(lib.mde_center(n, d, lib_mod.ptr(Xc), lib_mod.ptr(work), lib_mod.stream_ptr(X.device)))
    norms = np.sqrt(np.maximum(np.diag(_gram(lib_mod, lib, Xc, Xc, work)), 0.0))
    return _right_multiply(lib_mod, lib, Xc, np.diag(1.0 / norms)), mean, norms


def procrustes(X_source, X_target):
    """argmin_Q ||X_source Q - X_target||_F over orthogonal Q [ref: util.py:201-205]; returns the
    d x d matrix as a float32 tensor on the source's device."""
    import numpy as np
    lib_mod, lib, device, S, n, d = _blocks(X_source)
    T = X_target.detach().to(device=device, dtype=torch.float32).contiguous()
    with torch.cuda.device(device):
        work = work_buffer(device, d)
        M = _gram(lib_mod, lib, T, S, work)  # X_target^T X_source
    U, _, Vh = np.linalg.svd(M, full_matrices=False)
    return torch.from_numpy((Vh.T @ U.T).astype(np.float32)).to(device)


def align(source, target):
    """Rotate / reflect ``source`` onto ``target`` (orthogonal Procrustes on the centred,
    column-normalised embeddings), then restore the source's column scales and mean
    [ref: util.py:302-331]."""
    import numpy as np
    lib_mod, lib, device, S, n, d = _blocks(source)
    T = target.detach().to(device=device, dtype=torch.float32).contiguous()
    with torch.cuda.device(device):
        work = work_buffer(device, d)
        Sn, mean, norms = _centered_unit_columns(lib_mod, lib, S, work)
        Tn, _, _ = _centered_unit_columns(lib_mod, lib, T, work)
        U, _, Vh = np.linalg.svd(_gram(lib_mod, lib, Tn, Sn, work), full_matrices=False)
        Q = Vh.T @ U.T
        out = _right_multiply(lib_mod, lib, Sn, Q * norms[None, :])
        shift = torch.from_numpy(np.ascontiguousarray(mean, dtype=np.float64)).to(device)
        torch.cuda.current_stream(device).synchronize()
        lib_mod.check(lib.mde_shift_rows(n, d, lib_mod.ptr(shift), lib_mod.ptr(out),
                                         lib_mod.stream_ptr(device)))
    return out


def rotate(X, degrees):
    """Rotate a 2-D embedding by ``degrees`` (scalar) or a 3-D one by three angles about the x, y
    and z axes in that order [ref: util.py:208-299]."""
    import numpy as np
    if X.shape[1] not in (2, 3):
        raise ValueError("Only 2 or 3 dimensional embeddings can be rotated using this method.")
    deg = np.atleast_1d(np.asarray(degrees.detach().cpu() if isinstance(degrees, torch.Tensor) else degrees,
                                   dtype=np.float64))
    if X.shape[1] == 2:
        if deg.size != 1:
            raise ValueError("`degrees` must be a scalar.")
        t = np.deg2rad(deg[0])
        R = np.array([[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]])
    else:
        if deg.size != 3:
            raise ValueError("`degrees` must be a length-3 tensor.")
        a, b, g = np.deg2rad(deg)
        rx = np.array([[1, 0, 0], [0, np.cos(a), np.sin(a)], [0, -np.sin(a), np.cos(a)]])
        ry = np.array([[np.cos(b), 0, -np.sin(b)], [0, 1, 0], [np.sin(b), 0, np.cos(b)]])
        rz = np.array([[np.cos(g), np.sin(g), 0], [-np.sin(g), np.cos(g), 0], [0, 0, 1]])
        R = rx @ ry @ rz
    lib_mod, lib, device, Xc, n, d = _blocks(X)
    with torch.cuda.device(device):
        return _right_multiply(lib_mod, lib, Xc, R)
