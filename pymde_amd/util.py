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


_WORK = {}


def work_buffer(device, d):
    """Per-device scratch (doubles) for the reduction / Gram kernels."""
    lib = _lib.load()
    need = int(lib.mde_work_doubles(int(d)))
    key = str(device)
    buf = _WORK.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.empty(need, dtype=torch.float64, device=device)
        _WORK[key] = buf
    return buf


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
