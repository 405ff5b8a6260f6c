"""Distortion-function base class.

Mirrors the protocol of the reference's ``pymde.functions.function.Function``
[ref: pymde/functions/function.py:9-30]: a ``torch.nn.Module`` whose tensor attributes are
registered as buffers, callable as ``f(distances) -> distortions``.  In this package a
built-in function is additionally a *descriptor* of a closed form that the HIP kernels
evaluate in registers (``_hip_spec``); its ``forward`` runs the same device code
element-wise (``mde_distortions``), so there is exactly one implementation of every formula.
"""
import ctypes

import torch

from pymde_amd import _lib

# function kinds of include/mde_hip.h
KIND = dict(
    NONE=0, LINEAR=1, QUADRATIC=2, CUBIC=3, POWER=4, HUBER=5, LOGISTIC=6, SIGMOID=7, HINGE=8,
    LOG1P=9, LOG=10, INVPOWER=11, LOGRATIO=12, DEADZONE_QUADRATIC=13, DEADZONE_CUBIC=14,
    CLIPPED_QUADRATIC=15, L_QUADRATIC=32, L_WEIGHTED_QUADRATIC=33, L_HUBER=34, L_CUBIC=35,
    L_POWER=36, L_WEIGHTED_POWER=37, L_ABSOLUTE=38, L_LOGISTIC=39, L_FRACTIONAL=40,
    L_SOFT_FRACTIONAL=41, L_CLIPPED_QUADRATIC=42, L_LOG1P=43)


class HipSpec(object):
    """What the kernels need to evaluate a built-in function: kind(s), per-edge arrays in the
    caller's EDGE order, scalars."""

    def __init__(self, kind, a0, a1=None, scalars=(0.0, 0.0, 0.0), kind_neg=0,
                 scalars_neg=(0.0, 0.0, 0.0)):
        self.kind = int(kind)
        self.kind_neg = int(kind_neg)
        self.a0 = a0
        self.a1 = a1
        self.scalars = tuple(float(s) for s in scalars)
        self.scalars_neg = tuple(float(s) for s in scalars_neg)

    def arrays(self):
        return [a for a in (self.a0, self.a1) if a is not None]

    def to_struct(self, a0, a1):
        """Fill a ``struct mde_func`` with the given device arrays (edge or plan order)."""
        f = _lib.MdeFunc()
        f.kind, f.kind_neg = self.kind, self.kind_neg
        f.a0 = a0.data_ptr()
        f.a0_scalar = 1 if a0.numel() == 1 else 0
        if a1 is not None:
            f.a1 = a1.data_ptr()
            f.a1_scalar = 1 if a1.numel() == 1 else 0
        else:
            f.a1 = None
            f.a1_scalar = 0
        f.s0, f.s1, f.s2 = self.scalars
        f.n0, f.n1, f.n2 = self.scalars_neg
        return f


def read_scalars(module, names):
    """The values of the scalar attributes ``names`` of a function object as Python floats.

    A scalar that is a tensor on the GPU (the reference registers ``exponent`` / ``threshold`` as buffers,
    function.py:22-26, so ``mde.to(device)`` moves them) costs a device-to-host copy AND a stream
    synchronisation per ``.item()``; the fused path asks for the scalars on every evaluation (has a parameter
    changed?), so the value is remembered per attribute and read again only when the attribute holds another
    tensor, another storage, or the tensor's version counter says it was written in place.  A write that bypasses
    the version counter (``f.exponent.data.fill_(3)``) is NOT seen -- the reference re-reads the tensor on every
    call --: ``invalidate_scalars(f)`` drops the remembered values, and ``MDE.embed`` calls it once per solve."""
    cache = module.__dict__.setdefault("_scalar_values", {})
    out = []
    for name in names:
        x = getattr(module, name)
        if not isinstance(x, torch.Tensor):
            out.append(float(x))
            continue
        if not x.is_cuda:
            out.append(float(x.item()))
            continue
        hit = cache.get(name)
        if hit is None or hit[0] is not x or hit[1] != x._version or hit[2] != x.data_ptr():
            hit = (x, x._version, x.data_ptr(), float(x.item()))
            cache[name] = hit
        out.append(hit[3])
    return out


def invalidate_scalars(module):
    """Forget the scalar values ``read_scalars`` remembers for ``module`` and its sub-modules (after a write that
    bypassed the tensors' version counters, e.g. through ``.data``)."""
    seen = [module] + (list(module.modules()) if isinstance(module, torch.nn.Module) else [])
    for m in seen:
        getattr(m, "__dict__", {}).pop("_scalar_values", None)


def _as_param(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous().reshape(-1)


class _Elementwise(torch.autograd.Function):
    """f(distances) element-wise on the device, differentiable w.r.t. distances."""

    @staticmethod
    def forward(ctx, distances, spec):
        _lib.require_gpu()
        if not distances.is_cuda:
            raise RuntimeError(
                "pymde_amd distortion functions evaluate on the GPU only; got a %s tensor"
                % distances.device)
        lib = _lib.load()
        d = distances.detach().to(torch.float32).contiguous().reshape(-1)
        p = d.numel()
        a0 = _as_param(spec.a0, d.device)
        a1 = _as_param(spec.a1, d.device) if spec.a1 is not None else None
        for a in (a0, a1):
            if a is not None and a.numel() not in (1, p):
                raise ValueError(
                    "distortion function has %d parameters but received %d distances"
                    % (a.numel(), p))
        out = torch.empty_like(d)
        need_grad = distances.requires_grad
        dout = torch.empty_like(d) if need_grad else None
        f = spec.to_struct(a0, a1)
        with torch.cuda.device(d.device):
            _lib.check(lib.mde_distortions(p, _lib.ptr(d), ctypes.byref(f), _lib.ptr(out),
                                           _lib.ptr(dout), _lib.stream_ptr(d.device)))
        if need_grad:
            ctx.save_for_backward(dout)
        ctx.shape = distances.shape
        return out.reshape(distances.shape)

    @staticmethod
    def backward(ctx, grad_output):
        (dout,) = ctx.saved_tensors
        return grad_output * dout.reshape(ctx.shape), None


class Function(torch.nn.Module):
    """Distortion function: maps a vector of embedding distances to a vector of distortions."""

    def __init__(self):
        super(Function, self).__init__()

    def __setattr__(self, name, value):
        if isinstance(value, torch.Tensor) and not isinstance(value, torch.nn.Parameter):
            self.register_buffer(name, value)
        else:
            super(Function, self).__setattr__(name, value)

    @property
    def device(self):
        bufs = list(self.buffers())
        if not bufs:
            return None
        dev = str(bufs[0].device)
        return dev if all(str(b.device) == dev for b in bufs) else None

    def _hip_spec(self):
        """Return a HipSpec, or None when the function has no closed form in the kernels
        (the MDE then takes the unfused path)."""
        return None

    def forward(self, distances):
        spec = self._hip_spec()
        if spec is None:
            raise NotImplementedError
        return _Elementwise.apply(distances, spec)
