"""Multi-GPU evaluation: one process per GPU, vertex-range ownership, one RCCL all-reduce.

The objective is a sum over edges, so it shards with ONE exchange step per evaluation
(SURVEY section 8e).  Every rank keeps a full replica of X and builds an edge plan for the
vertex range [lo, hi) it owns (balanced by half-edge count, ``mde_shard_bounds``); the fused
kernel writes only rows [lo, hi) of the gradient -- each already final, owner-computes, no
atomics -- plus the rank's share of the loss into one buffer ``[grad | loss]`` that is zero
elsewhere.  A single ``all_reduce(SUM)`` of that n*d+1 float buffer (RCCL over xGMI;
``backend='nccl'`` on ROCm) leaves every rank with the identical full gradient and loss;
because every element has exactly one non-zero contribution the result is bitwise identical
to the single-GPU one.  The optimiser then runs replicated (no further communication).

``ShardedMDE`` is an ``MDE`` whose plan covers this rank's range and whose evaluations go
through the reducer; ``MDE.embed`` works unchanged on it.
"""
import ctypes

import torch
import torch.distributed as dist

from pymde_amd import _lib
from pymde_amd import average_distortion as _ad
from pymde_amd import problem


def all_reduce_grad_loss(buf, group=None):
    """Sum the ``[grad | loss]`` buffers of all ranks in place (the only data-path collective)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


def shard_range(bounds, rank):
    """Vertex range [lo, hi) of ``rank`` given the world+1 boundaries."""
    return int(bounds[rank]), int(bounds[rank + 1])


def shard_bounds(n_items, edges, world_size):
    """Balanced vertex-range boundaries computed on the device (identical on every rank)."""
    lib = _lib.load()
    edges = edges.contiguous()
    out = (ctypes.c_int64 * (world_size + 1))()
    with torch.cuda.device(edges.device):
        _lib.check(lib.mde_shard_bounds(int(n_items), int(edges.shape[0]), _lib.ptr(edges),
                                        int(world_size), out, _lib.stream_ptr(edges.device)))
    return [int(v) for v in out]


class ShardedMDE(problem.MDE):
    """An MDE problem whose edges are sharded across the ranks of a process group."""

    def __init__(self, n_items, embedding_dim, edges, distortion_function, constraint=None,
                 device=None, group=None, rank=None, world_size=None):
        self._group = group
        self._rank = dist.get_rank(group) if rank is None else int(rank)
        self._world = dist.get_world_size(group) if world_size is None else int(world_size)
        self._bounds = None
        super(ShardedMDE, self).__init__(n_items, embedding_dim, edges, distortion_function,
                                         constraint=constraint, device=device)
        self._reducer = lambda buf: all_reduce_grad_loss(buf, self._group)

    def _make_plan(self, edges):
        self._bounds = shard_bounds(self._n, edges, self._world)
        lo, hi = shard_range(self._bounds, self._rank)
        return _ad.EdgePlan(self._n, edges, lo, hi)

    def average_distortion(self, X=None):
        """E(X) with gradient, computed from this rank's shard and all-reduced."""
        X = self._embedding_arg(X)
        return _ShardedAverageDistortion.apply(X, self._binding(), self._reducer)


class _ShardedAverageDistortion(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, binding, reducer):
        if not binding.fused:
            raise NotImplementedError("sharded evaluation needs a built-in distortion function")
        Xc = X.detach().contiguous()
        n, d = Xc.shape
        buf = torch.zeros(n * d + 1, dtype=torch.float32, device=X.device)
        grad = buf[:n * d].view(n, d) if X.requires_grad else None
        _ad.fused_evaluate(binding, Xc, grad, buf[n * d:])
        reducer(buf)
        if X.requires_grad:
            ctx.save_for_backward(buf[:n * d].view(n, d))
        return buf[n * d].clone()

    @staticmethod
    def backward(ctx, grad_output):
        (grad,) = ctx.saved_tensors
        return grad * grad_output, None, None
