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


def all_reduce_grad_loss(buf, group=None, force=False):
    """Sum the ``[grad | loss]`` buffers of all ranks in place (the only data-path collective)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


class GradExchange(object):
    """The per-evaluation exchange of a sharded problem.

    Default: one ``all_reduce(SUM)`` of the zero-padded ``[grad | loss]`` buffer.  When every rank
    owns the same number of rows the exchange is an **all-gather** instead -- each rank contributes
    only the rows it owns (already final: owner-computes), i.e. half the bytes of a ring all-reduce
    and no reduction arithmetic -- plus an all-gather of the N loss shares.  The gather path is
    verified against the all-reduce on its first use (bitwise, gradient) and silently replaced by it
    if the backend cannot run it or the results differ."""

    def __init__(self, n, d, bounds, rank, world, group=None, force=False):
        self.n, self.d, self.rank, self.world, self.group = int(n), int(d), int(rank), int(world), group
        self.bounds = [int(b) for b in bounds]
        sizes = [self.bounds[r + 1] - self.bounds[r] for r in range(world)]
        # force: issue the collectives even in a world of one (a single-GPU box can then run the
        # RCCL calls themselves -- tests/test_gpu_rccl.py)
        self.force = bool(force)
        self.uniform = (world > 1 or self.force) and len(set(sizes)) == 1
        self.mode = None  # decided at the first exchange
        self._losses = None

    def _gather(self, buf):
        N = self.n * self.d
        lo, hi = self.bounds[self.rank] * self.d, self.bounds[self.rank + 1] * self.d
        if self._losses is None:
            self._losses = torch.empty(self.world, dtype=buf.dtype, device=buf.device)
        # in place: this rank's rows already sit at their offset of the gathered buffer (the
        # NCCL / RCCL in-place all-gather form), so no staging copy; rows owned by other ranks
        # are overwritten and need not be zeroed first
        h1 = dist.all_gather_into_tensor(buf[:N], buf[lo:hi], group=self.group, async_op=True)
        h2 = dist.all_gather_into_tensor(self._losses, buf[N:N + 1], group=self.group, async_op=True)
        h1.wait()
        h2.wait()
        # (one launch: the sum of the N loss shares straight into the buffer's loss slot)
        torch.sum(self._losses, dim=0, keepdim=True, out=buf[N:N + 1])
        return buf

    def needs_zero(self):
        """Whether rows of other ranks must be zero before the exchange (all-reduce only)."""
        return self.mode != "all_gather"

    def __call__(self, buf):
        if (self.world <= 1 and not self.force) or not (dist.is_available() and dist.is_initialized()):
            return buf
        if self.mode is None:
            self.mode = "all_reduce"
            if self.uniform:
                try:
                    ref = all_reduce_grad_loss(buf.clone(), self.group, self.force)
                    got = self._gather(buf.clone())
                    N = self.n * self.d
                    same = torch.equal(got[:N], ref[:N]) and bool(
                        (got[N] - ref[N]).abs() <= 1e-6 * ref[N].abs() + 1e-30)
                    flag = torch.tensor([1.0 if same else 0.0], device=buf.device)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
                    if float(flag.item()) == 1.0:
                        self.mode = "all_gather"
                except Exception:  # backend without all_gather_into_tensor, etc.
                    self.mode = "all_reduce"
        if self.mode == "all_gather":
            return self._gather(buf)
        return all_reduce_grad_loss(buf, self.group, self.force)


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
    bounds = [int(v) for v in out]
    # equal row counts let the exchange be an all-gather (GradExchange); prefer them when the
    # half-edge balance they give is within 3 % of the balanced split (e.g. random graphs)
    n = int(n_items)
    if world_size > 1 and n % world_size == 0:
        step = n // world_size
        uniform = [r * step for r in range(world_size + 1)]
        deg = torch.zeros(n, dtype=torch.int64, device=edges.device)
        ones = torch.ones(edges.shape[0], dtype=torch.int64, device=edges.device)
        deg.index_add_(0, edges[:, 0], ones)
        deg.index_add_(0, edges[:, 1], ones)
        per = deg.view(world_size, step).sum(1).double()
        if float(per.max()) <= 1.03 * float(per.mean()):
            return uniform
    return bounds


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
        self._reducer = GradExchange(self._n, self._d, self._bounds, self._rank, self._world, self._group)

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
        # (the all-gather exchange overwrites the other ranks' rows: nothing to zero then)
        alloc = torch.zeros if reducer.needs_zero() else torch.empty
        buf = alloc(n * d + 1, dtype=torch.float32, device=X.device)
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
