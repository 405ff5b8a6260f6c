"""Multi-GPU evaluation: one process per GPU, vertex-range ownership, one exchange step per evaluation.

The objective is a sum over edges, so it shards with ONE exchange step per evaluation
(SURVEY section 8e).  Every rank keeps a full replica of X and builds edge plans for the
vertex ranges it owns; the fused kernel writes only those rows of the gradient -- each already
final, owner-computes, no atomics -- plus the rank's share of the loss into one buffer
``[grad | loss]``.  The exchange leaves every rank with the identical full gradient and loss; the
optimiser then runs replicated (no further communication).

Ownership and exchange (``ShardLayout``):

* **uniform, K slices** (chosen whenever it keeps the half-edge balance within 3 %): the rows are cut
  into K slices, every slice into ``world`` equal ranges, and rank r owns range r of every slice.  The
  exchange of slice k is an IN-PLACE all-gather (this rank's rows already sit at their offset of the
  slice: no staging copy, no zero-fill, no reduction arithmetic), issued on a side stream as soon as
  slice k's kernel has finished -- it travels over xGMI while slice k + 1 is computed (round 5: at
  d = 128 the gather of 256 MB takes longer than a rank's kernels, so the overlap is what an 8-GPU
  evaluation costs; at d = 2 one slice is the whole thing -- a second launch costs more than the 1 MB
  it would hide).  The ranks' loss shares travel as ONE all-reduce of a single float.
* **balanced bounds** (graphs whose degrees are skewed along the vertex order): one contiguous range
  per rank with boundaries that balance the half-edge count (``mde_shard_bounds``) and one
  ``all_reduce(SUM)`` of the zero-padded n*d+1 floats.  Every element has exactly one non-zero
  contribution, so the result is what the gather gives.

The gather path is verified against the all-reduce on first use; a backend that cannot run it is
reported (``warnings``) and replaced by the all-reduce.

``ShardedMDE`` is an ``MDE`` whose plans cover this rank's ranges and whose evaluations go through
``ShardedEvaluator``; ``MDE.embed`` works unchanged on it.  Arbitrary callables (no ``_hip_spec``)
are sharded too: distances and the callable run replicated (elementwise work over the p edges), the
scatter of the gradient -- the expensive stage -- only over the owned rows.
"""
import ctypes
import os
import warnings

import torch
import torch.distributed as dist

from pymde_amd import _lib
from pymde_amd import average_distortion as _ad
from pymde_amd import problem


def _active(group=None, force=False):
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)


def all_reduce_grad_loss(buf, group=None, force=False):
    """Sum the ``[grad | loss]`` buffers of all ranks in place (the only data-path collective)."""
    if _active(group, force):
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


class GradExchange(object):
    """The per-evaluation exchange of a sharded problem with ONE contiguous range per rank.

    Default: one ``all_reduce(SUM)`` of the zero-padded ``[grad | loss]`` buffer.  When every rank
    owns the same number of rows the exchange is an **all-gather** instead -- each rank contributes
    only the rows it owns (already final: owner-computes), i.e. half the bytes of a ring all-reduce
    and no reduction arithmetic -- plus an all-reduce of the one-float loss share.  The gather path is
    verified against the all-reduce on its first use (bitwise, gradient) and replaced by it, with a
    warning, if the backend cannot run it or the results differ."""

    def __init__(self, n, d, bounds, rank, world, group=None, force=False):
        self.n, self.d, self.rank, self.world, self.group = int(n), int(d), int(rank), int(world), group
        self.bounds = [int(b) for b in bounds]
        sizes = [self.bounds[r + 1] - self.bounds[r] for r in range(world)]
        # force: issue the collectives even in a world of one (a single-GPU box can then run the
        # RCCL calls themselves -- tests/test_gpu_rccl.py)
        self.force = bool(force)
        self.uniform = (world > 1 or self.force) and len(set(sizes)) == 1
        self.mode = None  # decided at the first exchange

    def _gather(self, buf):
        N = self.n * self.d
        lo, hi = self.bounds[self.rank] * self.d, self.bounds[self.rank + 1] * self.d
        # in place: this rank's rows already sit at their offset of the gathered buffer (the
        # NCCL / RCCL in-place all-gather form), so no staging copy; rows owned by other ranks
        # are overwritten and need not be zeroed first.  The loss shares are summed by the
        # collective itself (one float: no gather + torch.sum launch).
        # (plain calls, not async_op=True + wait(): with RCCL both forms only make the current STREAM wait for the
        # collective, but the handles cost the host 45 us per step against 17 -- and an 8-way shard of config 4 has
        # 38 us of GPU work per step to hide the host behind; tools/exchange_host_cost.py)
        dist.all_gather_into_tensor(buf[:N], buf[lo:hi], group=self.group)
        dist.all_reduce(buf[N:N + 1], op=dist.ReduceOp.SUM, group=self.group)
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
                    else:
                        warnings.warn("pymde_amd.distributed: the in-place all-gather of the gradient rows differs "
                                      "from the all-reduce on this backend; using the all-reduce")
                except (RuntimeError, NotImplementedError, ValueError) as exc:
                    # a backend without all_gather_into_tensor (or without it for this device type)
                    warnings.warn("pymde_amd.distributed: all-gather exchange not available on this backend "
                                  "(%s: %s); using the all-reduce" % (type(exc).__name__, str(exc).split("\n")[0]))
                    self.mode = "all_reduce"
        if self.mode == "all_gather":
            return self._gather(buf)
        return all_reduce_grad_loss(buf, self.group, self.force)


def shard_range(bounds, rank):
    """Vertex range [lo, hi) of ``rank`` given the world+1 boundaries."""
    return int(bounds[rank]), int(bounds[rank + 1])


def _half_edges_per_row(n_items, edges):
    deg = torch.zeros(int(n_items), dtype=torch.int64, device=edges.device)
    ones = torch.ones(edges.shape[0], dtype=torch.int64, device=edges.device)
    deg.index_add_(0, edges[:, 0], ones)
    deg.index_add_(0, edges[:, 1], ones)
    return deg


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
        per = _half_edges_per_row(n, edges).view(world_size, step).sum(1).double()
        if float(per.max()) <= 1.03 * float(per.mean()):
            return uniform
    return bounds


class ShardLayout(object):
    """Which rows every rank owns.  ``ranges[r]`` = the list of [lo, hi) ranges of rank r, in the order they
    are evaluated and exchanged; ``slices`` = K > 0 for the uniform slice-major layout (range k of every
    rank lies in slice k = rows [k n / K, (k + 1) n / K), rank r at offset r of it), 0 for balanced bounds."""

    def __init__(self, n, world, ranges, slices, bounds=None):
        self.n, self.world, self.ranges, self.slices, self.bounds = int(n), int(world), ranges, int(slices), bounds

    @staticmethod
    def uniform(n, world, slices):
        n, world, K = int(n), int(world), int(slices)
        m = n // (world * K)
        assert m * world * K == n
        return ShardLayout(n, world, [[((k * world + r) * m, (k * world + r + 1) * m) for k in range(K)]
                                      for r in range(world)], K,
                           bounds=[r * m for r in range(world + 1)] if K == 1 else None)

    @staticmethod
    def from_bounds(n, world, bounds):
        return ShardLayout(n, world, [[shard_range(bounds, r)] for r in range(world)], 0, bounds=[int(b) for b in bounds])


def default_slices(n_items, d, world_size):
    """Slices of the uniform layout: enough that a slice's all-gather (n d 4 / K bytes in total) hides behind
    the next slice's kernel, few enough that a slice still fills the GPU.  Below ~8 MB per rank one slice:
    the exchange is latency-bound there and a second kernel launch costs more than the gather it would hide
    (DESIGN 4).  ``MDE_SHARD_SLICES`` overrides."""
    e = os.environ.get("MDE_SHARD_SLICES")
    if e:
        return max(1, int(e))
    per_rank = 4.0 * n_items * d / max(world_size, 1)
    return 4 if per_rank >= (8 << 20) else 1


def shard_layout(n_items, edges, world_size, d=2, slices=None):
    """The layout ``ShardedMDE`` uses: uniform slices when they balance the half-edges within 3 %, balanced
    bounds otherwise (identical on every rank: computed from the edge list alone)."""
    n, W = int(n_items), int(world_size)
    K = default_slices(n, d, W) if slices is None else max(1, int(slices))
    deg = None
    while K >= 1:
        if W >= 1 and n % (W * K) == 0:
            if deg is None:
                deg = _half_edges_per_row(n, edges)
            m = n // (W * K)
            per = deg.view(K, W, m).sum(2).sum(0).double()      # half-edges per rank
            if W == 1 or float(per.max()) <= 1.03 * float(per.mean()):
                return ShardLayout.uniform(n, W, K)
        K //= 2
    return ShardLayout.from_bounds(n, W, shard_bounds(n, edges, W))


def gather_slice(gbuf, n, d, world, slices, k, rank, group=None, side=None, main=None):
    """In-place all-gather of slice k of the uniform layout (rows [k n / K, (k + 1) n / K) of the [n, d] gradient
    at the head of ``gbuf``; rank r's rows at offset r of the slice).  With ``side`` / ``main`` streams the
    collective is enqueued behind what ``main`` holds NOW (slice k's kernel) and runs beside what ``main`` is
    given next.  Returns the async work handle."""
    K, W = int(slices), int(world)
    m = int(n) // (W * K)
    lo = (k * W) * m * int(d)
    out = gbuf[lo:lo + W * m * int(d)]
    mine = out[rank * m * int(d):(rank + 1) * m * int(d)]
    if main is None or side is None:
        return dist.all_gather_into_tensor(out, mine, group=group, async_op=True)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        # (the collective waits for the stream that is current when it is enqueued: the side stream, which
        # waits for slice k's kernel -- not for the kernels the main stream enqueues afterwards)
        return dist.all_gather_into_tensor(out, mine, group=group, async_op=True)


class ShardedEvaluator(object):
    """Evaluation + exchange of a sharded problem on this rank: fills ``gbuf = [grad (n d) | loss]`` with the
    FULL gradient and loss, identical on every rank.  One plan (and one kernel launch) per owned range; with
    the uniform layout the all-gather of slice k runs on a side stream under the kernel of slice k + 1."""

    def __init__(self, n, d, edges, function, layout, rank, world, group=None, force=False, first_plan=None):
        self.n, self.d, self.rank, self.world, self.group = int(n), int(d), int(rank), int(world), group
        self.layout = layout
        self.force = bool(force)
        self.function = function
        mine = layout.ranges[self.rank]
        # (first_plan: the plan of the first range, when the caller has built it already)
        self.plans = [first_plan if (k == 0 and first_plan is not None and
                                     (first_plan.row_lo, first_plan.row_hi) == (lo, hi)) else _ad.EdgePlan(self.n, edges, lo, hi)
                      for k, (lo, hi) in enumerate(mine)]
        self.bindings = [_ad.Binding(pl, function) for pl in self.plans]
        self.device = self.plans[0].device
        self.mode = None            # "all_gather" | "all_reduce", decided at the first exchange
        self._lossvec = torch.zeros(max(len(self.plans), 1), dtype=torch.float32, device=self.device)
        self._side = None
        self._buf = None            # the autograd path's own [grad | loss] buffer (allocated once)
        self._exchange1 = (GradExchange(self.n, self.d, layout.bounds, self.rank, self.world, group, force)
                           if layout.bounds is not None else None)

    @property
    def fused(self):
        return self.bindings[0].fused

    def _is_active(self):
        return _active(self.group, self.force)

    def buffer(self):
        """A persistent [grad | loss] buffer for callers that have none of their own (zero at first)."""
        if self._buf is None:
            self._buf = torch.zeros(self.n * self.d + 1, dtype=torch.float32, device=self.device)
        return self._buf

    # ---- local evaluation of one owned range
    def _local(self, k, X, grad, loss_k, shared):
        b = self.bindings[k]
        if b.fused:
            _ad.fused_evaluate(b, X, grad, loss_k)
            return
        # arbitrary callable: distances and f replicated (shared across the ranges), scatter over the owned rows
        lib = _lib.load()
        plan = self.plans[k]
        dist_e, gnorm, value = shared
        if grad is not None:
            with torch.cuda.device(plan.device):
                _lib.check(lib.mde_scatter(plan.handle, _lib.ptr(X), self.d, _lib.ptr(gnorm), _lib.ptr(dist_e), 1.0,
                                           _lib.ptr(grad), _lib.stream_ptr(plan.device)))
        # every rank holds the full mean already: rank 0's first range carries it, the others add 0
        loss_k.copy_(value.reshape(1) if (self.rank == 0 and k == 0) else torch.zeros_like(loss_k))

    def _shared_unfused(self, X, want_grad):
        if self.fused:
            return None
        lib = _lib.load()
        plan = self.plans[0]
        f = self.function
        dist_e = torch.empty(plan.p, dtype=torch.float32, device=X.device)
        with torch.cuda.device(plan.device):
            _lib.check(lib.mde_distances(self.n, plan.p, _lib.ptr(plan.edges), _lib.ptr(X), self.d, _lib.ptr(dist_e),
                                         _lib.stream_ptr(plan.device)))
        if not want_grad:
            with torch.no_grad():
                return dist_e, None, f(dist_e).mean().to(torch.float32)
        with torch.enable_grad():
            norms = dist_e.detach().requires_grad_(True)
            distortion = f(norms).mean()
            (gnorm,) = torch.autograd.grad(distortion, norms)
        return dist_e, gnorm.to(torch.float32).contiguous(), distortion.detach().to(torch.float32)

    # ---- the whole step
    def evaluate(self, X, gbuf, want_grad=True):
        """gbuf <- [full gradient | E(X)] (every rank the same).  X: contiguous fp32 [n, d] on this device."""
        N = self.n * self.d
        grad = gbuf[:N].view(self.n, self.d) if want_grad else None
        active = self._is_active()
        K = len(self.plans)
        shared = self._shared_unfused(X, want_grad)
        one_range = self._exchange1 is not None       # one contiguous range per rank: GradExchange does the exchange
        if active and want_grad and not one_range and self.mode is None:
            self._decide(X, gbuf)
        chunked = active and want_grad and not one_range and self.mode == "all_gather"
        if active and want_grad and not chunked and (self._exchange1.needs_zero() if one_range else True):
            gbuf[:N].zero_()   # all-reduce form: the rows of other ranks must be zero
        handles = []
        main = None
        if chunked and gbuf.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)
            main = torch.cuda.current_stream(self.device)
        for k in range(K):
            self._local(k, X, grad, self._lossvec[k:k + 1], shared)
            if chunked:
                handles.append(self._gather_slice(k, gbuf, main))
        # the rank's loss share = the sum of its ranges' shares; the ranks' shares meet in ONE all-reduce of that float
        if K == 1:
            gbuf[N:N + 1].copy_(self._lossvec[:1])
        else:
            torch.sum(self._lossvec[:K], dim=0, keepdim=True, out=gbuf[N:N + 1])
        if not active:
            return gbuf
        if chunked:
            dist.all_reduce(gbuf[N:N + 1], op=dist.ReduceOp.SUM, group=self.group)
            for w in handles:
                w.wait()
            return gbuf
        if not want_grad:
            dist.all_reduce(gbuf[N:N + 1], op=dist.ReduceOp.SUM, group=self.group)
            return gbuf
        if one_range:
            self._exchange1(gbuf)
            self.mode = self._exchange1.mode
            return gbuf
        return all_reduce_grad_loss(gbuf, self.group, self.force)

    def _gather_slice(self, k, gbuf, main):
        return gather_slice(gbuf, self.n, self.d, self.world, self.layout.slices, k, self.rank, self.group,
                            self._side, main)

    def _decide(self, X, gbuf):
        """First exchange: run the gather form and the all-reduce form on the same local results and keep the
        gather only if every rank sees them agree bit for bit."""
        N = self.n * self.d
        self.mode = "all_reduce"
        if self.layout.slices == 1:
            return  # (GradExchange decides for the one-range layout)
        try:
            a = torch.zeros_like(gbuf)
            shared = self._shared_unfused(X, True)
            for k in range(len(self.plans)):
                self._local(k, X, a[:N].view(self.n, self.d), self._lossvec[k:k + 1], shared)
            b = a.clone()
            dist.all_reduce(a[:N], op=dist.ReduceOp.SUM, group=self.group)
            hs = [self._gather_slice(k, b, None) for k in range(len(self.plans))]
            for h in hs:
                h.wait()
            flag = torch.tensor([1.0 if torch.equal(a[:N], b[:N]) else 0.0], device=gbuf.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            if float(flag.item()) == 1.0:
                self.mode = "all_gather"
            else:
                warnings.warn("pymde_amd.distributed: the sliced all-gather differs from the all-reduce on this "
                              "backend; using the all-reduce")
        except (RuntimeError, NotImplementedError, ValueError) as exc:
            warnings.warn("pymde_amd.distributed: all-gather exchange not available on this backend "
                          "(%s: %s); using the all-reduce" % (type(exc).__name__, str(exc).split("\n")[0]))
            self.mode = "all_reduce"

    # (what optim._NativeProblem calls when it finds a reducer without `evaluate`: kept for GradExchange users)
    def __call__(self, buf):
        raise TypeError("ShardedEvaluator.evaluate(X, gbuf) runs the kernels and the exchange together")


class ShardedMDE(problem.MDE):
    """An MDE problem whose edges are sharded across the ranks of a process group."""

    def __init__(self, n_items, embedding_dim, edges, distortion_function, constraint=None,
                 device=None, group=None, rank=None, world_size=None, slices=None, force_exchange=False,
                 shard_solver=True):
        # (force_exchange: issue the collectives even in a world of one -- single-GPU RCCL tests)
        # shard_solver (round 6): embed() keeps its vectors sharded by rows (optim._ShardedEngine: no gradient
        # exchange, the trial point's owned rows are gathered instead, the L-BFGS history is 1 / world per rank).
        # That solver wants ONE row range per rank, so the slice-major layout (the gather of one slice under the
        # kernel of the next: evaluation-only use at large d) is chosen only when slices is given explicitly.
        if slices is None and shard_solver:
            slices = 1
        self._force = bool(force_exchange)
        self._group = group
        self._rank = dist.get_rank(group) if rank is None else int(rank)
        self._world = dist.get_world_size(group) if world_size is None else int(world_size)
        self._slices = slices
        self._layout = None
        self._bounds = None
        self._d_hint = int(embedding_dim)
        super(ShardedMDE, self).__init__(n_items, embedding_dim, edges, distortion_function,
                                         constraint=constraint, device=device)
        self._check_layout_agreement()
        self._reducer = self._make_evaluator()

    def _check_layout_agreement(self):
        """Every rank must have derived the SAME ownership (slices, ranges): it follows from the edge list and from
        ``MDE_SHARD_SLICES`` in each process's own environment, and ranks that disagree would gather rows at wrong
        offsets or hang.  One MIN / MAX all-reduce of a fingerprint at construction; raises on a mismatch."""
        if not _active(self._group, False):
            return
        L = self._layout
        h = 1469598103934665603
        for r in range(L.world):
            for lo, hi in L.ranges[r]:
                h = ((h ^ (lo * 1000003 + hi)) * 1099511628211) % (1 << 61)
        backend = dist.get_backend(self._group)
        dev = self.device if backend == "nccl" else torch.device("cpu")
        v = torch.tensor([L.slices, L.n, L.world, h], dtype=torch.int64, device=dev)
        lo_t, hi_t = v.clone(), v.clone()
        dist.all_reduce(lo_t, op=dist.ReduceOp.MIN, group=self._group)
        dist.all_reduce(hi_t, op=dist.ReduceOp.MAX, group=self._group)
        if not torch.equal(lo_t, hi_t):
            raise RuntimeError("pymde_amd.distributed: the ranks derived different shard layouts (slices %d here; is "
                               "MDE_SHARD_SLICES set differently per rank, or do the ranks hold different edge lists?)"
                               % L.slices)

    def _make_plan(self, edges):
        self._layout = shard_layout(self._n, edges, self._world, d=self._d_hint, slices=self._slices)
        self._bounds = self._layout.bounds
        lo, hi = self._layout.ranges[self._rank][0]
        return _ad.EdgePlan(self._n, edges, lo, hi)

    def _make_evaluator(self):
        ev = ShardedEvaluator(self._n, self._d, self.edges, self.distortion_function, self._layout, self._rank,
                              self._world, self._group, force=self._force, first_plan=self._plan)
        return ev

    def _binding(self):
        ev = getattr(self, "_reducer", None)
        if ev is None:
            return super(ShardedMDE, self)._binding()
        # rebuilt when the distortion function was replaced OR the problem moved (MDE.to rebuilds self._plan on
        # the new device; the evaluator's plans, bindings and buffers must not stay behind on the old one)
        if (ev.function is not self.distortion_function or ev.plans[0] is not self._plan
                or str(ev.device) != str(self.device)):
            self._reducer = ev = self._make_evaluator()
        return ev.bindings[0]

    def to(self, device):
        """Move the problem to another GPU: plan, evaluator (its plans, bindings, side stream, buffers) and all."""
        super(ShardedMDE, self).to(device)
        if getattr(self, "_reducer", None) is not None:
            self._reducer = self._make_evaluator()

    def average_distortion(self, X=None):
        """E(X) with gradient, computed from this rank's shard and exchanged."""
        X = self._embedding_arg(X)
        self._binding()  # (rebinds when the distortion function was replaced)
        return _ShardedAverageDistortion.apply(X, self._reducer)


class _ShardedAverageDistortion(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, evaluator):
        Xc = X.detach().contiguous()
        n, d = Xc.shape
        buf = evaluator.buffer()   # (allocated once per problem, not per evaluation)
        evaluator.evaluate(Xc, buf, want_grad=bool(X.requires_grad))
        if X.requires_grad:
            # the caller's gradient: a copy (the buffer is overwritten by the next evaluation)
            ctx.save_for_backward(buf[:n * d].view(n, d).clone())
        return buf[n * d].clone()

    @staticmethod
    def backward(ctx, grad_output):
        (grad,) = ctx.saved_tensors
        return grad * grad_output, None
