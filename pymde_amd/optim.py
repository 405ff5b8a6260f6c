"""The projected quasi-Newton solve loop on the GPU.

``lbfgs(...)`` keeps the signature and semantics of the reference's ``pymde.optim.lbfgs``
[ref: pymde/optim.py:69-184] (driven by ``MDE.embed``, problem.py:497-510) and
``SolveStats`` is the same record [ref: optim.py:11-66]; the implementation is not a
torch.optim optimizer: iterate, trial point, gradient, direction and the L-BFGS history
live in device buffers, every vector operation is a kernel of ``csrc/mde_vec.hip``, and the
host receives ONE small read-back per objective evaluation (loss, g.d, |g|, finiteness) plus
one per iteration (the inner products of the history update) -- the reference synchronises
>= 12 times per iteration.

Behaviours reproduced from the reference: retraction inside every
line-search trial and after the accepted step; the L-BFGS pair uses the unprojected
s = t d; the gradient seen at the next step is that of the LAST EVALUATED trial while the
loss is that of the accepted one (lbfgs.py:434 vs :550); convergence is tested on the
gradient recorded at the start of the step; first step t = min(1, 1/|g|_1); ``t == 0``
resets the memory.
"""
import ctypes
import warnings
import math
import os
import time

import numpy as np
import torch

from pymde_amd import _lib
from pymde_amd import constraints as _constraints
from pymde_amd import lbfgs as _host
from pymde_amd import util


class SolveStats(object):
    """Summary statistics for a solve.

    Attributes: ``average_distortions``, ``residual_norms``, ``step_size_percents`` (one entry
    per iteration), ``solve_time`` (s), ``iterations``, ``times``, ``snapshots``,
    ``snapshot_every``.
    """

    def __init__(self, average_distortions, residual_norms, step_size_percents, solve_time, times,
                 snapshots, snapshot_every, evaluations=None):
        self.average_distortions = average_distortions
        self.residual_norms = residual_norms
        self.step_size_percents = step_size_percents
        self.solve_time = solve_time
        self.iterations = len(average_distortions)
        self.times = times
        self.snapshots = snapshots
        self.snapshot_every = snapshot_every
        self.evaluations = evaluations  # objective evaluations of the solve (not in the reference's record)

    def __str__(self):
        return ("SolveStats:\n\taverage distortion {0:.3g}\n\tresidual norm {1:.3g}\n"
                "\tsolve_time (s) {2:.3g}\n\titerations {3}".format(
                    self.average_distortions[-1], self.residual_norms[-1], self.solve_time,
                    self.iterations))

    def _repr_pretty_(self, p, cycle):
        del cycle
        p.text(self.__str__())


# indices into the statistics board (mde_vec_stats)
_GD, _GG, _G1, _GMAX, _NONFINITE, _DD, _DMAX, _XX, _LOSS = range(9)
_DIR = 16  # offset (doubles) of the direction statistics written by update_direction


class _Engine(object):
    """Device state of one solve and the kernels that act on it."""

    def __init__(self, X, memory_size, history_elements=None):
        self.lib = _lib.load()
        self.device = util.require_cuda_device(X.device)
        if X.dtype != torch.float32:
            raise ValueError("pymde_amd computes in float32; got X of dtype %s" % X.dtype)
        self.n, self.d = X.shape
        self.N = self.n * self.d
        dev = self.device
        self.X = X.detach().clone().contiguous()
        self.X_trial = torch.empty_like(self.X)
        # gradient buffer with one trailing float: [grad | loss] is what a multi-GPU
        # evaluation all-reduces in a single collective.  A second trailing word holds the status
        # flag of the constraint kernels and the statistics board (512 doubles) follows at the next
        # 8-byte boundary: [loss | status | pad | board] is ONE contiguous read-back.
        pad = (self.N + 2) % 2
        self._board_off = self.N + 2 + pad                     # in floats
        self.gtail = torch.zeros(self._board_off + 1024, dtype=torch.float32, device=dev)
        self.gbuf = self.gtail[:self.N + 1]
        self.g = self.gbuf[:self.N].view(self.n, self.d)
        self.loss_dev = self.gbuf[self.N:]
        self.g_prev = torch.empty_like(self.X)
        self.dir = torch.empty_like(self.X)
        self.work = util.work_buffer(dev, self.d)
        util.reset_work_tickets(self.work)
        self.board = self.gtail[self._board_off:].view(torch.float64)
        self._ptrs = {}
        self._dir_board = None
        self.status = self.gtail[self.N + 1:self.N + 2].view(torch.int32)
        # pinned mirror of [loss | status | pad | board]
        head = self._board_off - self.N                        # floats before the board (2 or 3)
        self._head_bytes = 4 * head
        b0 = head % 2                                          # keeps the host board 8-byte aligned
        self.host_raw = torch.zeros(b0 + head + 1024, dtype=torch.float32).pin_memory()
        self.host = self.host_raw[b0 + head:].view(torch.float64)
        self.host_loss = self.host_raw[b0:b0 + 1]
        self.host_status = self.host_raw[b0 + 1:b0 + 2].view(torch.int32)
        # numpy views of the pinned mirror (a tensor index costs microseconds, and the read-back sits
        # on the critical path of every iteration)
        self._host_np = self.host.numpy()
        self._host_loss_np = self.host_loss.numpy()
        self._host_status_np = self.host_status.numpy()
        self._host_ptr = ctypes.c_void_p(self.host_raw.data_ptr() + 4 * b0)
        self._tail_ptr = ctypes.c_void_p(self.gtail.data_ptr() + 4 * self.N)
        # the solve runs on the stream that is current now; its handle is looked up once
        self._stream_obj = torch.cuda.current_stream(dev)
        self._stream = ctypes.c_void_p(self._stream_obj.cuda_stream)
        # the solve runs on the stream that is current now; its handle is looked up once
        self._stream_obj = torch.cuda.current_stream(dev)
        self._stream = ctypes.c_void_p(self._stream_obj.cuda_stream)
        if int(memory_size) > 63:
            # (the reference has no cap, optim.py:69-82; the device-resident L-BFGS keeps its pair statistics in one
            # wave.  A solve with the 63 newest pairs instead of more is still the same quasi-Newton method: go on)
            warnings.warn("pymde_amd keeps at most 63 L-BFGS pairs (memory_size=%d requested): using 63" % int(memory_size))
            memory_size = 63
        handle = ctypes.c_void_p()
        with torch.cuda.device(dev):  # the history buffer must live on X's GPU, not the current one
            _lib.check(self.lib.mde_lbfgs_create(self.N if history_elements is None else int(history_elements),
                                                 int(memory_size), ctypes.byref(handle)))
        self.lbfgs = handle

    def close(self):
        if self.lbfgs is not None:
            self.lib.mde_lbfgs_destroy(self.lbfgs)
            self.lbfgs = None

    def stream(self):
        return self._stream

    def p(self, t):
        """Device pointer of one of the engine's long-lived tensors (cached per tensor object)."""
        key = id(t)
        hit = self._ptrs.get(key)
        if hit is None or hit[0] is not t:
            hit = (t, ctypes.c_void_p(t.data_ptr()))
            self._ptrs[key] = hit
        return hit[1]

    # ---- vector kernels
    def axpy(self, alpha, x, y, out):
        _lib.check(self.lib.mde_axpy(self.N, float(alpha), _lib.ptr(x), _lib.ptr(y), _lib.ptr(out),
                                     self.stream()))

    def stats(self, g, d, x):
        _lib.check(self.lib.mde_vec_stats(self.N, self.p(g), None if d is None else self.p(d),
                                          None if x is None else self.p(x), self.p(self.board), self.p(self.work),
                                          self._stream))

    def enqueue_read(self, count):
        """Enqueue the device->host copy of [loss | status | first ``count`` doubles of the board]
        (one contiguous range)."""
        _lib.check(self.lib.mde_copy_to_host(self._host_ptr, self._tail_ptr, self._head_bytes + 8 * int(count),
                                             self.stream()))

    def finish_read(self, count):
        self._stream_obj.synchronize()  # (polling hipStreamQuery from C instead measured the same)
        vals = self._host_np[:count].copy()
        return vals, float(self._host_loss_np[0])

    def read_board(self, count):
        """One device->host read-back of the first ``count`` doubles plus the loss."""
        self.enqueue_read(count)
        return self.finish_read(count)

    # ---- L-BFGS memory
    def reset_memory(self):
        _lib.check(self.lib.mde_lbfgs_dev_reset(self.lbfgs, self.stream()))

    def update_direction(self, t_prev):
        """Stage (y, s), accept or reject the pair, run the two-loop recursion and form the new
        direction -- all on the device (``mde_lbfgs_dev_step``), no read-back.  The statistics of
        (g, d) are left in the second half of the board (``_DIR`` doubles in): they are not needed
        before the first trial evaluation of the line search has been enqueued, so they travel
        with its read-back."""
        if self._dir_board is None:
            self._dir_board = ctypes.c_void_p(self.board.data_ptr() + 8 * _DIR)
        _lib.check(self.lib.mde_lbfgs_dev_step(self.lbfgs, self.p(self.g), self.p(self.g_prev),
                                               self.p(self.dir), float(t_prev), self.p(self.dir),
                                               self._dir_board, self.p(self.work), self._stream))


class _NativeProblem(object):
    """Objective + constraint evaluated entirely by libmde_hip (built-in constraint, function
    bound to an edge plan).  ``evaluate(X, retract)`` leaves the projected gradient in the
    engine's g buffer and E(X) in loss_dev."""

    def __init__(self, engine, binding, constraint, reducer=None):
        self.e = engine
        self.binding = binding
        self.constraint = constraint
        self.reducer = reducer  # multi-GPU: all-reduce of [grad | loss]
        # the function's parameters are fixed for the duration of a solve: bind them once
        self.fstruct = binding.struct(engine.d) if binding.fused else None
        self._fref = ctypes.byref(self.fstruct) if binding.fused else None
        self._plan_handle = binding.plan.handle if binding.fused else None
        if isinstance(constraint, _constraints._Standardized):
            self.kind = "standardized"
        elif isinstance(constraint, _constraints._Centered):
            self.kind = "centered"
        elif isinstance(constraint, _constraints.Anchored):
            self.kind = "anchored"
            self.anchors, self.values = constraint._device_args(engine.device)
        else:
            raise TypeError("not a built-in constraint")

    def turn_desc(self):
        """The steady-state iteration as one descriptor (``mde_turn_desc``), or None when this problem
        cannot take that path (exchange between ranks, unfused function, anchored rows)."""
        e = self.e
        if self.reducer is not None or not self.binding.fused or self.kind not in ("centered", "standardized"):
            return None
        T = _lib.MdeTurnDesc()
        T.plan = self._plan_handle
        T.func = ctypes.cast(ctypes.pointer(self.fstruct), ctypes.c_void_p)
        T.n, T.d, T.kind = e.n, e.d, 0 if self.kind == "centered" else 1
        T.X[0], T.X[1] = e.X.data_ptr(), e.X_trial.data_ptr()
        T.g, T.g_prev, T.dir, T.loss_dev = e.g.data_ptr(), e.g_prev.data_ptr(), e.dir.data_ptr(), e.loss_dev.data_ptr()
        T.board, T.work, T.status, T.lbfgs = e.board.data_ptr(), e.work.data_ptr(), e.status.data_ptr(), e.lbfgs
        T.host_dst, T.tail_src = e._host_ptr, e._tail_ptr
        T.read_bytes = e._head_bytes + 8 * (8 + _DIR)
        T.host_loss, T.host_status, T.host_board = e.host_loss.data_ptr(), e.host_status.data_ptr(), e.host.data_ptr()
        return T

    def retract(self, X):
        e, lib = self.e, self.e.lib
        if self.kind == "centered":
            _lib.check(lib.mde_center(e.n, e.d, _lib.ptr(X), _lib.ptr(e.work), e.stream()))
        elif self.kind == "standardized":
            _lib.check(lib.mde_std_retract(e.n, e.d, _lib.ptr(X), 1, _lib.ptr(e.work),
                                           _lib.ptr(e.status), e.stream()))
        else:
            _lib.check(lib.mde_anchor_rows(self.anchors.numel(), e.d, _lib.ptr(self.anchors),
                                           _lib.ptr(self.values), _lib.ptr(X), e.stream()))

    def retract_step(self, t, out):
        """out <- retraction(X + t dir): the line search's trial point, the step folded into the
        retraction's first pass where the library can."""
        e, lib = self.e, self.e.lib
        if self.kind == "centered":
            _lib.check(lib.mde_center_step(e.n, e.d, e.p(e.X), e.p(e.dir), float(t), e.p(out), e.p(e.work), e._stream))
        elif self.kind == "standardized":
            _lib.check(lib.mde_std_retract_step(e.n, e.d, e.p(e.X), e.p(e.dir), float(t), e.p(out), 1, e.p(e.work),
                                                e.p(e.status), e._stream))
        else:
            e.axpy(t, e.dir, e.X, out)
            self.retract(out)

    def value_grad_stats(self, X):
        """value_and_grad(X) followed by the statistics of (g, dir, X) on the board."""
        e, lib = self.e, self.e.lib
        if self.kind == "standardized":
            self.value_and_grad(X, project=False)
            _lib.check(lib.mde_std_tangent_stats(e.n, e.d, e.p(X), e.p(e.g), e.p(e.dir), e.p(e.board), e.p(e.work),
                                                 e._stream))
        else:
            self.value_and_grad(X)
            e.stats(e.g, e.dir, X)

    def value_and_grad(self, X, project=True):
        from pymde_amd import average_distortion as ad
        e, lib = self.e, self.e.lib
        if self.reducer is not None and hasattr(self.reducer, "evaluate"):
            # sharded problem: the evaluator runs this rank's kernels and the exchange (slice by slice, the
            # all-gather of one slice under the kernel of the next) and leaves [grad | loss] complete
            self.reducer.evaluate(X, e.gbuf)
        else:
            if self.reducer is not None and getattr(self.reducer, "needs_zero", lambda: True)():
                e.gbuf.zero_()  # (an all-gather exchange overwrites the other ranks' rows instead)
            if self.binding.fused:
                _lib.check(lib.mde_average_distortion(
                    self._plan_handle, e.p(X), e.d, self._fref, 1.0, e.p(e.g), e.p(e.loss_dev), e._stream))
            else:
                grad, value = ad._unfused(self.binding, X, True)
                e.g.copy_(grad)
                e.loss_dev.copy_(value.reshape(1))
            if self.reducer is not None:
                self.reducer(e.gbuf)
        if self.kind == "standardized":
            if project:
                _lib.check(lib.mde_std_tangent(e.n, e.d, _lib.ptr(X), _lib.ptr(e.g), _lib.ptr(e.work),
                                               e.stream()))
        elif self.kind == "anchored":
            _lib.check(lib.mde_anchor_rows(self.anchors.numel(), e.d, _lib.ptr(self.anchors), None,
                                           _lib.ptr(e.g), e.stream()))

    def check_status(self):
        # the status word travels with every read-back of the statistics board
        if self.kind == "standardized" and int(self.e._host_status_np[0]) != 0:
            raise util.SolverError("Standardized retraction failed: X^T X is singular")


class _ShardedEngine(_Engine):
    """The device state of a solve whose VECTORS ARE SHARDED BY ROWS across the ranks of a process group (round 6).

    Every rank keeps a replica of the iterate (the edge kernel gathers x_u from anywhere) but owns one contiguous row
    range [lo, hi): the kernel writes the gradient rows of that range only -- final, owner-computes -- and the
    gradient is NEVER exchanged.  The optimiser's vector work runs on the owned rows: the L-BFGS history is this
    rank's rows of (s, y) (160 MB / world at config 4), statistics, direction and trial point are formed for the owned
    rows.  What crosses the ranks per iteration [ref: the loop being sharded, lbfgs.py:461-507 and optim.py:100-175]:
      * the 4 + 5 m partial inner products of the history update: one all-reduce of doubles (``update_direction``);
      * the owned rows of the trial point: one in-place all-gather into the replica (``_ShardedProblem.retract_step``)
        -- the bytes the gradient all-gather of rounds 2-5 moved;
      * d column sums (and, Standardized, two d x d Gram matrices) for the constraint: small all-reduces;
      * ONE all-gather of a 25-double record per rank -- statistics of (g, dir, X) over its rows, the statistics
        of the new direction, its loss share -- reduced with sums and maxima by ``mde_rank_reduce``.
    In a world of one (``bench.py --emulate-world``: rank 0 of a W-way shard, kernels only) the collectives are
    skipped.  ``_solve`` drives this engine unchanged, call by call."""

    _MAXMASK = (1 << _GMAX) | (1 << _DMAX) | (1 << (_DIR + _GMAX)) | (1 << (_DIR + _DMAX))
    _REC = _DIR + 8 + 1        # board[0:8] | (unused) | direction board[16:24] | loss share

    def __init__(self, X, memory_size, lo, hi, group, rank, world, active):
        super(_ShardedEngine, self).__init__(X, memory_size, history_elements=(int(hi) - int(lo)) * int(X.shape[1]))
        import torch.distributed as dist
        self.dist = dist
        self.lo, self.hi, self.group, self.rank, self.world, self.active = int(lo), int(hi), group, int(rank), int(world), bool(active)
        self.n_own = self.hi - self.lo
        self.N_own = self.n_own * self.d
        dev = self.device
        # this rank's record and the gathered records of all ranks (doubles)
        self.rec = torch.zeros(self._REC, dtype=torch.float64, device=dev)
        self.rec_all = torch.zeros((self.world if self.active else 1) * self._REC, dtype=torch.float64, device=dev)
        self.ndots = int(self.lib.mde_lbfgs_dev_dots(self.lbfgs))
        self.small = torch.zeros(max(self.d * self.d, 8), dtype=torch.float64, device=dev)
        # (MDE_SHARD_XGATHER=0: exchange the trial point's rows as a zero-padded all-reduce -- what a backend without the
        # in-place all-gather gets anyway)
        self._gather_ok = False if os.environ.get("MDE_SHARD_XGATHER") == "0" else None
        self.loss_fresh = False   # loss_dev holds THIS RANK'S share (set by the evaluation, cleared by the reduction)

    def own(self, t):
        """This rank's rows of a full [n, d] tensor (a view); tensors of the owned shape pass through."""
        return t[self.lo:self.hi] if t.shape[0] == self.n else t

    # ---- vector kernels on the owned rows
    def axpy(self, alpha, x, y, out):
        _lib.check(self.lib.mde_axpy(self.N_own, float(alpha), _lib.ptr(self.own(x)), _lib.ptr(self.own(y)),
                                     _lib.ptr(self.own(out)), self.stream()))

    def all_reduce_small(self, t):
        if self.active:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def stats(self, g, d, x):
        """Statistics of (g, d, x) over ALL rows: this rank's part over its rows, then the ranks' records -- this
        board, the direction board of the last ``update_direction`` and the loss share the edge kernel left in
        ``loss_dev`` -- meet in one all-gather and are reduced by ``mde_rank_reduce`` into the board and ``loss_dev``."""
        part = self.rec
        _lib.check(self.lib.mde_vec_stats(self.N_own, _lib.ptr(self.own(g)), None if d is None else _lib.ptr(self.own(d)),
                                          None if x is None else _lib.ptr(self.own(x)), _lib.ptr(part), self.p(self.work),
                                          self._stream))
        # the loss travels only when loss_dev holds a fresh share (an evaluation since the last reduction): after the
        # reduction it holds the TOTAL, which must not be summed again by a statistics pass without an evaluation
        fresh = self.loss_fresh
        if fresh:
            # (the share in DOUBLE, straight from the plan: the ranks' shares are summed before the one rounding to float)
            _lib.check(self.lib.mde_plan_loss_double(self.plan_handle, ctypes.c_void_p(part.data_ptr() + 8 * (self._REC - 1)),
                                                     self._stream))
        src = part
        if self.active:
            self.dist.all_gather_into_tensor(self.rec_all, part, group=self.group)
            src = self.rec_all
        _lib.check(self.lib.mde_rank_reduce(self.world if self.active else 1, self._REC, self._MAXMASK, _lib.ptr(src),
                                            self.p(self.board), self._REC - 1 if fresh else -1,
                                            self.p(self.loss_dev) if fresh else None, self._stream))
        self.loss_fresh = False

    def update_direction(self, t_prev):
        go, gp, dr = self.own(self.g), self.own(self.g_prev), self.own(self.dir)
        _lib.check(self.lib.mde_lbfgs_dev_stage(self.lbfgs, _lib.ptr(go), _lib.ptr(gp), _lib.ptr(dr), float(t_prev),
                                                self.p(self.work), self._stream))
        self.all_reduce_small(self.work[:self.ndots])
        # (the statistics of (g, dir) over the owned rows go into the record's direction slots; the next stats() call
        # carries them across)
        dirrec = self.rec[_DIR:_DIR + 8]
        _lib.check(self.lib.mde_lbfgs_dev_finish(self.lbfgs, _lib.ptr(go), _lib.ptr(dr), _lib.ptr(dirrec), self.p(self.work),
                                                 self._stream))

    def gather_rows(self, Z):
        """Every rank's owned rows of Z -> the replica Z (in place: this rank's rows already sit at their offset)."""
        if not self.active:
            return
        flat = Z.view(-1)
        mine = flat[self.lo * self.d:self.hi * self.d]
        if self._gather_ok is not False:
            try:
                self.dist.all_gather_into_tensor(flat, mine, group=self.group)
                self._gather_ok = True
                return
            except (RuntimeError, NotImplementedError, ValueError):
                if self._gather_ok:
                    raise
                self._gather_ok = False
        # a backend without the in-place all-gather: zero the other ranks' rows and sum
        flat[:self.lo * self.d].zero_()
        flat[self.hi * self.d:].zero_()
        self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group)


class _ShardedProblem(_NativeProblem):
    """Objective + constraint of a row-sharded solve: the edge kernel on this rank's plan (no gradient exchange),
    the constraint maps on the owned rows with their small reductions summed across the ranks."""

    def __init__(self, engine, binding, constraint):
        super(_ShardedProblem, self).__init__(engine, binding, constraint, reducer=None)

    def turn_desc(self):
        return None

    def _colshift(self, Z, step=None):
        """Z_own -= column mean of the whole Z (mde_center_step in two halves: this rank's column means, summed
        across the ranks in place, subtracted with the rank's share of the rows as the scale).  step = t: Z_own is
        first formed as X_own + t dir_own in the same pass."""
        e, lib = self.e, self.e.lib
        Zo = e.own(Z)
        if step is None:
            _lib.check(lib.mde_center_step_begin(e.n_own, e.d, None, None, 0.0, _lib.ptr(Zo), e.p(e.work), e._stream))
        else:
            _lib.check(lib.mde_center_step_begin(e.n_own, e.d, _lib.ptr(e.own(e.X)), _lib.ptr(e.own(e.dir)), float(step),
                                                 _lib.ptr(Zo), e.p(e.work), e._stream))
        e.all_reduce_small(e.work[:e.d])
        _lib.check(lib.mde_center_step_end(e.n_own, e.d, _lib.ptr(Zo), e.p(e.work), float(e.n_own) / float(e.n), e._stream))

    def _standardize(self, Z):
        """Z_own <- sqrt(n) Z_own C^{-1/2}, C = Z^T Z over all rows (Z centred).  The d x d factor is formed on the
        host in float64 from the summed Gram matrix (one small read-back; the single-GPU path iterates on the device)."""
        e, lib = self.e, self.e.lib
        C = e.small[:e.d * e.d]
        Zo = e.own(Z)
        _lib.check(lib.mde_gram(e.n_own, e.d, e.d, _lib.ptr(Zo), _lib.ptr(Zo), _lib.ptr(C), e.p(e.work), e._stream))
        e.all_reduce_small(C)
        Ch = C.view(e.d, e.d).cpu().numpy()
        lam, Q = np.linalg.eigh(0.5 * (Ch + Ch.T))
        if not np.all(np.isfinite(lam)) or lam.min() <= 1e-12 * max(lam.max(), 1e-300):
            raise util.SolverError("Standardized retraction failed: X^T X is singular")
        M = math.sqrt(e.n) * (Q / np.sqrt(lam)) @ Q.T
        Md = torch.from_numpy(np.ascontiguousarray(M)).to(e.device)
        _lib.check(lib.mde_right_multiply_add(e.n_own, e.d, e.d, _lib.ptr(Zo), _lib.ptr(Md), 1.0, None, _lib.ptr(Zo), e._stream))
        e._keep_M = Md   # (alive until the kernel has read it)

    def retract(self, X, step=None):
        e = self.e
        if self.kind == "anchored":
            return super(_ShardedProblem, self).retract(X)
        self._colshift(X, step)
        if self.kind == "standardized":
            self._standardize(X)
        e.gather_rows(X)

    def retract_step(self, t, out):
        e = self.e
        if self.kind == "anchored":
            # (anchor rows are rewritten wherever they live; every rank then gathers everybody's rows)
            e.axpy(t, e.dir, e.X, out)
            super(_ShardedProblem, self).retract(out)
            e.gather_rows(out)
            return
        self.retract(out, step=t)

    def value_and_grad(self, X, project=True):
        e, lib = self.e, self.e.lib
        _lib.check(lib.mde_average_distortion(self._plan_handle, e.p(X), e.d, self._fref, 1.0, e.p(e.g), e.p(e.loss_dev),
                                              e._stream))
        e.loss_fresh = True
        e.plan_handle = self._plan_handle
        if self.kind == "standardized" and project:
            # g_own -= (1 / n) X_own (g^T X), g^T X summed over all rows
            G = e.small[:e.d * e.d]
            go, Xo = e.own(e.g), e.own(X)
            _lib.check(lib.mde_gram(e.n_own, e.d, e.d, _lib.ptr(go), _lib.ptr(Xo), _lib.ptr(G), e.p(e.work), e._stream))
            e.all_reduce_small(G)
            _lib.check(lib.mde_right_multiply_add(e.n_own, e.d, e.d, _lib.ptr(Xo), _lib.ptr(G), -1.0 / e.n, _lib.ptr(go),
                                                  _lib.ptr(go), e._stream))
        elif self.kind == "anchored":
            _lib.check(lib.mde_anchor_rows(self.anchors.numel(), e.d, _lib.ptr(self.anchors), None, _lib.ptr(e.g), e.stream()))

    def value_grad_stats(self, X):
        self.value_and_grad(X)
        self.e.stats(self.e.g, self.e.dir, X)

    def check_status(self):
        pass


class _GenericProblem(object):
    """Arbitrary ``objective_fn`` (torch autograd) and/or custom ``Constraint`` object: the
    callbacks run in Python exactly as in the reference's closure (optim.py:100-105); vectors
    and the optimizer state still live in the engine."""

    def __init__(self, engine, objective_fn, constraint):
        self.e = engine
        self.objective_fn = objective_fn
        self.constraint = constraint

    def turn_desc(self):
        return None

    def retract(self, X):
        with torch.no_grad():
            self.constraint.project_onto_constraint(X, inplace=True)

    def retract_step(self, t, out):
        self.e.axpy(t, self.e.dir, self.e.X, out)
        self.retract(out)

    def value_grad_stats(self, X):
        self.value_and_grad(X)
        self.e.stats(self.e.g, self.e.dir, X)

    def value_and_grad(self, X):
        e = self.e
        Xg = X.detach().requires_grad_(True)
        with torch.enable_grad():
            value = self.objective_fn(Xg)
            (grad,) = torch.autograd.grad(value, Xg)
        grad = grad.detach().to(torch.float32).contiguous()
        with torch.no_grad():
            grad = self.constraint.project_onto_tangent_space(X, grad, inplace=True)
            e.g.copy_(grad)
            e.loss_dev.copy_(value.detach().to(torch.float32).reshape(1))

    def check_status(self):
        pass


def _is_native(objective_fn, constraint, require_fused_single_gpu=False):
    """``objective_fn`` is ``MDE.average_distortion`` of a problem with a built-in constraint."""
    owner = getattr(objective_fn, "__self__", None)
    builtin = isinstance(constraint, (_constraints._Standardized, _constraints._Centered,
                                      _constraints.Anchored))
    native = (builtin and owner is not None and hasattr(owner, "_binding")
              and getattr(objective_fn, "__name__", "") == "average_distortion")
    if native and require_fused_single_gpu:
        native = getattr(owner, "_reducer", None) is None and bool(owner._binding().fused)
    return native


def _sharded_solver_args(objective_fn, constraint):
    """(binding, lo, hi, group, rank, world, active) when the solve can run with its vectors sharded by rows
    (``_ShardedEngine``): ``objective_fn`` is ``ShardedMDE.average_distortion`` of a problem with a built-in
    constraint and a built-in distortion function whose evaluator owns ONE contiguous row range per rank.
    ``MDE_SHARD_SOLVER=0`` keeps the replicated optimiser of rounds 2-5 (gradient all-gather, every rank runs the
    whole vector work)."""
    if os.environ.get("MDE_SHARD_SOLVER", "1") == "0" or not _is_native(objective_fn, constraint):
        return None
    owner = objective_fn.__self__
    binding = owner._binding()
    ev = getattr(owner, "_reducer", None)
    if ev is None or not hasattr(ev, "plans") or len(ev.plans) != 1 or not binding.fused:
        return None
    plan = ev.plans[0]
    if plan.is_full and not getattr(ev, "force", False) and ev.world <= 1:
        return None
    return (binding, plan.row_lo, plan.row_hi, ev.group, ev.rank, ev.world, ev._is_active())


def _make_problem(engine, objective_fn, constraint):
    """Pick the native path when ``objective_fn`` is ``MDE.average_distortion`` of a problem
    with a built-in constraint; otherwise the generic (callback) path."""
    owner = getattr(objective_fn, "__self__", None)
    if _is_native(objective_fn, constraint):
        # (binding first: a sharded owner replaces its evaluator in _binding() when the distortion function or the
        # device has changed -- the reducer read before that call would be the stale one)
        binding = owner._binding()
        reducer = getattr(owner, "_reducer", None)
        if reducer is not None and not binding.fused and not hasattr(reducer, "evaluate"):
            # the unfused path evaluates the full mean on every rank and fills only the owned
            # gradient rows: a bare exchange would produce garbage (ShardedEvaluator handles it)
            raise NotImplementedError("a sharded problem with a plain exchange object needs a built-in distortion "
                                      "function (pymde_amd.penalties / pymde_amd.losses)")
        return _NativeProblem(engine, binding, constraint, reducer)
    return _GenericProblem(engine, objective_fn, constraint)


def lbfgs(X, objective_fn, constraint, eps, max_iter, memory_size, use_line_search, use_cached_loss,
          verbose, print_every, snapshot_every, logger):
    """Minimise ``objective_fn`` over the constraint set, starting from ``X``.

    Returns ``(X_final, SolveStats)``.  See the module docstring for the semantics kept from
    the reference.  ``use_line_search=False`` takes fixed steps t (lbfgs.py:552-554).
    """
    start_time = time.time()
    average_distortions, grad_norms, step_size_percents, times, snapshots = [], [], [], [], []
    n_evals = [0]

    # (Replaying the usual iteration as one HIP graph was built in round 2 and measured SLOWER than the
    # launches it replaces on ROCm 7.2 -- config 2: 0.249 vs 0.178 ms per iteration -- and was removed in
    # round 3; cutting the launch count is what paid.)
    device = util.require_cuda_device(X.device)
    caller_stream = torch.cuda.current_stream(device)
    with torch.cuda.device(device), torch.cuda.stream(caller_stream):
        sh = _sharded_solver_args(objective_fn, constraint)
        engine = _Engine(X, memory_size) if sh is None else _ShardedEngine(X, memory_size, *sh[1:])
        try:
            with torch.no_grad():
                problem = (_make_problem(engine, objective_fn, constraint) if sh is None
                           else _ShardedProblem(engine, sh[0], constraint))
                _solve(engine, problem, eps, max_iter, use_line_search, use_cached_loss, verbose,
                       print_every, snapshot_every, logger, average_distortions, grad_norms,
                       step_size_percents, times, snapshots, n_evals)
            X_final = engine.X
        finally:
            engine._stream_obj.synchronize()
            engine.close()
    if isinstance(X, torch.Tensor) and X.shape == X_final.shape and X.device == X_final.device \
            and X.is_contiguous() and not X.requires_grad:
        X.copy_(X_final)  # the reference updates the caller's tensor in place
        X_final = X
    stats = SolveStats(average_distortions, grad_norms, step_size_percents,
                       time.time() - start_time, times, snapshots, snapshot_every, evaluations=n_evals[0])
    return X_final, stats


def _solve(e, problem, eps, max_iter, use_line_search, use_cached_loss, verbose, print_every,
           snapshot_every, logger, average_distortions, grad_norms, step_size_percents, times,
           snapshots, n_evals):
    digits = len(str(max_iter))
    start = time.time()

    n_iter = 0          # L-BFGS steps since the last reset (state["n_iter"])
    cached_loss = None  # loss of the accepted point (self._cached_loss)
    last_gg = 0.0       # ||g||^2 of the last evaluated gradient (X.grad)
    norm_X = None       # ||X||_F of the current iterate (None: not known yet)
    t = 0.0

    def evaluate_at_current():
        """closure() at X: no move, no retraction (lbfgs.py:426)."""
        problem.value_and_grad(e.X)
        n_evals[0] += 1
        e.stats(e.g, None, e.X)
        vals, loss = e.read_board(8)
        return loss, vals

    last_eval = {"t": None}

    def enqueue_trial(tt):
        problem.retract_step(tt, e.X_trial)
        problem.value_grad_stats(e.X_trial)
        n_evals[0] += 1

    def finish_trial(tt, extra=0):
        v, f = e.finish_read(8 + extra)
        last_eval["t"] = tt
        last_eval["gg"] = v[_GG]
        last_eval["xx"] = v[_XX]
        return (f, v[_GD], v[_NONFINITE] == 0), v

    def phi(tt, extra=0):
        enqueue_trial(tt)
        e.enqueue_read(8 + extra)
        return finish_trial(tt, extra)

    # The next iteration's direction update and first trial (t = 1) are enqueued as soon as the line
    # search has accepted a point -- before this iteration's bookkeeping, which then runs while the GPU
    # works (the iteration is a chain of ~10 short kernels behind ~40 us of Python).  Where the library
    # can describe the whole iteration (mde_turn_desc), the wait, the acceptance test of the first trial
    # and the launch of the next iteration are ONE call (mde_turn_wait): no Python between the
    # read-back and the next kernel.
    ahead = False
    turn = problem.turn_desc() if (use_line_search and use_cached_loss and not os.environ.get("MDE_NO_TURN")) else None
    if turn is not None:
        turn_ref = ctypes.byref(turn)
        turn_bufs = (e.X, e.X_trial)              # the tensors behind turn.X[0], turn.X[1]
        turn_out = np.zeros(24, dtype=np.float64)
        turn_out_ptr = ctypes.c_void_p(turn_out.ctypes.data)

    for iteration in range(max_iter):
        if snapshot_every is not None and iteration % snapshot_every == 0:
            snapshots.append(e.X.detach().cpu().clone())

        # ---- opt.step(value_and_grad)  [lbfgs.py:390-590 with max_iter = 1]
        if use_cached_loss and n_iter > 0 and use_line_search and cached_loss is not None:
            loss = cached_loss
        else:
            loss, vals = evaluate_at_current()
            last_gg = vals[_GG]
            norm_X = math.sqrt(vals[_XX])
        if norm_X is None:                        # ||X||_F before the step (optim.py:129-130)
            e.stats(e.X, None, None)
            vals, _ = e.read_board(8)
            norm_X = math.sqrt(vals[_GG])
        average_distortions.append(loss)          # callback(loss, X.grad), optim.py:94-96
        grad_norms.append(math.sqrt(last_gg))

        n_iter += 1
        if n_iter == 1:
            # d = -g, empty history, H_diag = 1     (lbfgs.py:461-466)
            e.reset_memory()
            e.axpy(-2.0, e.g, e.g, e.dir)           # dir = g - 2 g = -g (exact)
            e.axpy(0.0, e.g, e.g, e.g_prev)         # g_prev <- g
            e.stats(e.g, e.dir, None)
            vals, _ = e.read_board(8)
            gtd, d_norm2, d_max = vals[_GD], math.sqrt(vals[_DD]), vals[_DMAX]
            g1 = vals[_G1]
            t = min(1.0, 1.0 / g1) if g1 > 0 else 1.0   # initial step (lbfgs.py:521-524), lr = 1
        last_eval["t"] = None

        if n_iter > 1:
            # the first trial point is enqueued before the direction statistics are known; both
            # come back in one read
            t_prev, t = t, 1.0
            accepted_here = False
            if ahead and turn is not None:
                # wait, test the first trial, and (when it is accepted and the solve goes on) launch the
                # next iteration -- one call
                go_on = (iteration + 1 < max_iter and math.sqrt(last_gg) > eps
                         and not (snapshot_every is not None and (iteration + 1) % snapshot_every == 0))
                # (the iteration launched in there may carry the L-BFGS step of the one after it, behind a
                # gate: only when that one is inside the loop too; its gradient test is made in the library)
                pre_ok = (iteration + 2 < max_iter
                          and not (snapshot_every is not None and (iteration + 2) % snapshot_every == 0))
                cur = 0 if e.X is turn_bufs[0] else 1
                _lib.check(e.lib.mde_turn_wait(turn_ref, cur, float(loss), 1 if go_on else 0, 1e-4, 0.9,
                                               float(eps) if pre_ok else -1.0, turn_out_ptr, e._stream))
                o = turn_out
                n_evals[0] += 1 if o[2] != 0.0 else 0  # (the next iteration's first trial, launched inside mde_turn_wait)
                tv = o[4:12]
                last_eval["t"], last_eval["gg"], last_eval["xx"] = t, tv[_GG], tv[_XX]
                first = (float(o[0]), tv[_GD], tv[_NONFINITE] == 0)
                v = None
                dv = o[12:20]
                accepted_here = o[1] != 0.0
                ahead_next = o[2] != 0.0
                turn_status = int(o[3])
            elif ahead:
                first, v = finish_trial(t, extra=_DIR)   # (enqueued at the end of the last iteration)
            else:
                e.update_direction(t_prev)
                if use_line_search:
                    first, v = phi(t, extra=_DIR)
                else:
                    first = None
                    v, _ = e.read_board(_DIR + 8)
            ahead = False
            if v is not None:
                dv = v[_DIR:_DIR + 8]
            gtd, d_norm2, d_max = dv[_GD], math.sqrt(dv[_DD]), dv[_DMAX]
        else:
            first = None
            accepted_here = False

        def phi_ls(tt, _cache=[first, t]):
            if _cache[0] is not None and tt == _cache[1]:
                out, _cache[0] = _cache[0], None
                return out
            _cache[0] = None
            return phi(tt)[0]

        if use_line_search and accepted_here:
            # (mde_turn_wait applied the first pass of the search: Armijo and curvature hold at t = 1)
            cached_loss = first[0]
            last_gg = last_eval["gg"]
        elif use_line_search:
            try:
                loss_new, t, _ = _host.strong_wolfe(phi_ls, t, loss, gtd, d_max)
            except _host.LineSearchError as err:
                raise util.SolverError(str(err))
            cached_loss = loss_new
            last_gg = last_eval["gg"]
        else:
            cached_loss = None

        # X <- retract(X + t d)    (lbfgs.py:551 + optim.py:135-136)
        if use_line_search and last_eval["t"] == t:
            e.X, e.X_trial = e.X_trial, e.X
            new_xx = last_eval["xx"]
        else:
            problem.retract_step(t, e.X_trial)
            e.X, e.X_trial = e.X_trial, e.X
            new_xx = None
        norm_grad = grad_norms[-1]
        if accepted_here:
            # the status word and the next launch were handled inside mde_turn_wait
            if turn_status != 0:
                raise util.SolverError("Standardized retraction failed: X^T X is singular")
            ahead = ahead_next
        else:
            problem.check_status()
            # (not when the accepted t is not the last evaluated point -- new_xx is None: the next
            # iteration first has to compute ||X|| with a statistics pass of its own, and that pass
            # would overwrite the board slots the enqueued trial reports into)
            if (use_line_search and use_cached_loss and t != 0 and norm_grad > eps and iteration + 1 < max_iter
                    and new_xx is not None
                    and not (snapshot_every is not None and (iteration + 1) % snapshot_every == 0)):
                if turn is not None:
                    pre_ok = (iteration + 2 < max_iter and math.sqrt(last_gg) > eps
                              and not (snapshot_every is not None and (iteration + 2) % snapshot_every == 0))
                    _lib.check(e.lib.mde_turn_enqueue(turn_ref, 0 if e.X is turn_bufs[0] else 1, float(t),
                                                      float(cached_loss), 1e-4, 0.9, 1 if pre_ok else 0, e._stream))
                    n_evals[0] += 1
                else:
                    e.update_direction(t)
                    enqueue_trial(1.0)
                    e.enqueue_read(8 + _DIR)
                ahead = True

        times.append(time.time() - start)
        h = t
        percent_change = 100.0 * h * d_norm2 / norm_X if norm_X > 0 else 0.0
        step_size_percents.append(float(percent_change))
        norm_X = math.sqrt(new_xx) if new_xx is not None else None

        if verbose and ((iteration % print_every == 0) or (iteration == max_iter - 1)):
            logger.info(
                "iteration %0*d | distortion %6f | residual norm %g | "
                "step length %g | percent change %g"
                % (digits, iteration, average_distortions[-1], norm_grad, h, percent_change))
        if norm_grad <= eps:
            if verbose:
                logger.info("Converged in %03d iterations, with residual norm %g"
                            % (iteration + 1, norm_grad))
            break
        elif h == 0:
            n_iter = 0  # opt.reset(): drop the memory, re-evaluate next step (optim.py:172-173)
            cached_loss = None
            e.reset_memory()
