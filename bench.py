"""Benchmark of the north-star hot path: one average_distortion forward+backward per step.

    python bench.py --gpus N --steps K --warmup W
    (N > 1 without a launcher: bench.py starts its own N ranks, one per GPU over RCCL, and rank 0 prints the
     line; under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it just joins)

Default workload = BASELINE.json configs[3] ("Synthetic scale"): n = 1M items,
|E| = 50M uniform-random edges (out-degree 50), d = 2, penalties.Log1p(exponent 1.5), weights in
{1, 2}; synthetic, seeded, generated on the device.  A step is one fused evaluation producing the
scalar loss and the full [n, d] gradient on every rank (multi-GPU: incl. the exchange of
[grad | loss]).  The problem size is fixed as N grows (strong scaling, as the metric is stated).

Prints ONE JSON line: metric/value = edges/s/iter, plus
  roofline     -- algorithmic bytes (12.32 B/edge) / fused-kernel duration (median of
                  per-launch HIP-event times on the launching stream), against the 8 TB/s HBM peak;
                  `traffic` = HBM bytes per launch from the rocprofv3 PMC passes of THIS kernel
                  source (profiles/r05_pmc_traffic.json, ignored when the source has changed since)
  cpu_baseline -- the reference's op sequence (average_distortion.py:68-105: index gathers,
                  pow/sum/sqrt, the penalty under autograd, two scatter_add_) restated in torch and
                  timed on ALL of this host's cores (kind "torch-aten-sequence"); the OpenMP CPU
                  oracle (a port of the algorithm) beside it as `port`  (N = 1 only)
The timed region is repeated (--blocks, default 10); `ms_per_step` / `value` are the MEDIAN block.

Other workloads (secondary records; same JSON shape):
  --variant 4b SURVEY 8d config 4b: the last third of the edges repulsive, PushAndPull(Log1p, Log)
  --config 5   BASELINE configs[4]: n = 500k, |E| = 20M, d = 128 (Log1p and Quadratic; the
               Standardized projection kernels at that shape are timed too); --embed: a Standardized
               embed() at that shape, s/iter with a per-kernel breakdown
  --config 2   MNIST-like preserve_neighbors (70k points, 15-NN + repulsive edges, Standardized): embed() s/iter
  --config 3   Google-Scholar-like preserve_distances (40k-node scale-free graph, Huber loss): embed() s/iter
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ITEMS = 1_000_000
OUT_DEGREE = 50
DIM = 2
HBM_PEAK_BPS = 8.0e12           # MI355X_MICROARCH.md: 8 TB/s spec
ALG_BYTES_PER_EDGE = 8 + 4      # two int32 endpoints + one fp32 parameter  
# The reference (cvxgrp/pymde, torch CPU) on config 4, measured ONCE in the build container
# (8 vCPU Xeon, torch 2.10, 8 threads; tools/ref_cpu_time.py) -- /root/reference does not
# exist on the GPU box, so this is a recorded figure from another machine, never a same-node ratio.
REFERENCE_TORCH_CPU = {"value": 5.57e6, "unit": "edges/s/iter", "cores": 8,
                       "where": "build container (8 vCPU Xeon 2.6 GHz, 8.98 s per evaluation), NOT the GPU box",
                       "source": "tools/ref_cpu_time.py, round 2 (the survey session's container measured 9.3e6)"}
# the files that define the measured kernel (k_fused_ring, its layout and its functors); the CSR
# kernels of mde_distortion.hip are not what the PMC passes measure
KERNEL_SOURCES = ["pymde_amd/csrc/mde_ring.h", "pymde_amd/csrc/mde_ring_kernel.h", "pymde_amd/csrc/mde_ring.hip",
                  "pymde_amd/csrc/mde_ring_k_log1p.hip", "pymde_amd/csrc/mde_ring_k_pushpull.hip",
                  "pymde_amd/csrc/mde_functions.h"]


# the general-d kernel (config 5's PMC record, profiles/r06_pmc_traffic_c5.json)
WIDE_SOURCES = ["pymde_amd/csrc/mde_distortion.hip", "pymde_amd/csrc/mde_functions.h"]


def source_sha(files=None):
    h = hashlib.sha256()
    for f in (files or KERNEL_SOURCES):
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(key, tag=""):
    """HBM bytes per launch recorded by the PMC passes (tools/pmc_traffic.sh), or None when the
    record is missing or belongs to another version of the kernel source.  tag: "" the headline workload, "_4b",
    "_d3", "_n2m" the secondary records of round 6."""
    name = "r06_pmc_traffic%s.json" % tag
    path = os.path.join(ROOT, "profiles", name)
    try:
        rec = json.load(open(path))
    except Exception:
        return None, None
    if key == "wide4":
        if rec.get("source_sha_wide") != source_sha(WIDE_SOURCES):
            return None, "profiles/%s is from another kernel source (stale): ignored" % name
    elif rec.get("source_sha") != source_sha():
        return None, "profiles/%s is from another kernel source (stale): ignored" % name
    b = rec.get("bytes_per_launch", {})
    v = b.get(key)
    if v is not None and key.startswith("ring"):
        # (whatever else the evaluation launches behind the ring kernel: the combine of > 2 column groups, the hub rows)
        v += sum(b.get(k, 0.0) for k in ("ring_combine", "hub_rows", "hub_finish"))
    return v, "profiles/%s (rocprofv3 --pmc, bytes/launch of the evaluation's kernels)" % name


def make_workload(device, n=N_ITEMS, deg=OUT_DEGREE, d=DIM, graph="uniform"):
    """SURVEY 8d config 4a, generated on the device from fixed seeds (torch's device generator
    instead of numpy's default_rng(0): same distribution, different stream).

    graph (round 6, secondary records): "uniform" -- the survey's graph; "hub" -- the same with 5e5 of its edges
    re-pointed at ONE vertex (a hub of degree 5e5); "powerlaw" -- preferential attachment by the copy model: the
    target of a link is, with probability 1/2, an endpoint of an EARLIER link (degrees follow a power law with
    hubs of thousands of half-edges, and they FALL along the vertex order: the early vertices collect the edges)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(0)
    p = n * deg
    src = torch.arange(n, device=device, dtype=torch.int64).repeat_interleave(deg)
    dst = torch.randint(0, n - 1, (p,), device=device, dtype=torch.int64, generator=gen)
    dst += (dst >= src).to(torch.int64)
    if graph == "hub":
        h = min(500_000, n // 2)
        hub = n // 3
        other = torch.randperm(n - 1, device=device, generator=gen)[:h]
        other += (other >= hub).to(torch.int64)
        idx = torch.randperm(p, device=device, generator=gen)[:h]
        src[idx] = hub
        dst[idx] = other
    elif graph == "powerlaw":
        r = torch.rand(p, device=device, generator=gen)
        pos = (r * torch.arange(p, device=device, dtype=torch.float64).clamp_(min=1.0)).to(torch.int64)
        coin = torch.rand(p, device=device, generator=gen) < 0.5
        # four rounds of pointer chasing approximate the copy model (each: with probability 1/2 the source of an
        # earlier link, else the current target)
        base = torch.where(coin, src[pos], dst)
        for _ in range(3):
            base = torch.where(coin, base[pos], base)
        dst = base
        bad = dst == src
        dst[bad] = (src[bad] + 1) % n
    elif graph == "clusters":
        # planted clusters of 1000 consecutive items: 2/3 of an item's links stay inside its cluster ('neighbours'),
        # the rest go anywhere ('dissimilar' pairs; weight -1 below) -- a problem with something to learn under a
        # Standardized constraint, unlike the uniform graph
        csize = 1000
        inside = (torch.arange(p, device=device) % deg) < (2 * deg) // 3
        dst_in = (src // csize) * csize + torch.randint(0, csize - 1, (p,), device=device, dtype=torch.int64, generator=gen)
        dst_in += (dst_in >= src).to(torch.int64)
        dst_in.clamp_(max=n - 1)
        dst = torch.where(inside, dst_in, dst)
        dst = torch.where(dst == src, (src + 1) % n, dst)
    elif graph != "uniform":
        raise SystemExit("unknown --graph " + graph)
    edges = torch.stack([torch.minimum(src, dst), torch.maximum(src, dst)], dim=1).contiguous()
    w = 1.0 + (torch.rand(p, device=device, generator=gen) < 0.3).to(torch.float32)
    if graph == "clusters":
        w = torch.where((torch.arange(p, device=device) % deg) < (2 * deg) // 3, w, -torch.ones_like(w))
    gen.manual_seed(0)
    X = torch.randn((n, d), device=device, generator=gen)
    X -= X.mean(0)
    return edges, w, X.contiguous()


def make_workload_survey(device, n=N_ITEMS, deg=OUT_DEGREE, d=DIM):
    """SURVEY 8d config 4a EXACTLY as the survey writes it: numpy default_rng(0) on the host for the edges and
    the weights, torch.manual_seed(0) + CPU randn for X, then uploaded -- the tensors a reference-side run of the
    recipe builds, bit for bit (bench.py --survey-seed; the default line keeps the device generator)."""
    rng = np.random.default_rng(0)
    p = n * deg
    src = np.repeat(np.arange(n), deg)
    dst = rng.integers(0, n - 1, p)
    dst += dst >= src
    edges = np.stack([np.minimum(src, dst), np.maximum(src, dst)], axis=1)
    w = (1 + (rng.random(p) < 0.3)).astype(np.float32)
    torch.manual_seed(0)
    X = torch.randn(n, d)
    X -= X.mean(0)
    return (torch.from_numpy(edges).to(device).contiguous(), torch.from_numpy(w).to(device),
            X.to(device).contiguous())


def torch_reference_sequence(X, lhs, rhs, w, exponent=1.5):
    """The reference's op sequence for one evaluation [ref: pymde/average_distortion.py:68-105,
    penalties.py:310-321], restated with plain torch ops: index gathers, pow / sum / sqrt, the
    penalty differentiated by autograd, the NaN/Inf -> 1 fix-ups, two scatter_add_."""
    diff = X[lhs] - X[rhs]
    norms = diff.pow(2).sum(dim=1).sqrt()
    with torch.enable_grad():
        norms.requires_grad_(True)
        E = (w * torch.log1p(norms.pow(exponent))).mean()
        E.backward()
    g = norms.grad / norms.detach()
    g[torch.isnan(g)] = 1.0
    g[torch.isinf(g)] = 1.0
    contrib = g[:, None] * diff
    out = torch.zeros_like(X)
    idx = lhs[:, None].expand(-1, X.shape[1])
    out.scatter_add_(0, idx, contrib)
    out.scatter_add_(0, rhs[:, None].expand(-1, X.shape[1]), -contrib)
    return float(E.detach()), out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(edges, w, X, p):
    """The CPU path timed on THIS host's cores: (1) the reference's torch op sequence on the FULL
    workload (all 50M edges, min of up to 3 passes) at the best thread count of a sweep, with the
    all-hardware-threads figure beside it; (2) the OpenMP oracle (a port of the algorithm) on the full workload."""
    from oracle import oracle
    ncores = os.cpu_count() or 1
    e_cpu = edges.cpu()
    Xc, wc = X.cpu(), w.cpu()
    lhs, rhs = e_cpu[:, 0].contiguous(), e_cpu[:, 1].contiguous()
    # all hardware threads first (what the north star asks for); scatter_add_ does not scale to every
    # SMT thread of a two-socket host, so half and a quarter of them are tried too.  The sweep runs on
    # every 25th edge (an unbiased 2M-edge sample: the edge list is sorted by source, a prefix would
    # touch only the first rows of the table)
    stride = max(p // 2_000_000, 1)
    ls, rs, ws = lhs[::stride].contiguous(), rhs[::stride].contiguous(), wc[::stride].contiguous()
    sub = int(ls.numel())
    sweep = {}
    for nt in sorted({ncores, max(ncores // 2, 1), max(ncores // 4, 1)}, reverse=True):
        torch.set_num_threads(nt)
        torch_reference_sequence(Xc, ls[:100000], rs[:100000], ws[:100000])   # warm the thread pool
        t0 = time.perf_counter()
        torch_reference_sequence(Xc, ls, rs, ws)
        sweep[nt] = sub / (time.perf_counter() - t0)
    best_nt = max(sweep, key=sweep.get)
    torch.set_num_threads(best_nt)
    times = []
    t_start = time.perf_counter()
    while len(times) < 3 and (time.perf_counter() - t_start) < 25.0:
        t0 = time.perf_counter()
        torch_reference_sequence(Xc, lhs, rhs, wc)
        times.append(time.perf_counter() - t0)
    # one full pass on ALL hardware threads too when that is affordable (the sweep predicts its duration)
    all_full = None
    if best_nt != ncores and p / sweep[ncores] < 25.0:
        torch.set_num_threads(ncores)
        t0 = time.perf_counter()
        torch_reference_sequence(Xc, lhs, rhs, wc)
        all_full = p / (time.perf_counter() - t0)
    elif best_nt == ncores:
        all_full = p / min(times)
    aten = {"value": p / min(times), "unit": "edges/s/iter", "cores": int(best_nt), "kind": "torch-aten-sequence",
            "cpu": cpu_model(), "host_threads": int(ncores),
            "all_hardware_threads": {"cores": int(ncores), "value_full_workload": all_full,
                                     "value_on_sweep_sample": sweep[ncores]},
            "thread_sweep_edges_per_s": {str(k): v for k, v in sweep.items()},
            "sample": "ALL %d edges (full n=1M x 2 table, Log1p), min of %d fwd+bwd passes of the reference's op "
                      "sequence (gathers, pow/sum/sqrt, autograd penalty, 2x scatter_add_) in torch %s with "
                      "torch.set_num_threads(%d) -- the best of %s threads on every %d-th edge (%d edges) -- on this host; "
                      "all %d hardware threads: see all_hardware_threads"
                      % (p, len(times), torch.__version__, best_nt, sorted(sweep), stride, sub, ncores)}
    # (2) the OpenMP port on the full workload; thread count calibrated on a 10 % sample (the oracle's
    # per-thread gradient accumulators make very wide runs slower)
    e = edges.cpu().numpy()
    wn = w.cpu().numpy()
    Xn = X.cpu().numpy()
    L = oracle.lib()
    max_threads = L.oracle_num_threads()
    sub = max(p // 10, 1000)
    fsub = oracle.func("LOG1P", wn[:sub], None, (1.5,))
    best_t, best = 1, None
    for t in [c for c in (8, 16, 32, 64, 128) if c <= max_threads] or [max_threads]:
        L.oracle_set_num_threads(t)
        oracle.average_distortion(e[:sub], Xn, fsub)
        t0 = time.perf_counter()
        oracle.average_distortion(e[:sub], Xn, fsub)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best_t, best = t, dt
    L.oracle_set_num_threads(best_t)
    fd = oracle.func("LOG1P", wn, None, (1.5,))
    times = []
    t_start = time.perf_counter()
    while len(times) < 2 and (time.perf_counter() - t_start) < 12.0:
        t0 = time.perf_counter()
        E, _ = oracle.average_distortion(e, Xn, fd)
        times.append(time.perf_counter() - t0)
    aten["port"] = {"value": p / min(times), "unit": "edges/s/iter", "cores": int(best_t), "kind": "port",
                    "sample": "full workload, min of %d fwd+bwd evaluations of oracle/mde_oracle.c with OpenMP on %d "
                              "of %d host threads (best of a thread sweep)" % (len(times), best_t, max_threads)}
    aten["reference_torch_cpu_other_machine"] = REFERENCE_TORCH_CPU
    return aten, E


SPINUP_LOG = []


def spin_up(fn, device, group=20, tol=0.01, budget_s=0.8, label=None, min_s=0.3):
    """Run `fn` back to back until the GPU has left its idle clocks: groups of `group` launches, each group timed with
    one HIP-event pair, for at least `min_s` seconds AND until three consecutive groups agree within `tol`, or until
    `budget_s` seconds have gone by.  (Round 5's driver line -- `--steps 20` after an idle gap -- was still ramping: its
    11 blocks fell from 0.207 to 0.167 ms per step and a 20-launch secondary read 23 % high.  Round 6, first form: no
    minimum time -- right behind other GPU processes three groups agreed after 37 ms on a plateau below the top clock and
    the blocks then fell from 0.222 to 0.162.)  Untimed; returns the launches it made."""
    t0 = time.perf_counter()
    times = []
    n = 0
    while True:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(group):
            fn()
        b.record()
        b.synchronize()
        n += group
        times.append(a.elapsed_time(b))
        el = time.perf_counter() - t0
        if el >= min_s and len(times) >= 3 and max(times[-3:]) <= (1.0 + tol) * min(times[-3:]):
            break
        if el > budget_s:
            break
    SPINUP_LOG.append({"before": label or "timed launches", "launches": n,
                       "last_groups_ms_per_launch": [round(t / group, 5) for t in times[-3:]]})
    return n


def time_launches(fn, count, device, spin=True):
    """Median / mean duration (ms) of `count` launches of fn, each bracketed by HIP events on
    the current stream (after a spin-up: see spin_up)."""
    if spin:
        spin_up(fn, device)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(count)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize(device)
    t = np.array([a.elapsed_time(b) for a, b in ev])
    return float(np.median(t)), float(t.mean())


# ------------------------------------------------------------------------------------------ config 4
def timed_blocks(step, barrier, steps, blocks, world, device):
    """`blocks` repetitions of the timed region (barrier + sync, `steps` steps, barrier + sync);
    seconds per block, MAX over ranks."""
    out = []
    for _ in range(blocks):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        out.append(time.perf_counter() - t0)
    t = torch.tensor(out, dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t.cpu()]


def run_config4(args, world, rank, device):
    import pymde_amd
    from pymde_amd import distributed
    from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate

    n, d = args.n, args.dim
    edges, w, X = make_workload_survey(device, n=n, d=d) if args.survey_seed else make_workload(device, n=n, deg=args.degree, d=d, graph=args.graph)
    p = edges.shape[0]
    headline_shape = n == N_ITEMS and d == DIM and args.graph == "uniform" and args.degree == OUT_DEGREE
    if args.variant == "4b":
        # the last third of the edges repulsive (w = -1), PushAndPull(Log1p, Log)
        w = w.clone()
        w[(2 * p) // 3:] = -1.0
        f = pymde_amd.penalties.PushAndPull(w, pymde_amd.penalties.Log1p, pymde_amd.penalties.Log)
        fname = "PushAndPull(Log1p(1.5), Log(1)), weights {1,2} on the first 2/3 of the edges, -1 on the last third"
    elif args.function != "log1p":
        # secondary records: what another distortion function costs on the same kernel (compile-time
        # functors for the common kinds, the run-time functor for the rest)
        pen, los = pymde_amd.penalties, pymde_amd.losses
        dev = 0.5 + w  # deviations in {1.5, 2.5} for the losses
        f, fname = {
            "quadratic": lambda: (pen.Quadratic(w), "penalties.Quadratic"),
            "linear": lambda: (pen.Linear(w), "penalties.Linear"),
            "cubic": lambda: (pen.Cubic(w), "penalties.Cubic"),
            "huber": lambda: (pen.Huber(w, 0.5), "penalties.Huber(0.5)"),
            "log": lambda: (pen.Log(-w), "penalties.Log(1), weights in {-1,-2}"),
            "log1p2": lambda: (pen.Log1p(w, exponent=2.0), "penalties.Log1p(2)"),
            "l_huber": lambda: (los.Huber(dev, 0.5), "losses.Huber(0.5), deviations in {1.5,2.5}"),
            "l_quadratic": lambda: (los.Quadratic(dev), "losses.Quadratic, deviations in {1.5,2.5}"),
            "logistic": lambda: (pen.Logistic(w), "penalties.Logistic"),
            "power": lambda: (pen.Power(w, 2.5), "penalties.Power(2.5) (compile-time kind, run-time exponent)"),
            "power15": lambda: (pen.Power(w, 1.5), "penalties.Power(1.5)"),
            "sigmoid": lambda: (pen.Sigmoid(w, 1.0), "penalties.Sigmoid(threshold 1)"),
            "hinge": lambda: (pen.Hinge(w, 1.0), "penalties.Hinge(threshold 1)"),
            "invpower": lambda: (pen.InvPower(-w, 1), "penalties.InvPower(1), weights in {-1,-2}"),
            "logratio": lambda: (pen.LogRatio(-w, 2), "penalties.LogRatio(2), weights in {-1,-2}"),
            "l_cubic": lambda: (los.Cubic(dev), "losses.Cubic, deviations in {1.5,2.5}"),
            "l_power": lambda: (los.Power(dev, 1.5), "losses.Power(1.5), deviations in {1.5,2.5}"),
            "l_logistic": lambda: (los.Logistic(dev), "losses.Logistic, deviations in {1.5,2.5}"),
            "l_fractional": lambda: (los.Fractional(dev), "losses.Fractional, deviations in {1.5,2.5}"),
            "l_softfractional": lambda: (los.SoftFractional(dev), "losses.SoftFractional, deviations in {1.5,2.5}"),
            "runtime": lambda: (pen._ClippedQuadratic(w, 1.0) if hasattr(pen, "_ClippedQuadratic") else pen.Logistic(w),
                                "a private kind through the run-time functor on the ring kernel"),
        }[args.function]()
    else:
        f = pymde_amd.penalties.Log1p(w)
        fname = "penalties.Log1p(1.5), weights in {1,2}"
    bounds = None
    if world > 1 or args.emulate_world > 1:
        W = world if world > 1 else args.emulate_world
        bounds = distributed.shard_bounds(n, edges, W)
        lo, hi = distributed.shard_range(bounds, rank)
        # (every rank holds the full edge list here -- 0.8 GB at this size; a rank only needs the
        # edges with an endpoint in its range, which is what the plan keeps)
        plan = EdgePlan(n, edges, lo, hi)
    else:
        plan = EdgePlan(n, edges)
    binding = Binding(plan, f)
    buf = torch.zeros(n * d + 1, dtype=torch.float32, device=device)
    grad, loss = buf[:n * d].view(n, d), buf[n * d:]
    exchange = distributed.GradExchange(n, d, bounds, rank, world) if world > 1 else None

    def step():
        if world > 1 and exchange.needs_zero():
            buf.zero_()          # rows of other ranks must be zero for the all-reduce
        fused_evaluate(binding, X, grad, loss)
        if world > 1:
            exchange(buf)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    # spin-up (untimed, beside the caller's warm-up steps): until the step time has settled -- the blocks below are
    # then steady state whatever --steps is
    spinup = spin_up(step, device, label="headline blocks") if world == 1 else 0
    # the timed region (exactly `steps` steps between barrier + synchronize), repeated: the median
    # block is reported (a 5 ms region is at the mercy of one scheduler hiccup)
    block_s = timed_blocks(step, barrier, args.steps, max(args.blocks, 1), world, device)
    elapsed = float(np.median(block_s))
    # the same steps once more with a HIP-event pair around each one: the per-step median (the
    # events themselves cost a few microseconds per step, so they stay out of the timed region)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in ev:
        a.record()
        step()
        b.record()
    barrier()
    step_ms = np.array([a.elapsed_time(b) for a, b in ev])
    gpu_loss = float(loss.item())

    # dominant kernel: the fused gather/scatter kernel, timed per launch with HIP events on the
    # stream it is launched on (torch's current stream)
    k_ms, k_mean = time_launches(lambda: fused_evaluate(binding, X, grad, loss), max(args.steps, 1), device)
    per_rank_ms = [k_ms]
    if world > 1:
        t = torch.zeros(world, dtype=torch.float64, device=device)
        t[rank] = k_ms
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank_ms = [float(v) for v in t.cpu()]
    edges_local = plan.half_edges / 2.0
    alg_bytes = ALG_BYTES_PER_EDGE * edges_local + 2.0 * 4.0 * d * (plan.row_hi - plan.row_lo)
    achieved = alg_bytes / (k_ms * 1e-3)
    layout = int(binding.struct(d).layout)
    fn_name = ("Log1p" if args.function == "log1p" else args.function) if args.variant == "4a" else "PushPull<Log1p,Log>"
    ring = plan.ring_info()
    kernel = ("k_fused_ring<%d,%s,%s> (LDS-resident rows, chunk ring filled through the producers' VGPRs; loss and, with two "
              "column groups per row block, the groups' rows added in the same launch; k_ring_combine behind it otherwise)"
              % (d, fn_name, binding.stream_kind + " stream")) if layout == 1 \
        else "k_fused_flat<%d,%s> (CSR, edge-balanced tiles) + k_flat_fixup" % (d, fn_name)
    if layout == 1 and ring["hub_rows"]:
        kernel += " + k_hub_rows / k_hub_finish for %d peeled hub rows (%d half-edges)" % (ring["hub_rows"], ring["hub_half_edges"])
    if layout == 1 and ring["permuted"]:
        kernel += "; row blocks dealt by degree"
    traffic, traffic_src = (None, None)
    if world == 1 and args.emulate_world <= 1 and args.function == "log1p":
        tag = None
        if headline_shape:
            tag = "" if args.variant == "4a" else "_4b"
        elif args.variant == "4a" and args.graph == "uniform":
            tag = {(N_ITEMS, 3): "_d3", (2 * N_ITEMS, 2): "_n2m"}.get((n, d))
        if tag is not None:
            traffic, traffic_src = pmc_traffic({"codebook": "ring_codebook", "byte index": "ring_bytes"}.get(binding.stream_kind, "ring_fp32")
                                               if layout == 1 else "csr", tag)

    if rank != 0:
        return None
    out = {
        "metric": "edges/sec/iter (avg_distortion fwd+bwd), n=1M |E|=50M d=2" if headline_shape else
                  "edges/sec/iter (avg_distortion fwd+bwd), n=%d |E|=%d d=%d, %s graph" % (n, p, d, args.graph),
        "value": p * args.steps / elapsed, "unit": "edges/s/iter",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spinup_launches": spinup,
        "ms_per_step": 1e3 * elapsed / args.steps, "ms_per_step_median_events": float(np.median(step_ms)),
        "blocks": len(block_s), "ms_per_step_blocks": [round(1e3 * b / args.steps, 5) for b in block_s],
        "ms_per_step_fastest_block": 1e3 * min(block_s) / args.steps,
        "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3] / SURVEY 8d config %s%s: n=%d, |E|=%d %s edges "
                               "(out-degree %d), d=%d, %s; %s" % (args.variant, "" if headline_shape else " at ANOTHER SHAPE (secondary record)",
                                                                   n, p, {"uniform": "uniform-random", "hub": "uniform-random + one hub of degree 5e5",
                                                                          "powerlaw": "preferential-attachment (copy model)",
                                                                          "clusters": "planted-cluster"}[args.graph], args.degree, d, fname,
                               "the survey's exact tensors: numpy default_rng(0) edges and weights, torch.manual_seed(0) "
                               "CPU randn X, built on the host and uploaded (--survey-seed)" if args.survey_seed else
                               "seeded on the device with torch.Generator(0) (the survey's recipe uses numpy "
                               "default_rng(0): same distribution, another stream; --survey-seed builds that one)"),
                   "parallelism": ("vertex-range shards x%d + %s of [grad|loss]" % (world, exchange.mode))
                   if world > 1 else ("single GPU" if args.emulate_world <= 1 else
                                      "rank 0 of a %d-way shard, kernel only (emulation, no collective)" % args.emulate_world),
                   "exchange": exchange.mode if world > 1 else None,
                   "kernel_ms_per_rank": per_rank_ms,
                   "parameter_stream": ("codebook: %d distinct weights ride in the packed half-edge word "
                                        "(4 B/half-edge)" % (2 if args.variant == "4a" else 3) if binding.codebook
                                        else "byte index: one index byte per half-edge beside the packed word (5 B/half-edge)"
                                        if binding.byte_stream else "fp32 weight per half-edge (8 B/half-edge)"),
                   "ring_layout": ring if layout == 1 else None,
                   "ms_per_1e8_half_edges": 1e3 * elapsed / args.steps * 1e8 / max(plan.half_edges, 1),
                   "loss": gpu_loss},
        "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK_BPS / 1e9,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_BPS, "traffic": traffic,
                     "traffic_source": traffic_src, "kernel": kernel, "kernel_ms": k_ms,
                     "kernel_ms_mean": k_mean, "alg_bytes_per_launch": alg_bytes},
    }
    if world == 1 and binding.codebook and not args.no_codebook and args.variant == "4a" and args.function == "log1p" and headline_shape:
        # secondary: the general case (continuous per-edge parameters, configs 2 / 3) streams an
        # fp32 parameter per half-edge
        os.environ["MDE_CODEBOOK"] = "0"
        os.environ["MDE_BYTE_STREAM"] = "0"
        b2 = Binding(plan, pymde_amd.penalties.Log1p(w.clone()))
        fused_evaluate(b2, X, grad, loss)
        assert b2.stream_kind == "fp32", b2.stream_kind
        k2, _ = time_launches(lambda: fused_evaluate(b2, X, grad, loss), max(args.steps, 1), device)
        os.environ.pop("MDE_CODEBOOK")
        os.environ.pop("MDE_BYTE_STREAM")
        out["config"]["fp32_parameter_stream"] = {
            "kernel_ms": k2, "value": p / (k2 * 1e-3), "unit": "edges/s/iter (kernel time)",
            "roofline_frac": alg_bytes / (k2 * 1e-3) / HBM_PEAK_BPS}
        # ... and 40 distinct values (the hop counts of a distance problem on a graph are tens of distinct integers)
        # an index byte per half-edge beside the packed word: 5 B/half-edge
        del b2
        g40 = torch.Generator(device=device).manual_seed(3)
        w40 = 1.0 + torch.randint(0, 40, (p,), device=device, generator=g40).float() / 40.0
        b3 = Binding(plan, pymde_amd.penalties.Log1p(w40))
        fused_evaluate(b3, X, grad, loss)
        k3, _ = time_launches(lambda: fused_evaluate(b3, X, grad, loss), max(args.steps, 1), device)
        out["config"]["byte_index_parameter_stream"] = {
            "distinct_values": 40, "stream": b3.stream_kind, "kernel_ms": k3, "value": p / (k3 * 1e-3),
            "unit": "edges/s/iter (kernel time)", "roofline_frac": alg_bytes / (k3 * 1e-3) / HBM_PEAK_BPS}
        del b3
    if args.survey_seed and world == 1 and args.variant == "4a" and args.function == "log1p":
        # the loss of the survey's exact workload next to the oracle's (the OpenMP restatement on the same tensors)
        from oracle import oracle as _oracle
        o_loss, _ = _oracle.average_distortion(edges.cpu().numpy(), X.cpu().numpy(),
                                               _oracle.func("LOG1P", w.cpu().numpy(), None, (1.5,)), want_grad=False)
        out["config"]["survey_seed"] = {"loss": gpu_loss, "oracle_loss": float(o_loss),
                                        "rel_diff": abs(gpu_loss - float(o_loss)) / abs(float(o_loss))}
        assert abs(float(o_loss) - gpu_loss) <= 1e-5 * abs(float(o_loss)), (o_loss, gpu_loss)
    if (world == 1 and args.emulate_world <= 1 and not args.no_cpu_baseline and args.variant == "4a" and args.function == "log1p"
            and headline_shape):
        cb, cpu_loss = cpu_baseline(edges, w, X, p)
        out["cpu_baseline"] = cb
        out["config"]["oracle_loss"] = cpu_loss
        assert abs(cpu_loss - gpu_loss) <= 1e-5 * abs(cpu_loss), (cpu_loss, gpu_loss)
    out["spinup"] = SPINUP_LOG
    return out


def run_exchange_only(args, world, rank):
    """No kernel (CPU / gloo): every rank fills its own rows of [grad | loss] with a known pattern
    and runs the product's exchange; checks the result and prints the N-rank line.  This is what
    the CPU test of the self-launcher runs -- it is NOT a measurement of the hot path."""
    from pymde_amd import distributed
    n, d = args.n, DIM
    step = (n + world - 1) // world
    bounds = [min(r * step, n) for r in range(world + 1)]
    lo, hi = bounds[rank], bounds[rank + 1]
    ex = distributed.GradExchange(n, d, bounds, rank, world)
    buf = torch.zeros(n * d + 1, dtype=torch.float32)
    t = []
    for _ in range(args.warmup + args.steps):
        buf.zero_()
        buf[lo * d:hi * d] = float(rank + 1)
        buf[n * d] = 1.0
        dist.barrier()
        t0 = time.perf_counter()
        ex(buf)
        t.append(time.perf_counter() - t0)
    for r in range(world):
        assert bool((buf[bounds[r] * d:bounds[r + 1] * d] == float(r + 1)).all()), "exchange result"
    assert abs(float(buf[n * d]) - world) < 1e-6
    if rank != 0:
        return None
    ms = 1e3 * float(np.median(t[args.warmup:]))
    return {"metric": "exchange only (no kernel): [grad|loss] of n=%d, d=2" % n, "value": ms, "unit": "ms/exchange",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": False,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "exchange-only self-launch check (backend %s), not the hot path" % args.backend,
                       "parallelism": "vertex-range shards x%d + %s of [grad|loss]" % (world, ex.mode),
                       "exchange": ex.mode}}


# ------------------------------------------------------------------------------------------ config 5
def run_config5(args, device):
    """BASELINE configs[4]: n = 500k, |E| = 20M, d = 128 (out-degree 40, weights {1,2})."""
    import pymde_amd
    from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate
    n, deg, d = 500_000, 40, 128
    edges, w, _ = make_workload(device, n=n, deg=deg, d=2)
    p = edges.shape[0]
    c = pymde_amd.Standardized()
    torch.manual_seed(0)
    X = c.initialization(n, d, device=device).contiguous()
    plan = EdgePlan(n, edges)
    buf = torch.zeros(n * d + 1, dtype=torch.float32, device=device)
    grad, loss = buf[:n * d].view(n, d), buf[n * d:]
    alg_bytes = ALG_BYTES_PER_EDGE * p + 2.0 * 4.0 * d * n      # 37.6 B/edge
    gather_bytes = 2.0 * p * d * 4                              # every half-edge reads one 512-byte row
    res = {}
    for name, f in (("Log1p", pymde_amd.penalties.Log1p(w)), ("Quadratic", pymde_amd.penalties.Quadratic(w))):
        b = Binding(plan, f)
        for _ in range(args.warmup):
            fused_evaluate(b, X, grad, loss)
        k_ms, k_mean = time_launches(lambda: fused_evaluate(b, X, grad, loss), args.steps, device)
        res[name] = {"kernel_ms": k_ms, "kernel_ms_mean": k_mean, "value": p / (k_ms * 1e-3),
                     "row_gather_TBps": gather_bytes / (k_ms * 1e-3) / 1e12,
                     "roofline_frac": alg_bytes / (k_ms * 1e-3) / HBM_PEAK_BPS, "loss": float(loss.item())}
    # the Standardized projection at this shape (d x d Gram on the f32 MFMA path)
    Z = torch.randn((n, d), device=device)
    t_tan, _ = time_launches(lambda: c.project_onto_tangent_space(X, Z, inplace=True), 20, device)
    Y = X.clone()
    t_ret, _ = time_launches(lambda: c.project_onto_constraint(Y, inplace=True), 20, device)
    k = res["Log1p"]
    traffic, traffic_src = pmc_traffic("wide4", "_c5")
    return {
        "metric": "edges/sec/iter (avg_distortion fwd+bwd), n=500k |E|=20M d=128",
        "value": k["value"], "unit": "edges/s/iter", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": k["kernel_ms_mean"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[4] / SURVEY 8d config 5: n=%d, |E|=%d uniform-random edges "
                               "(out-degree 40), d=128, penalties.Log1p(1.5) (Quadratic alongside), weights in "
                               "{1,2}, X = Standardized().initialization" % (n, p),
                   "parallelism": "single GPU", "functions": res,
                   "standardized_tangent_ms": t_tan, "standardized_retract_ms": t_ret},
        "roofline": {"bound": "hbm", "achieved": alg_bytes / (k["kernel_ms"] * 1e-3) / 1e9,
                     "peak": HBM_PEAK_BPS / 1e9, "unit": "GB/s", "frac": k["roofline_frac"], "traffic": traffic,
                     "traffic_source": traffic_src,
                     "kernel": "k_fused_wide4p<8> (CSR, 8 lanes x four float4s per 512-byte row, 8 half-edges per wave step, gathers of the next step in flight, XCD-aware row chunks)",
                     "kernel_ms": k["kernel_ms"], "alg_bytes_per_launch": alg_bytes,
                     "note": "the 37.6 B/edge figure assumes every row is read once; a uniform-random graph "
                             "at d = 128 has no reuse to exploit (each half-edge needs its own 512-byte row: "
                             "%.1f GB per evaluation), so the attainable bound is the random-row gather rate "
                             "(tools/rowprobe: 7.4 TB/s from the 256 MB table) -> see row_gather_TBps"
                             % (gather_bytes / 1e9)},
    }


def run_config5_sharded(args, world, rank, device):
    """BASELINE configs[4] on N GPUs (or rank 0 of an N-way shard on one GPU, --emulate-world): n = 500k, |E| = 20M,
    d = 128, Log1p.  Uniform slice-major ownership (pymde_amd.distributed.ShardLayout): every rank evaluates its K
    slices one after the other and the in-place all-gather of slice k travels under the kernel of slice k + 1."""
    import pymde_amd
    from pymde_amd import distributed
    n, deg, d = 500_000, 40, 128
    edges, w, _ = make_workload(device, n=n, deg=deg, d=2)
    p = edges.shape[0]
    torch.manual_seed(0)
    X = pymde_amd.Standardized().initialization(n, d, device=device).contiguous()
    W = world if world > 1 else args.emulate_world
    layout = distributed.shard_layout(n, edges, W, d=d)
    f = pymde_amd.penalties.Log1p(w)
    ev = distributed.ShardedEvaluator(n, d, edges, f, layout, rank, W)
    buf = torch.zeros(n * d + 1, dtype=torch.float32, device=device)

    def step():
        ev.evaluate(X, buf)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(max(args.warmup, 2)):
        step()
    block_s = timed_blocks(step, barrier, args.steps, max(args.blocks, 1), world, device)
    elapsed = float(np.median(block_s))
    # this rank's kernels alone (no exchange), HIP events around the K launches
    shared = None

    def local_only():
        for k in range(len(ev.plans)):
            ev._local(k, X, buf[:n * d].view(n, d), ev._lossvec[k:k + 1], shared)
    k_ms, _ = time_launches(local_only, max(args.steps, 1), device)
    per_rank = [k_ms]
    if world > 1:
        t = torch.zeros(world, dtype=torch.float64, device=device)
        t[rank] = k_ms
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank = [float(v) for v in t.cpu()]
    if rank != 0:
        return None
    gather_bytes = 4.0 * n * d * (W - 1) / W
    return {
        "metric": "edges/sec/iter (avg_distortion fwd+bwd), n=500k |E|=20M d=128", "value": p * args.steps / elapsed,
        "unit": "edges/s/iter", "n_gpus": max(world, 1), "steps": args.steps, "warmup": max(args.warmup, 2),
        "ms_per_step": 1e3 * elapsed / args.steps, "blocks": len(block_s),
        "ms_per_step_blocks": [round(1e3 * b / args.steps, 5) for b in block_s], "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[4] / SURVEY 8d config 5: n=%d, |E|=%d uniform-random edges (out-degree 40), "
                               "d=128, penalties.Log1p(1.5), weights in {1,2}, X = Standardized().initialization" % (n, p),
                   "parallelism": ("vertex-range shards x%d, %d slices per rank, exchange %s" % (world, max(layout.slices, 1), ev.mode))
                   if world > 1 else "rank 0 of a %d-way shard, %d slices, kernels only (emulation, no collective)"
                   % (W, max(layout.slices, 1)),
                   "exchange": ev.mode if world > 1 else None, "slices": layout.slices,
                   "kernel_ms_per_rank": per_rank,
                   "all_gather_bytes_received_per_rank": gather_bytes,
                   "loss": float(buf[n * d].item())},
    }


def run_config5_embed(args, device):
    """BASELINE configs[4] as an embed(): n = 500k, |E| = 20M, d = 128, Log1p, Standardized -- seconds
    per projected L-BFGS iteration, with the component kernels timed on their own beside it."""
    import pymde_amd
    from pymde_amd.average_distortion import fused_evaluate
    n, deg, d = 500_000, 40, 128
    edges, w, _ = make_workload(device, n=n, deg=deg, d=2)
    p = edges.shape[0]
    c = pymde_amd.Standardized()
    # The mean over 2e7 edges of an O(1) penalty has gradient entries of ~5e-9: below the float32
    # resolution of the O(1) coordinates, so X + t d == X and the unscaled problem cannot leave its start in
    # float32 (the reference forms the same update in float32).  The weights are therefore scaled by 1e5
    # for the embed() record -- the arithmetic per iteration is the same.
    wscale = 1.0e5
    mde = pymde_amd.MDE(n, d, edges, pymde_amd.penalties.Log1p(w * wscale), constraint=c, device=device)
    torch.manual_seed(0)
    X0 = c.initialization(n, d, device=device).contiguous()
    iters = max(args.steps if args.steps != 200 else 20, 1)
    mde.embed(X=X0.clone(), max_iter=3, eps=0.0)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    mde.embed(X=X0.clone(), max_iter=iters, eps=0.0)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    st = mde.solve_stats
    n_it = max(int(st.iterations), 1)
    # the components of an iteration, each on its own (HIP events, median of 20)
    X = mde.X.detach().contiguous()
    buf = torch.zeros(n * d + 1, dtype=torch.float32, device=device)
    grad, loss = buf[:n * d].view(n, d), buf[n * d:]
    b = mde._binding()
    t_eval, _ = time_launches(lambda: fused_evaluate(b, X, grad, loss), 20, device)
    Z = torch.randn((n, d), device=device)
    t_tan, _ = time_launches(lambda: c.project_onto_tangent_space(X, Z, inplace=True), 20, device)
    Y = X.clone()
    t_ret, _ = time_launches(lambda: c.project_onto_constraint(Y, inplace=True), 20, device)
    vec_bytes = 4.0 * n * d
    return {
        "metric": "seconds/iteration of MDE.embed(), n=500k |E|=20M d=128 Standardized", "value": dt / n_it, "unit": "s/iter",
        "n_gpus": 1, "steps": n_it, "warmup": 3, "ms_per_step": 1e3 * dt / n_it, "higher_is_better": False,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[4] / SURVEY 8d config 5 as an embed(): n=%d, |E|=%d uniform-random edges "
                               "(out-degree 40), d=128, penalties.Log1p(1.5), weights {1,2} x 1e5 (unscaled, the gradient "
                               "entries of ~5e-9 are below float32 resolution of the coordinates and no float32 solver "
                               "moves), Standardized, L-BFGS memory 10, X0 = Standardized().initialization" % (n, p),
                   "parallelism": "single GPU", "edges_per_s_per_iter": p * n_it / dt,
                   "average_distortions": [float(v) for v in st.average_distortions],
                   "component_ms": {"average_distortion fwd+bwd (k_fused_wide4p)": t_eval,
                                    "Standardized tangent projection": t_tan,
                                    "Standardized retraction": t_ret},
                   "vector_bytes": vec_bytes,
                   "note": "an iteration = >= 1 evaluation + tangent projection + retraction per line-search trial, "
                           "plus the device L-BFGS update (4 + 5m inner products and the combine over %.0f MB "
                           "vectors); the kernel-by-kernel split is in profiles/r03_config5_embed_kernel_stats.csv"
                           % (vec_bytes / 1e6)},
    }


def run_config4_embed(args, device):
    """The headline shape as an embed() (SURVEY 8d's secondary metric): n = 1M, |E| = 50M, d = 2, Log1p,
    Centered and Standardized -- seconds per projected L-BFGS iteration after a warm-up solve, objective
    evaluations per iteration, and what building the problem costs (plan / ring layout / parameters)."""
    import ctypes
    import pymde_amd
    from pymde_amd import _lib
    from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate
    n, d = args.n, args.dim
    edges, w, _ = make_workload(device, n=n, d=d, graph=args.graph)
    p = edges.shape[0]
    lib = _lib.load()
    pen = pymde_amd.penalties
    make_f = (lambda: pen.PushAndPull(w, pen.Log1p, pen.Log)) if args.graph == "clusters" else (lambda: pen.Log1p(w))

    def timed(fn):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize(device)
        return r, time.perf_counter() - t0

    # what the problem costs to build, stage by stage (the same calls MDE.__init__ and the first
    # evaluation make)
    plan, t_plan = timed(lambda: EdgePlan(n, edges))
    _, t_layout = timed(lambda: lib.mde_plan_layout(plan.handle, d, _lib.stream_ptr(device)))
    f = make_f()
    binding, t_bind = timed(lambda: Binding(plan, f))
    _, t_struct = timed(lambda: binding.struct(d))
    del binding, plan
    # (the reference's embed() runs max_iter = 300 by default; 100 keeps the run short and is long enough that
    # the first steps from a random start -- several trials per line search -- no longer set the average)
    iters = max(args.steps if args.steps != 200 else 100, 1)
    records = {}
    for cname, c in (("Centered", pymde_amd.Centered()), ("Standardized", pymde_amd.Standardized())):
        if args.emulate_world > 1:
            # rank 0 of a W-way row-sharded solve, no collectives (a world of one): the per-rank kernels of an iteration --
            # the edge kernel on an eighth of the rows, the optimiser's vector work on the owned rows only
            from pymde_amd import distributed
            mde, t_mde = timed(lambda: distributed.ShardedMDE(n, d, edges, make_f(), constraint=c, device=device, rank=0,
                                                              world_size=args.emulate_world))
        else:
            mde, t_mde = timed(lambda: pymde_amd.MDE(n, d, edges, make_f(), constraint=c, device=device))
        torch.manual_seed(0)
        X0 = c.initialization(n, d, device=device).contiguous()
        _, t_first = timed(lambda: mde.embed(X=X0.clone(), max_iter=5, eps=0.0))     # warm-up (builds layout + binding)
        _, dt = timed(lambda: mde.embed(X=X0.clone(), max_iter=iters, eps=0.0))
        st = mde.solve_stats
        n_it = max(int(st.iterations), 1)
        # the same solve continued: iterations 6 .. 5 + iters of ONE embed() (SolveStats.times), i.e. without
        # the first steps from a random start, whose line searches need several trials each
        _, _ = timed(lambda: mde.embed(X=X0.clone(), max_iter=iters + 5, eps=0.0))
        st2 = mde.solve_stats
        tt = list(st2.times)
        late = (tt[-1] - tt[4]) / max(len(tt) - 5, 1) if len(tt) > 5 else float("nan")
        X = mde.X.detach().contiguous()
        buf = torch.zeros(n * d + 1, dtype=torch.float32, device=device)
        grad, loss = buf[:n * d].view(n, d), buf[n * d:]
        b = mde._binding()
        t_eval, _ = time_launches(lambda: fused_evaluate(b, X, grad, loss), 20, device)
        if args.emulate_world > 1:
            # The solve above cannot tell what an iteration costs: without the other ranks the iterate's foreign rows
            # are never refreshed, the objective stops making sense and every line search runs long.  So the kernels
            # and host calls of ONE steady-state iteration (first trial accepted: direction update, trial point,
            # evaluation, statistics, one read-back) are timed in a loop on rank 0's engine.
            from pymde_amd import optim as _optim
            sargs = _optim._sharded_solver_args(mde.average_distortion, mde.constraint)
            iter_ms = None
            if sargs is not None:
                with torch.no_grad():
                    eng = _optim._ShardedEngine(X0.clone(), 10, *sargs[1:])
                    prob = _optim._ShardedProblem(eng, sargs[0], mde.constraint)
                    prob.value_and_grad(eng.X)
                    eng.reset_memory()
                    eng.axpy(-2.0, eng.g, eng.g, eng.dir)
                    eng.axpy(0.0, eng.g, eng.g, eng.g_prev)

                    def one_iteration():
                        eng.update_direction(1e-3)
                        prob.retract_step(1e-3, eng.X_trial)
                        prob.value_grad_stats(eng.X_trial)
                        eng.read_board(24)
                    for _ in range(30):
                        one_iteration()
                    torch.cuda.synchronize(device)
                    t0 = time.perf_counter()
                    for _ in range(200):
                        one_iteration()
                    torch.cuda.synchronize(device)
                    iter_ms = 1e3 * (time.perf_counter() - t0) / 200
                    eng.close()
            records[cname] = {
                "steady_state_iteration_ms": iter_ms,
                "s_per_iter": dt / n_it, "ms_per_iter": 1e3 * dt / n_it, "iterations": n_it,
                "ms_per_iter_after_5_iterations": 1e3 * late, "evaluations": st.evaluations,
                "evaluations_per_iteration": (st.evaluations or 0) / n_it,
                "component_ms": {"average_distortion fwd+bwd, rank 0's rows": t_eval},
                "note": "rank 0 of a %d-way row-sharded solve in a world of one: kernels and host loop of ONE rank, no collective "
                        "(the other ranks' rows of the iterate are never refreshed, so the numbers of the solve mean nothing)"
                        % args.emulate_world}
            del mde
            continue
        Z = torch.randn((n, d), device=device)
        t_tan, _ = time_launches(lambda: c.project_onto_tangent_space(X, Z, inplace=True), 20, device)
        Y = X.clone()
        t_ret, _ = time_launches(lambda: c.project_onto_constraint(Y, inplace=True), 20, device)
        records[cname] = {
            "s_per_iter": dt / n_it, "ms_per_iter": 1e3 * dt / n_it, "iterations": n_it,
            "ms_per_iter_after_5_iterations": 1e3 * late, "evaluations_of_the_longer_solve": st2.evaluations,
            "iterations_of_the_longer_solve": int(st2.iterations),
            "evaluations": st.evaluations, "evaluations_per_iteration": (st.evaluations or 0) / n_it,
            "first_embed_5_iterations_s": t_first, "mde_constructor_s": t_mde,
            "average_distortions": [float(v) for v in st.average_distortions[:8]],
            "average_distortion_after_the_longer_solve": float(st2.average_distortions[-1]),
            "component_ms": {"average_distortion fwd+bwd (ring kernel + combine)": t_eval,
                             "tangent projection": t_tan, "retraction": t_ret},
        }
        del mde
    main = records["Centered"]
    return {
        "metric": "seconds/iteration of MDE.embed(), n=1M |E|=50M d=2 Log1p", "value": main["s_per_iter"], "unit": "s/iter",
        "n_gpus": 1, "steps": main["iterations"], "warmup": 5, "ms_per_step": main["ms_per_iter"], "higher_is_better": False,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3] / SURVEY 8d config 4a as an embed(): n=%d, |E|=%d %s edges "
                               "(out-degree 50), d=%d, %s, L-BFGS memory 10, "
                               "X0 = constraint.initialization; value = the Centered solve"
                               % (n, p, {"uniform": "uniform-random", "clusters": "planted-cluster (1000 items per cluster, 2/3 of the links inside)",
                                         "hub": "uniform-random + one hub", "powerlaw": "preferential-attachment"}[args.graph], d,
                                  "PushAndPull(Log1p(1.5), Log(1)): weights {1,2} inside the clusters, -1 across" if args.graph == "clusters"
                                  else "penalties.Log1p(1.5), weights in {1,2}"),
                   "parallelism": "single GPU" if args.emulate_world <= 1 else
                                  "rank 0 of a %d-way ROW-SHARDED solve (optim._ShardedEngine), kernels + host loop only" % args.emulate_world,
                   "edges_per_s_per_iter": p / main["s_per_iter"],
                   "embed": records,
                   "problem_build_s": {"edge plan (validate, sort, CSR)": t_plan, "ring layout": t_layout,
                                       "parameters (expand + codebook)": t_bind + t_struct,
                                       "total": t_plan + t_layout + t_bind + t_struct}},
    }


# ------------------------------------------------------------------------------------------ configs 2, 3
def scale_free_edges(n, m, seed):
    """Barabasi-Albert-style preferential attachment (m links per new node), numpy only."""
    rng = np.random.default_rng(seed)
    targets = list(range(m))
    repeated = []
    src, dst = [], []
    for v in range(m, n):
        for t in set(targets):
            src.append(v)
            dst.append(t)
        repeated.extend(targets)
        repeated.extend([v] * m)
        idx = rng.integers(0, len(repeated), m)
        targets = [repeated[i] for i in idx]
    return np.stack([np.array(dst), np.array(src)], axis=1)


def run_embed_config(args, device, which):
    import functools
    import pymde_amd
    t0 = time.perf_counter()
    if which == 2:
        # MNIST stand-in: 70k points of a 10-component Gaussian mixture in R^784
        g = torch.Generator(device=device)
        g.manual_seed(0)
        n, nf = 70_000, 784
        centers = 4.0 * torch.randn((10, nf), device=device, generator=g)
        data = centers[torch.randint(0, 10, (n,), device=device, generator=g)] + \
            torch.randn((n, nf), device=device, generator=g)
        mde = pymde_amd.preserve_neighbors(data, embedding_dim=2, n_neighbors=15,
                                           attractive_penalty=pymde_amd.penalties.Log1p,
                                           repulsive_penalty=pymde_amd.penalties.LogRatio,
                                           constraint=pymde_amd.Standardized(), device=device)
        label = ("BASELINE configs[1] stand-in: 70k x 784 Gaussian mixture, preserve_neighbors(k=15, Log1p / "
                 "LogRatio, Standardized), d=2")
        functor = "FnPushPull<Log1p(1.5), LogRatio(2)> (compile-time pair)"
    else:
        # Google-Scholar stand-in: 40k-node scale-free graph, sampled shortest-path distances, Huber loss
        e = scale_free_edges(40_000, 5, 0)
        graph = pymde_amd.Graph.from_edges(torch.tensor(e), device=device)
        mde = pymde_amd.preserve_distances(graph, embedding_dim=2,
                                           loss=functools.partial(pymde_amd.losses.Huber, threshold=1.0),
                                           max_distances=5e7, device=device)
        label = ("BASELINE configs[2] stand-in: 40k-node scale-free graph (5 links per node), "
                 "preserve_distances(Huber(1.0), max_distances=5e7), d=2")
        functor = "FnSingle<L_HUBER> (compile-time), CSR and LDS-ring kernels"
    torch.cuda.synchronize(device)
    build_s = time.perf_counter() - t0
    iters = 100
    mde.embed(max_iter=5)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    mde.embed(max_iter=iters, eps=0.0)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    n_it = max(int(mde.solve_stats.iterations), 1)
    p = int(mde.edges.shape[0])
    layout = int(mde._binding().struct(2).layout)
    return {
        "metric": "seconds/iteration of MDE.embed()", "value": dt / n_it, "unit": "s/iter", "n_gpus": 1,
        "steps": n_it, "warmup": 5, "ms_per_step": 1e3 * dt / n_it, "higher_is_better": False,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": label, "n_items": int(mde.n_items), "edges": p, "problem_build_s": build_s,
                   "edges_per_s_per_iter": p * n_it / dt, "kernel_layout": "LDS ring" if layout == 1 else "CSR",
                   "parameter_stream": mde._binding().stream_kind,
                   "distinct_parameter_values": int(torch.unique(next(iter(mde.distortion_function.buffers()))).numel())
                   if hasattr(mde.distortion_function, "buffers") and list(mde.distortion_function.buffers()) else None,
                   "functor": functor, "final_average_distortion": float(mde.value)},
    }


def self_launch(args):
    """--gpus N > 1 outside a launcher: start the N ranks of this script (one per GPU, rendezvous on
    127.0.0.1); rank 0 prints the JSON line."""
    import socket
    import subprocess
    if args.backend == "nccl":
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit("bench.py --gpus %d needs %d visible GPUs for the RCCL ranks (one process per GPU); "
                             "this host has %d" % (args.gpus, args.gpus, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    # one process per rank with the environment torch.distributed.run would give it (env:// rendezvous
    # on 127.0.0.1); the script's own arguments go through untouched
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        for pr in procs:
            code = pr.wait()
            if code != 0 and rc == 0:
                rc = code
                for other in procs:       # a rank died: do not leave the others in a collective
                    if other.poll() is None:
                        other.terminate()
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--blocks", type=int, default=11, help="repetitions of the timed region (the median is reported; odd: the median is a measured block)")
    ap.add_argument("--config", type=int, default=4, choices=(2, 3, 4, 5))
    ap.add_argument("--variant", default="4a", choices=("4a", "4b"),
                    help="config 4 only: 4a Log1p (the headline), 4b PushAndPull(Log1p, Log) with 1/3 repulsive edges")
    ap.add_argument("--function", default="log1p",
                    choices=("log1p", "quadratic", "linear", "cubic", "huber", "log", "log1p2", "l_huber", "l_quadratic",
                             "logistic", "power", "power15", "sigmoid", "hinge", "invpower", "logratio", "l_cubic", "l_power",
                             "l_logistic", "l_fractional", "l_softfractional", "runtime"),
                    help="config 4 only: another distortion function on the same graph (secondary records)")
    ap.add_argument("--embed", action="store_true", help="configs 4 and 5: a full embed() at that shape (s/iter)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--survey-seed", action="store_true",
                    help="config 4: SURVEY 8d's exact numpy default_rng(0) / torch.manual_seed(0) workload, built on the "
                         "host and uploaded; prints its loss next to the oracle's (the default line keeps the device generator)")
    ap.add_argument("--no-codebook", action="store_true",
                    help="stream the weights as fp32 (8 B/half-edge) even though they take 2 values")
    ap.add_argument("--n", type=int, default=N_ITEMS)
    ap.add_argument("--degree", type=int, default=OUT_DEGREE, help="out-degree of the synthetic graph (secondary records)")
    ap.add_argument("--dim", type=int, default=DIM, choices=(1, 2, 3, 4),
                    help="config 4 only: embedding dimension (secondary records; the headline is d = 2)")
    ap.add_argument("--graph", default="uniform", choices=("uniform", "hub", "powerlaw", "clusters"),
                    help="config 4 only: the survey's uniform-random graph, the same with one hub of degree 5e5, or a "
                         "preferential-attachment graph (secondary records)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--exchange-only", action="store_true",
                    help="no kernel: run only the [grad|loss] exchange of an N-rank job (CPU / gloo check of the launcher)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="single process: time only the kernel of rank 0 of a W-way shard (no collective)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if args.no_codebook:
        os.environ["MDE_CODEBOOK"] = "0"
        os.environ["MDE_BYTE_STREAM"] = "0"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one rank per GPU" % (args.gpus, world))
    if args.exchange_only:
        if world < 2:
            raise SystemExit("--exchange-only needs --gpus N > 1")
        dist.init_process_group(backend=args.backend)
        out = run_exchange_only(args, world, rank)
        if out is not None:
            print(json.dumps(out))
        dist.destroy_process_group()
        return
    ndev = torch.cuda.device_count()
    if world > 1:
        if args.backend == "nccl" and ndev < world:
            raise SystemExit("%d ranks but %d visible GPUs: one process per GPU" % (world, ndev))
        torch.cuda.set_device(local_rank % ndev)
        dist.init_process_group(backend=args.backend)
    device = torch.device("cuda", (local_rank % ndev) if world > 1 else 0)
    torch.cuda.set_device(device)

    if args.config == 4 and args.embed:
        if world > 1:
            raise SystemExit("--config 4 --embed is a single-GPU record")
        out = run_config4_embed(args, device)
    elif args.config == 4:
        out = run_config4(args, world, rank, device)
    elif args.config == 5 and (world > 1 or args.emulate_world > 1) and not args.embed:
        out = run_config5_sharded(args, world, rank, device)
    elif world > 1:
        raise SystemExit("--config %d is a single-GPU record" % args.config)
    elif args.config == 5:
        out = run_config5_embed(args, device) if args.embed else run_config5(args, device)
    else:
        out = run_embed_config(args, device, args.config)
    if out is not None:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
