"""Benchmark of the north-star hot path: one average_distortion forward+backward per step.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Default workload = BASELINE.json configs[3] ("Synthetic scale"): n = 1M items,
|E| = 50M uniform-random edges (out-degree 50), d = 2, penalties.Log1p(exponent 1.5), weights in
{1, 2}; synthetic, seeded, generated on the device.  A step is one fused evaluation producing the
scalar loss and the full [n, d] gradient on every rank (multi-GPU: incl. the exchange of
[grad | loss]).  The problem size is fixed as N grows (strong scaling, as the metric is stated).

Prints ONE JSON line: metric/value = edges/s/iter, plus
  roofline     -- algorithmic bytes (12.32 B/edge) / fused-kernel duration (median of
                  per-launch HIP-event times on the launching stream), against the 8 TB/s HBM peak;
                  `traffic` = HBM bytes per launch from the rocprofv3 PMC passes of THIS kernel
                  source (profiles/r02_pmc_traffic.json, ignored when the source has changed since)
  cpu_baseline -- the OpenMP CPU oracle (a port of the reference's algorithm) timed on this
                  host's cores on the same workload (N = 1 only)

Other workloads (secondary records; same JSON shape):
  --config 5   BASELINE configs[4]: n = 500k, |E| = 20M, d = 128 (Log1p and Quadratic; the
               Standardized projection kernels at that shape are timed too)
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
KERNEL_SOURCES = ["pymde_amd/csrc/mde_ring.hip", "pymde_amd/csrc/mde_functions.h"]


def source_sha():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(key):
    """HBM bytes per launch recorded by the PMC passes (tools/pmc_traffic.sh), or None when the
    record is missing or belongs to another version of the kernel source."""
    path = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")
    try:
        rec = json.load(open(path))
    except Exception:
        return None, None
    if rec.get("source_sha") != source_sha():
        return None, "profiles/r02_pmc_traffic.json is from another kernel source (stale): ignored"
    return rec.get("bytes_per_launch", {}).get(key), "profiles/r02_pmc_traffic.json (rocprofv3 --pmc, bytes/launch)"


def make_workload(device, n=N_ITEMS, deg=OUT_DEGREE, d=DIM):
    """SURVEY 8d config 4a, generated on the device from fixed seeds (torch's device generator
    instead of numpy's default_rng(0): same distribution, different stream)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(0)
    p = n * deg
    src = torch.arange(n, device=device, dtype=torch.int64).repeat_interleave(deg)
    dst = torch.randint(0, n - 1, (p,), device=device, dtype=torch.int64, generator=gen)
    dst += (dst >= src).to(torch.int64)
    edges = torch.stack([torch.minimum(src, dst), torch.maximum(src, dst)], dim=1).contiguous()
    w = 1.0 + (torch.rand(p, device=device, generator=gen) < 0.3).to(torch.float32)
    gen.manual_seed(0)
    X = torch.randn((n, d), device=device, generator=gen)
    X -= X.mean(0)
    return edges, w, X.contiguous()


def cpu_baseline(edges, w, X, p):
    """Time the CPU oracle (OpenMP port of the reference algorithm) on the host cores.

    The thread count is calibrated first on a 10 % sample (the oracle's per-thread gradient
    accumulators make very wide runs slower), then the full workload is timed (bounded to
    ~25 s)."""
    from oracle import oracle
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
    while len(times) < 3 and (time.perf_counter() - t_start) < 25.0:
        t0 = time.perf_counter()
        E, _ = oracle.average_distortion(e, Xn, fd)
        times.append(time.perf_counter() - t0)
    return {"value": p / min(times), "unit": "edges/s/iter", "cores": int(best_t), "kind": "port",
            "sample": "full workload (n=1M, |E|=50M, d=2, Log1p), min of %d fwd+bwd evaluations of "
                      "oracle/mde_oracle.c with OpenMP on %d of %d host threads (best of a thread sweep)"
                      % (len(times), best_t, max_threads),
            "reference_torch_cpu": REFERENCE_TORCH_CPU}, E


def time_launches(fn, count, device):
    """Median / mean duration (ms) of `count` launches of fn, each bracketed by HIP events on
    the current stream."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(count)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize(device)
    t = np.array([a.elapsed_time(b) for a, b in ev])
    return float(np.median(t)), float(t.mean())


# ------------------------------------------------------------------------------------------ config 4
def run_config4(args, world, rank, device):
    import pymde_amd
    from pymde_amd import distributed
    from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate

    n, d = args.n, DIM
    edges, w, X = make_workload(device, n=n)
    p = edges.shape[0]
    f = pymde_amd.penalties.Log1p(w)
    bounds = None
    if world > 1 or args.emulate_world > 1:
        W = world if world > 1 else args.emulate_world
        bounds = distributed.shard_bounds(n, edges, W)
        lo, hi = distributed.shard_range(bounds, rank)
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
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
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
    edges_local = plan.half_edges / 2.0
    alg_bytes = ALG_BYTES_PER_EDGE * edges_local + 2.0 * 4.0 * d * (plan.row_hi - plan.row_lo)
    achieved = alg_bytes / (k_ms * 1e-3)
    layout = int(binding.struct(d).layout)
    kernel = ("k_fused_ring<2,Log1p,%s> (LDS-resident rows + LDS-DMA chunk ring, loss reduced in the same launch)"
              % ("codebook" if binding.codebook else "fp32 stream")) if layout == 1 \
        else "k_fused_small<2,G,Log1p> (CSR) + 1-block loss finalize"
    traffic, traffic_src = (None, None)
    if world == 1 and args.emulate_world <= 1 and n == N_ITEMS:
        traffic, traffic_src = pmc_traffic("ring_codebook" if binding.codebook else "ring_fp32" if layout == 1 else "csr")

    if rank != 0:
        return None
    out = {
        "metric": "edges/sec/iter (avg_distortion fwd+bwd), n=1M |E|=50M d=2",
        "value": p * args.steps / elapsed, "unit": "edges/s/iter",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "ms_per_step_median_events": float(np.median(step_ms)),
        "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3] / SURVEY 8d config 4a: n=%d, |E|=%d uniform-random edges "
                               "(out-degree 50), d=2, penalties.Log1p(1.5), weights in {1,2}; seeded on the "
                               "device with torch.Generator(0) (the survey's recipe uses numpy default_rng(0): "
                               "same distribution, another stream)" % (n, p),
                   "parallelism": ("vertex-range shards x%d + %s of [grad|loss]" % (world, exchange.mode))
                   if world > 1 else ("single GPU" if args.emulate_world <= 1 else
                                      "rank 0 of a %d-way shard, kernel only (emulation, no collective)" % args.emulate_world),
                   "parameter_stream": ("codebook: 2 distinct weights ride in the packed half-edge word "
                                        "(4 B/half-edge)" if binding.codebook
                                        else "fp32 weight per half-edge (8 B/half-edge)"),
                   "loss": gpu_loss},
        "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK_BPS / 1e9,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_BPS, "traffic": traffic,
                     "traffic_source": traffic_src, "kernel": kernel, "kernel_ms": k_ms,
                     "kernel_ms_mean": k_mean, "alg_bytes_per_launch": alg_bytes},
    }
    if world == 1 and binding.codebook and not args.no_codebook:
        # secondary: the general case (continuous per-edge parameters, configs 2 / 3) streams an
        # fp32 parameter per half-edge
        os.environ["MDE_CODEBOOK"] = "0"
        b2 = Binding(plan, pymde_amd.penalties.Log1p(w.clone()))
        fused_evaluate(b2, X, grad, loss)
        k2, _ = time_launches(lambda: fused_evaluate(b2, X, grad, loss), max(args.steps, 1), device)
        os.environ.pop("MDE_CODEBOOK")
        out["config"]["fp32_parameter_stream"] = {
            "kernel_ms": k2, "value": p / (k2 * 1e-3), "unit": "edges/s/iter (kernel time)",
            "roofline_frac": alg_bytes / (k2 * 1e-3) / HBM_PEAK_BPS}
    if world == 1 and args.emulate_world <= 1 and not args.no_cpu_baseline:
        cb, cpu_loss = cpu_baseline(edges, w, X, p)
        out["cpu_baseline"] = cb
        out["config"]["oracle_loss"] = cpu_loss
        assert abs(cpu_loss - gpu_loss) <= 1e-5 * abs(cpu_loss), (cpu_loss, gpu_loss)
    return out


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
                     "peak": HBM_PEAK_BPS / 1e9, "unit": "GB/s", "frac": k["roofline_frac"], "traffic": None,
                     "kernel": "k_fused_wide4<32,1> (CSR, one half wave per 512-byte row, 16-byte loads)",
                     "kernel_ms": k["kernel_ms"], "alg_bytes_per_launch": alg_bytes,
                     "note": "the 37.6 B/edge figure assumes every row is read once; a uniform-random graph "
                             "at d = 128 has no reuse to exploit (each half-edge needs its own 512-byte row: "
                             "%.1f GB per evaluation), so the attainable bound is the random-row gather rate "
                             "(tools/rowprobe: 7.4 TB/s from the 256 MB table) -> see row_gather_TBps"
                             % (gather_bytes / 1e9)},
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
        functor = "FnSingle<L_HUBER> on the CSR kernel, FnRuntime on the LDS-ring kernel"
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
                   "functor": functor, "final_average_distortion": float(mde.value)},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=4, choices=(2, 3, 4, 5))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-codebook", action="store_true",
                    help="stream the weights as fp32 (8 B/half-edge) even though they take 2 values")
    ap.add_argument("--n", type=int, default=N_ITEMS)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="single process: time only the kernel of rank 0 of a W-way shard (no collective)")
    args = ap.parse_args()

    if args.no_codebook:
        os.environ["MDE_CODEBOOK"] = "0"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    if world > 1:
        torch.cuda.set_device(local_rank % ndev)
        dist.init_process_group(backend=args.backend)
    device = torch.device("cuda", (local_rank % ndev) if world > 1 else 0)
    torch.cuda.set_device(device)

    if args.config == 4:
        out = run_config4(args, world, rank, device)
    elif world > 1:
        raise SystemExit("--config %d is a single-GPU record" % args.config)
    elif args.config == 5:
        out = run_config5(args, device)
    else:
        out = run_embed_config(args, device, args.config)
    if out is not None:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
