"""Benchmark of the north-star hot path: one average_distortion forward+backward per step.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload = BASELINE.json configs[3] ("Synthetic scale"): n = 1M items, |E| = 50M uniform-random
edges (out-degree 50), d = 2, penalties.Log1p(exponent 1.5), weights in {1, 2}; synthetic,
seeded, generated on the device.  A step is one fused evaluation producing the
scalar loss and the full [n, d] gradient on every rank (multi-GPU: incl. the RCCL all-reduce of
[grad | loss]).  The problem size is fixed as N grows (strong scaling, as the metric is stated).

Prints ONE JSON line: metric/value = edges/s/iter, plus
  roofline     -- algorithmic bytes (12.32 B/edge) / fused-kernel duration, timed
                  with HIP events on the launching stream, against the 8 TB/s HBM peak
  cpu_baseline -- the OpenMP CPU oracle (a port of the reference's algorithm) timed on this
                  host's cores on the same workload (N = 1 only)
"""
import argparse
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
# HBM-side bytes per launch of the config-4 kernel from the rocprofv3 PMC passes committed in
# profiles/r01_pmc_summary.md (separate --pmc runs; FETCH_SIZE doubled for wide coalesced reads as
# MI355X_MICROARCH.md prescribes, + WRITE_SIZE).  bench.py cannot collect PMC counters itself.
PMC_TRAFFIC_BYTES = {(1, True): 2 * 244.6e6 + 7.8e6,   # k_fused_panel, codebook stream (4 B/half-edge)
                     (1, False): 2 * 450.2e6 + 7.8e6,  # k_fused_panel  (LDS column panels, fp32 parameter stream)
                     (0, False): 4.45e9 + 9.4e6}       # k_fused_small  (CSR; narrow gather line fills, no doubling)


def make_workload(device, n=N_ITEMS, deg=OUT_DEGREE, d=DIM):
    """SURVEY 8d config 4a, generated on the device from fixed seeds."""
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
                      % (len(times), best_t, max_threads)}, E


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
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

    import pymde_amd
    from pymde_amd import distributed
    from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate

    n, d = args.n, DIM
    edges, w, X = make_workload(device, n=n)
    p = edges.shape[0]
    f = pymde_amd.penalties.Log1p(w)
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
        if world > 1 and exchange.mode != "all_gather":
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
    gpu_loss = float(loss.item())

    # dominant kernel: the fused gather/scatter kernel, timed per launch with HIP events on the
    # stream it is launched on (torch's current stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(max(args.steps, 1))]
    for a, b in ev:
        a.record()
        fused_evaluate(binding, X, grad, loss)
        b.record()
    torch.cuda.synchronize(device)
    k_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    edges_local = plan.half_edges / 2.0
    alg_bytes = ALG_BYTES_PER_EDGE * edges_local + 2.0 * 4.0 * d * (plan.row_hi - plan.row_lo)
    achieved = alg_bytes / (k_ms * 1e-3)

    if rank == 0:
        out = {
            "metric": "edges/sec/iter (avg_distortion fwd+bwd), n=1M |E|=50M d=2",
            "value": p * args.steps / elapsed, "unit": "edges/s/iter",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: n=%d, |E|=%d uniform-random edges, d=2, "
                                   "penalties.Log1p(1.5), weights in {1,2}" % (n, p),
                       "parallelism": ("vertex-range shards x%d + %s of [grad|loss]" % (world, exchange.mode))
                       if world > 1 else "single GPU",
                       "parameter_stream": ("codebook: 2 distinct weights ride in the packed half-edge word "
                                            "(4 B/half-edge)" if binding.codebook
                                            else "fp32 weight per half-edge (8 B/half-edge)"),
                       "loss": gpu_loss},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK_BPS / 1e9,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_BPS,
                         "traffic": (PMC_TRAFFIC_BYTES.get((int(binding.struct(d).layout), bool(binding.codebook)))
                                     if (world == 1 and n == N_ITEMS) else None),
                         "traffic_source": "profiles/r01_pmc_summary.md (rocprofv3 --pmc, bytes/launch)",
                         "kernel": ("k_fused_panel<2,Log1p> (LDS column panels, loss reduced in the same launch)"
                                    if binding.struct(d).layout == 1
                                    else "k_fused_small<2,G,Log1p> (CSR) + 1-block loss finalize"),
                         "kernel_ms": k_ms, "alg_bytes_per_launch": alg_bytes},
        }
        if world == 1 and not args.no_cpu_baseline:
            cb, cpu_loss = cpu_baseline(edges, w, X, p)
            out["cpu_baseline"] = cb
            out["config"]["oracle_loss"] = cpu_loss
            assert abs(cpu_loss - gpu_loss) <= 1e-5 * abs(cpu_loss), (cpu_loss, gpu_loss)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
