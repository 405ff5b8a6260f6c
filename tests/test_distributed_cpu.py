"""N > 1 path on CPU: two gloo ranks each own a vertex range; their [grad | loss] buffers,
all-reduced by the product's reducer, equal the unsharded oracle result (world_size 2)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    from pymde_amd import distributed
    rng = np.random.default_rng(5)
    n, p, d = 200, 1500, 2
    pairs = np.stack(np.triu_indices(n, 1), axis=1)
    edges = pairs[np.sort(rng.choice(len(pairs), p, replace=False))]
    w = rng.uniform(0.5, 2.0, p).astype(np.float32)
    X = rng.standard_normal((n, d)).astype(np.float32)
    fd = oracle.func("LOG1P", w, None, (1.5,))
    E_full, grad_full = oracle.average_distortion(edges, X, fd)
    # owner-computes share of this rank: its gradient rows are final, its loss share counts
    # every incident half-edge with weight 1/2
    bounds = oracle.shard_bounds(n, edges, world)
    lo, hi = distributed.shard_range(bounds, rank)
    f_edge = oracle.distortions(oracle.distances(edges, X), fd).astype(np.float64)
    own_i = (edges[:, 0] >= lo) & (edges[:, 0] < hi)
    own_j = (edges[:, 1] >= lo) & (edges[:, 1] < hi)
    share = (0.5 * f_edge * own_i + 0.5 * f_edge * own_j).sum() / p
    buf = torch.zeros(n * d + 1, dtype=torch.float32)
    buf[lo * d:hi * d] = torch.from_numpy(grad_full[lo:hi].reshape(-1))
    buf[-1] = share
    distributed.all_reduce_grad_loss(buf)
    np.testing.assert_allclose(buf[:-1].numpy().reshape(n, d), grad_full, rtol=0, atol=0)
    # the exchange object used by ShardedMDE / bench.py: with equal row counts it may switch to an
    # all-gather after verifying it against the all-reduce; either way the result is the same
    ub = [0, n // 2, n]
    ulo, uhi = distributed.shard_range(ub, rank)
    uown_i = (edges[:, 0] >= ulo) & (edges[:, 0] < uhi)
    uown_j = (edges[:, 1] >= ulo) & (edges[:, 1] < uhi)
    ex = distributed.GradExchange(n, d, ub, rank, world)
    for _ in range(3):
        b2 = torch.zeros(n * d + 1, dtype=torch.float32)
        b2[ulo * d:uhi * d] = torch.from_numpy(grad_full[ulo:uhi].reshape(-1))
        b2[-1] = (0.5 * f_edge * uown_i + 0.5 * f_edge * uown_j).sum() / p
        ex(b2)
        np.testing.assert_allclose(b2[:-1].numpy().reshape(n, d), grad_full, rtol=0, atol=0)
        assert abs(float(b2[-1]) - E_full) < 1e-5 * abs(E_full)
    assert ex.mode in ("all_gather", "all_reduce")
    ret["mode%d" % rank] = ex.mode
    assert abs(float(buf[-1]) - E_full) < 1e-5 * abs(E_full)
    # both ranks hold the identical reduced buffer
    gathered = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf)
    assert all(torch.equal(gathered[0], g) for g in gathered)
    # plan shards tile the full plan
    full = oracle.plan_csr(n, edges)
    mine = oracle.plan_csr(n, edges, lo, hi)
    np.testing.assert_array_equal(mine[1], full[1][full[0][lo]:full[0][hi]])
    ret[rank] = 1
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_matches_unsharded():
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert ret[0] == 1 and ret[1] == 1 and ret["mode0"] == ret["mode1"]


def _single_worker(port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    from pymde_amd import distributed
    n, d = 50, 3
    buf = torch.arange(n * d + 1, dtype=torch.float32) * 0.5 - 7.0
    # a world of one exchanges nothing unless forced ...
    ex = distributed.GradExchange(n, d, [0, n], 0, 1)
    assert ex(buf.clone()).equal(buf) and ex.mode is None
    # ... and forced (what the single-GPU RCCL test does) both forms are the identity
    forced = distributed.GradExchange(n, d, [0, n], 0, 1, force=True)
    out = forced(buf.clone())
    assert out.equal(buf) and forced.mode == "all_gather" and not forced.needs_zero()
    b = buf.clone()
    assert forced(b) is b and b.equal(buf)
    reduce_only = distributed.GradExchange(n, d, [0, n], 0, 1, force=True)
    reduce_only.mode = "all_reduce"
    assert reduce_only(buf.clone()).equal(buf) and reduce_only.needs_zero()
    ret["ok"] = 1
    dist.destroy_process_group()


def test_forced_exchange_in_a_world_of_one_is_the_identity():
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 31500 + (os.getpid() % 2000)
    proc = ctx.Process(target=_single_worker, args=(port, ret))
    proc.start()
    proc.join(180)
    assert proc.exitcode == 0 and ret.get("ok") == 1


def test_bench_self_launches_two_gloo_ranks():
    """`bench.py --gpus 2` outside a launcher starts its own two ranks; rank 0 prints ONE JSON line
    that says n_gpus = 2 and names the exchange (gloo, exchange-only: there is no GPU here)."""
    import json
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--n", "20000",
                          "--exchange-only", "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["config"]["exchange"] in ("all_gather", "all_reduce")
    assert "x2" in rec["config"]["parallelism"]


def test_bench_refuses_more_ranks_than_gpus():
    """With the RCCL backend, --gpus N on a host with fewer than N GPUs fails loudly (one process per GPU)."""
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "needs 64 visible GPUs" in (out.stderr + out.stdout)


def _sliced_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pymde_amd import distributed
    # the uniform slice-major layout: rank r owns range r of every slice; the slices' in-place all-gathers
    # (issued one by one, as the evaluator does behind each slice's kernel) assemble the full [n, d] rows
    n, d, K = 48, 128, 4
    lay = distributed.ShardLayout.uniform(n, world, K)
    assert lay.slices == K and lay.bounds is None
    m = n // (world * K)
    assert lay.ranges[rank] == [((k * world + rank) * m, (k * world + rank + 1) * m) for k in range(K)]
    covered = sorted(r for rr in lay.ranges for r in rr)
    assert covered[0][0] == 0 and covered[-1][1] == n and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    full = torch.arange(n * d, dtype=torch.float32) * 0.25 - 100.0
    buf = torch.full((n * d + 1,), float("nan"))
    for lo, hi in lay.ranges[rank]:
        buf[lo * d:hi * d] = full[lo * d:hi * d]
    buf[n * d] = float(rank + 1)
    handles = [distributed.gather_slice(buf, n, d, world, K, k, rank) for k in range(K)]
    h = dist.all_reduce(buf[n * d:n * d + 1], op=dist.ReduceOp.SUM, async_op=True)
    for w in handles:
        w.wait()
    h.wait()
    assert torch.equal(buf[:n * d], full)
    assert float(buf[n * d]) == float(sum(range(1, world + 1)))
    # K = 1 of the uniform layout carries bounds (the one-range exchange object takes it)
    one = distributed.ShardLayout.uniform(n, world, 1)
    assert one.bounds == [r * (n // world) for r in range(world + 1)] and one.slices == 1
    skew = distributed.ShardLayout.from_bounds(n, world, [0, 10, n])
    assert skew.slices == 0 and skew.ranges[1] == [(10, n)]
    # slices by size: one below 8 MB per rank, four above; the environment overrides
    assert distributed.default_slices(1_000_000, 2, 8) == 1 and distributed.default_slices(500_000, 128, 8) == 4
    os.environ["MDE_SHARD_SLICES"] = "2"
    assert distributed.default_slices(1_000_000, 2, 8) == 2
    os.environ.pop("MDE_SHARD_SLICES")
    ret[rank] = 1
    dist.destroy_process_group()


def test_sliced_in_place_all_gather_two_gloo_ranks():
    """The chunked exchange of the uniform slice-major layout (round 5), world_size 2 on gloo."""
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sliced_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert ret[0] == 1 and ret[1] == 1
