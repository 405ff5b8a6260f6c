"""Worker of test_gpu_rccl: ONE rank on cuda:0 with backend "nccl" (= RCCL on ROCm).  A single-GPU
box cannot run a second rank (RCCL refuses two ranks on one device), but it can run the calls
themselves: communicator set-up, the in-place all-gather form GradExchange uses (output = the
buffer, input = this rank's rows inside it), the two asynchronous handles, the all-reduce, and
their ordering against the solver's kernels on the current stream."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pymde_amd  # noqa: E402
from pymde_amd import distributed  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    rng = np.random.default_rng(11)
    n, p, d = 30000, 200000, 2
    i = rng.integers(0, n, p)
    j = (i + 1 + rng.integers(0, n - 1, p)) % n
    e = np.unique(np.sort(np.stack([i, j], 1), 1), axis=0)
    w = rng.choice(np.array([-1.0, 1.0, 2.0], dtype=np.float32), size=len(e))
    edges = torch.tensor(e, device=dev)
    X0 = torch.tensor(rng.standard_normal((n, d)).astype(np.float32), device=dev)

    # the exchange alone, forced to run in a world of one: both forms must leave [grad | loss] as it is
    single = pymde_amd.MDE(n, d, edges, pymde_amd.penalties.PushAndPull(torch.tensor(w, device=dev)),
                           constraint=pymde_amd.Centered(), device=dev)
    x = X0.clone().requires_grad_(True)
    E = single.average_distortion(x)
    E.backward()
    buf = torch.cat([x.grad.reshape(-1), E.detach().reshape(1)])
    ex = distributed.GradExchange(n, d, [0, n], 0, 1, force=True)
    out = ex(buf.clone())
    assert ex.mode == "all_gather", ex.mode  # verified against the all-reduce on first use
    assert torch.equal(out, buf)
    for _ in range(3):  # steady state: in place on the caller's buffer
        b = buf.clone()
        assert ex(b) is b and torch.equal(b, buf)
    red = distributed.GradExchange(n, d, [0, n], 0, 1, force=True)
    red.mode = "all_reduce"
    assert torch.equal(red(buf.clone()), buf)

    # a sharded solve whose every evaluation goes through the RCCL exchange == the plain solve
    Xs = single.embed(X=X0.clone(), max_iter=20).clone()
    for slices in (1, 4):
        sharded = distributed.ShardedMDE(n, d, edges, pymde_amd.penalties.PushAndPull(torch.tensor(w, device=dev)),
                                         constraint=pymde_amd.Centered(), device=dev, slices=slices, force_exchange=True)
        assert sharded._layout.slices == slices and len(sharded._reducer.plans) == slices
        if slices == 1:
            # the ROW-SHARDED solver (round 6) with its collectives on RCCL: the all-reduce of the history update's
            # inner products, the in-place all-gather of the trial point's rows, the all-gather of the per-rank records
            from pymde_amd import optim
            assert optim._sharded_solver_args(sharded.average_distortion, sharded.constraint) is not None
            Xr = sharded.embed(X=X0.clone(), max_iter=20)
            np.testing.assert_allclose(sharded.solve_stats.average_distortions[:3],
                                       single.solve_stats.average_distortions[:3], rtol=1e-5)
            assert abs(sharded.value - single.value) <= 2e-2 * abs(single.value)
            assert float(Xr.double().mean(0).abs().max()) < 1e-4
            os.environ["MDE_SHARD_SOLVER"] = "0"   # ... and the replicated optimiser behind the gradient exchange
        Xd = sharded.embed(X=X0.clone(), max_iter=20)
        os.environ.pop("MDE_SHARD_SOLVER", None)
        # (one slice: the in-place all-gather of GradExchange; four: the sliced gathers on the side stream, each
        # behind its slice's kernel, with the loss share as a one-float all-reduce)
        assert sharded._reducer.mode == "all_gather", (slices, sharded._reducer.mode)
        if slices == 1:
            assert torch.equal(Xs, Xd), float((Xs - Xd).abs().max())
            np.testing.assert_array_equal(sharded.solve_stats.average_distortions, single.solve_stats.average_distortions)
        else:
            # (the loss is the fp32 sum of four slice shares: last bits; the gradient is bit-equal)
            xs = X0.clone().requires_grad_(True)
            single.average_distortion(xs).backward()
            xd = X0.clone().requires_grad_(True)
            sharded.average_distortion(xd).backward()
            assert torch.equal(xs.grad, xd.grad)
            np.testing.assert_allclose(sharded.solve_stats.average_distortions[:3],
                                       single.solve_stats.average_distortions[:3], rtol=1e-5)
    torch.cuda.synchronize()
    print("rccl single-rank ok")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
