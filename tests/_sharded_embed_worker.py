"""Worker of test_gpu_sharded_embed: two ranks (gloo, both on cuda:0) solve one sharded problem with
ShardedMDE.embed() and compare with the single-process solve of the same problem."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pymde_amd  # noqa: E402
from pymde_amd import distributed  # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank = dist.get_rank()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    rng = np.random.default_rng(5)
    for n, p, panel in ((4000, 40000, "0"), (120000, 700000, "1")):
        os.environ["MDE_PANEL"] = panel
        i = rng.integers(0, n, p)
        j = (i + 1 + rng.integers(0, n - 1, p)) % n
        e = np.unique(np.sort(np.stack([i, j], 1), 1), axis=0)
        w = rng.choice(np.array([-1.0, 1.0, 2.0], dtype=np.float32), size=len(e), p=[0.3, 0.4, 0.3])
        X0 = rng.standard_normal((n, 2)).astype(np.float32)
        edges = torch.tensor(e, device=dev)
        for cname, make in (("centered", pymde_amd.Centered), ("standardized", pymde_amd.Standardized)):
            c = make()
            f = pymde_amd.penalties.PushAndPull(torch.tensor(w, device=dev))
            x0 = c.project_onto_constraint(torch.tensor(X0, device=dev))
            single = pymde_amd.MDE(n, 2, edges, f, constraint=c, device=dev)
            Xs = single.embed(X=x0.clone(), max_iter=15).clone()
            sharded = distributed.ShardedMDE(n, 2, edges, f, constraint=make(), device=dev)
            Xd = sharded.embed(X=x0.clone(), max_iter=15)
            # every rank holds the same iterate, and it is the single-process one
            ref = Xd.clone()
            dist.broadcast(ref, src=0)
            assert torch.equal(ref, Xd), (rank, cname, "ranks disagree")
            assert torch.equal(Xd, Xs), (rank, cname, n, float((Xd - Xs).abs().max()))
            # (the loss is the fp32 sum of two shard losses there, one double-accumulated sum here)
            np.testing.assert_allclose(sharded.solve_stats.average_distortions,
                                       single.solve_stats.average_distortions, rtol=2e-6)
    dist.barrier()
    if rank == 0:
        print("sharded embed ok")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
