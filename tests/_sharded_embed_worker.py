"""Worker of test_gpu_sharded_embed: two ranks (gloo, both on cuda:0) solve one sharded problem with
ShardedMDE.embed() and compare with the single-process solve of the same problem, on the CSR kernel
(bit for bit) and on the LDS-ring kernel (column-group partial sums: to the last bits)."""
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
            sharded = distributed.ShardedMDE(n, 2, edges, f, constraint=make(), device=dev)
            assert single._binding().struct(2).layout == int(panel), "unexpected kernel layout"
            assert sharded._binding().struct(2).layout == int(panel), "unexpected kernel layout"
            # one evaluation: every rank holds the same loss and gradient, and they are the
            # single-process ones -- bit for bit on the CSR kernel (a row is summed in edge order
            # whoever owns it); on the LDS-ring kernel a shard sums a row per column group and adds
            # the group partials, so the last bits may differ
            xs = x0.clone().requires_grad_(True)
            Es = single.average_distortion(xs)
            Es.backward()
            xd = x0.clone().requires_grad_(True)
            Ed = sharded.average_distortion(xd)
            Ed.backward()
            ref = xd.grad.clone()
            dist.broadcast(ref, src=0)
            assert torch.equal(ref, xd.grad), (rank, cname, "ranks disagree on the gradient")
            if panel == "0":
                assert torch.equal(xd.grad, xs.grad), (rank, cname, float((xd.grad - xs.grad).abs().max()))
            else:
                scale = float(xs.grad.abs().max())
                assert float((xd.grad - xs.grad).abs().max()) <= 1e-5 * scale, (rank, cname)
            np.testing.assert_allclose(float(Ed), float(Es), rtol=2e-6)
            Xs = single.embed(X=x0.clone(), max_iter=15).clone()
            Xd = sharded.embed(X=x0.clone(), max_iter=15)
            # every rank holds the same iterate (the optimiser runs replicated on identical data)
            ref = Xd.clone()
            dist.broadcast(ref, src=0)
            assert torch.equal(ref, Xd), (rank, cname, "ranks disagree")
            if panel == "0":
                # ... and it is the single-process one
                assert torch.equal(Xd, Xs), (rank, cname, n, float((Xd - Xs).abs().max()))
                # (the loss is the fp32 sum of two shard losses there, one double-accumulated sum here)
                np.testing.assert_allclose(sharded.solve_stats.average_distortions,
                                           single.solve_stats.average_distortions, rtol=2e-6)
            else:
                # last-bit differences feed a line search that branches on rounding: compare the
                # start of the trajectory and the value reached
                np.testing.assert_allclose(sharded.solve_stats.average_distortions[:3],
                                           single.solve_stats.average_distortions[:3], rtol=1e-4)
                assert abs(sharded.value - single.value) <= 2e-2 * abs(single.value)
    dist.barrier()
    if rank == 0:
        print("sharded embed ok")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
