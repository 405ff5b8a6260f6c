"""exchange_host_cost.py -- is an 8-way shard of config 4 HOST-bound?  One process, RCCL in a world of one: the kernels of
rank 0 of an 8-way shard (38 us of GPU work per step) followed by the exchange's collectives (trivial device work in a
world of one, the full host-side cost of issuing them), back to back.  Prints ms per step with and without the exchange."""
import os, sys, time, torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import bench, pymde_amd
from pymde_amd import distributed
from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1)
n, d, W = 1000000, 2, int(os.environ.get("EMU_W", "8"))
edges, w, X = bench.make_workload(dev, n=n)
bounds = distributed.shard_bounds(n, edges, W)
lo, hi = distributed.shard_range(bounds, 0)
plan = EdgePlan(n, edges, lo, hi); b = Binding(plan, pymde_amd.penalties.Log1p(w))
buf = torch.zeros(n * d + 1, device=dev); grad, loss = buf[:n * d].view(n, d), buf[n * d:]
ex = distributed.GradExchange(n, d, [0, n], 0, 1, force=True)
def run(with_exchange, reps=300):
    def step():
        fused_evaluate(b, X, grad, loss)
        if with_exchange: ex(buf)
    for _ in range(20): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t2 - t0) / reps * 1e3, (t1 - t0) / reps * 1e3
for we in (False, True, False, True):
    total, host = run(we)
    print("exchange %-5s  %.4f ms per step (host enqueue alone %.4f ms), mode %s" % (we, total, host, ex.mode))
dist.destroy_process_group()
