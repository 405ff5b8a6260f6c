"""host cost of collective-call variants in a world of one (RCCL), microseconds per call, nothing else on the stream"""
import os, sys, time, torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29532")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1)
N = 2000000
buf = torch.zeros(N + 1, device=dev)
lossvec = torch.zeros(8, device=dev)
side = torch.cuda.Stream()
def t(f, reps=500):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / reps * 1e6
def v_async_pair():
    h1 = dist.all_gather_into_tensor(buf[:N], buf[:N], async_op=True)
    h2 = dist.all_reduce(buf[N:N + 1], async_op=True)
    h1.wait(); h2.wait()
def v_sync_pair():
    dist.all_gather_into_tensor(buf[:N], buf[:N])
    dist.all_reduce(buf[N:N + 1])
def v_gather_only():
    dist.all_gather_into_tensor(buf[:N], buf[:N])
def v_reduce_only():
    dist.all_reduce(buf[N:N + 1])
def v_allreduce_full():
    dist.all_reduce(buf)
def v_two_gathers_coalesced():
    with dist._coalescing_manager(device=dev, async_ops=False):
        dist.all_gather_into_tensor(buf[:N], buf[:N])
        dist.all_gather_into_tensor(lossvec[:1], lossvec[:1])
views = (buf[:N], buf[N:N + 1])
def v_sync_pair_preview():
    dist.all_gather_into_tensor(views[0], views[0])
    dist.all_reduce(views[1])
for name, f in (("async pair + waits (current)", v_async_pair), ("sync pair", v_sync_pair), ("sync pair, views made once", v_sync_pair_preview),
                ("all_gather only", v_gather_only), ("all_reduce(1 float) only", v_reduce_only), ("all_reduce(full buffer)", v_allreduce_full),
                ("two all_gathers, coalesced", v_two_gathers_coalesced)):
    try:
        print("%-34s %.1f us per call" % (name, t(f)))
    except Exception as e:
        print("%-34s failed: %s" % (name, str(e)[:120]))
dist.destroy_process_group()
