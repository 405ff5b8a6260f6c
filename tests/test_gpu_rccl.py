"""The exchange of a sharded problem on RCCL itself (backend "nccl"), one rank on one GPU."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_exchange_and_sharded_solve_on_rccl_single_rank():
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(here, "_rccl_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "rccl single-rank ok" in out.stdout, (out.stdout[-800:], out.stderr[-3000:])


def test_rccl_two_ranks():
    """Two RCCL ranks on two GPUs through bench.py's own launcher (lights up on a multi-GPU node):
    the line says n_gpus = 2, names the exchange and carries a kernel time per rank."""
    import json
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                          "--blocks", "3", "--n", "400000", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["exchange"] in ("all_gather", "all_reduce")
    assert len(rec["config"]["kernel_ms_per_rank"]) == 2 and min(rec["config"]["kernel_ms_per_rank"]) > 0
