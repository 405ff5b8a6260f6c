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
