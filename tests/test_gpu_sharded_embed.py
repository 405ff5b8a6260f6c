"""Two ranks on one GPU (gloo): a sharded solve reproduces the single-process solve (bit for bit on the CSR kernel)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_sharded_embed_equals_single_process():
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(here, "_sharded_embed_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    tb = "\n".join(ln for ln in out.stderr.splitlines() if ln.startswith("[rank0]"))
    assert out.returncode == 0 and "sharded embed ok" in out.stdout, (out.stdout[-400:], tb[-3000:] or out.stderr[-3000:])
