"""Time the REFERENCE (cvxgrp/pymde, torch CPU) on BASELINE config 4 in the build container
(/root/reference exists only there): one average_distortion forward+backward, min of 3.

    python tools/ref_cpu_time.py [n] [deg]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
import make_golden  # noqa: E402

pymde = make_golden.import_reference()
import torch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 50
torch.set_num_threads(os.cpu_count())
rng = np.random.default_rng(0)
p = n * deg
src = np.repeat(np.arange(n), deg)
dst = rng.integers(0, n - 1, p)
dst += dst >= src
edges = torch.tensor(np.stack([np.minimum(src, dst), np.maximum(src, dst)], 1))
w = torch.tensor(1.0 + (rng.random(p) < 0.3), dtype=torch.float32)
torch.manual_seed(0)
X = torch.randn(n, 2)
X -= X.mean(0)
mde = pymde.MDE(n, 2, edges, pymde.penalties.Log1p(w))
times = []
for rep in range(4):
    Xt = X.clone().requires_grad_(True)
    t0 = time.perf_counter()
    L = mde.average_distortion(Xt)
    L.backward()
    times.append(time.perf_counter() - t0)
print("reference torch-CPU, %d threads (%s): n=%d p=%d  min %.3f s  median %.3f s  => %.3e edges/s/iter"
      % (torch.get_num_threads(), os.popen("grep -m1 'model name' /proc/cpuinfo").read().split(":")[-1].strip(),
         n, p, min(times[1:]), float(np.median(times[1:])), p / min(times[1:])))
