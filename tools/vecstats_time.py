# vecstats_time.py -- mde_vec_stats on config-5-sized vectors (3 x 256 MB): time per call and rate
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from pymde_amd import _lib, util
lib = _lib.load(); dev = torch.device('cuda'); st = _lib.stream_ptr(dev)
N = 500000 * 128
g, d, x = (torch.randn(N, device=dev) for _ in range(3))
board = torch.zeros(64, dtype=torch.float64, device=dev); work = util.work_buffer(dev, 128)
for _ in range(3):
    _lib.check(lib.mde_vec_stats(N, _lib.ptr(g), _lib.ptr(d), _lib.ptr(x), _lib.ptr(board), _lib.ptr(work), st))
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    _lib.check(lib.mde_vec_stats(N, _lib.ptr(g), _lib.ptr(d), _lib.ptr(x), _lib.ptr(board), _lib.ptr(work), st))
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
print("mde_vec_stats N=%d: %.3f ms, %.2f TB/s; g.g=%.6e (torch %.6e)" % (N, ms, 3 * 4 * N / ms / 1e9, float(board[1]), float((g.double() ** 2).sum())))
