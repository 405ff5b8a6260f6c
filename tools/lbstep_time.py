# lbstep_time.py -- mde_lbfgs_dev_step on config-5-sized vectors (N = 64M floats, history 10): ms per step
# in steady state and the HBM rate over the bytes it has to move (stage: 23 reads + 3 writes, combine: 21
# reads + 1 write of 256 MB each)
import ctypes, sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from pymde_amd import _lib, util
lib = _lib.load(); dev = torch.device('cuda'); st = _lib.stream_ptr(dev)
N, hist = 500000 * 128, 10
h = ctypes.c_void_p(); _lib.check(lib.mde_lbfgs_create(N, hist, ctypes.byref(h)))
_lib.check(lib.mde_lbfgs_dev_reset(h, st))
work = util.work_buffer(dev, 128); board = torch.zeros(64, dtype=torch.float64, device=dev)
g_prev = torch.randn(N, device=dev); d = -g_prev.clone(); g = torch.empty_like(g_prev)
def step():
    torch.mul(g_prev, 0.9, out=g); g.add_(0.01)
    _lib.check(lib.mde_lbfgs_dev_step(h, _lib.ptr(g), _lib.ptr(g_prev), _lib.ptr(d), 0.3, _lib.ptr(d), _lib.ptr(board), _lib.ptr(work), st))
for _ in range(13): step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ms = []
for _ in range(6):
    torch.mul(g_prev, 0.9, out=g); g.add_(0.01)
    a.record()
    _lib.check(lib.mde_lbfgs_dev_step(h, _lib.ptr(g), _lib.ptr(g_prev), _lib.ptr(d), 0.3, _lib.ptr(d), _lib.ptr(board), _lib.ptr(work), st))
    b.record(); torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
ms.sort(); m = ms[len(ms) // 2]
print("mde_lbfgs_dev_step N=%d history %d: %.3f ms per step, %.2f TB/s over %.1f GB" % (N, hist, m, 48 * 4 * N / m / 1e9, 48 * 4 * N / 1e9))
