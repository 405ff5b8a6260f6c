# lbfused_probe.py -- where the time of k_lb_fused goes (library built with -DMDE_LB_PROBE: workgroup 0 leaves
# wall-clock stamps of its phases in the work buffer).  N = 140k floats, history 10 (config 2).
import ctypes, sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from pymde_amd import _lib, util
lib = _lib.load(); dev = torch.device('cuda'); st = _lib.stream_ptr(dev)
N, hist = int(sys.argv[1]) if len(sys.argv) > 1 else 140000, 10
h = ctypes.c_void_p(); _lib.check(lib.mde_lbfgs_create(N, hist, ctypes.byref(h)))
_lib.check(lib.mde_lbfgs_dev_reset(h, st))
work = util.work_buffer(dev, 2); board = torch.zeros(64, dtype=torch.float64, device=dev)
g_prev = torch.randn(N, device=dev); d = -g_prev.clone(); g = torch.empty_like(g_prev)
names = ["stage + partial dots", "arrival 0", "row sums", "arrival 1", "direction (one wave)", "combine + statistics",
         "arrival 2 (workgroup 0 waits)", "final rows + write-back"]
acc = [0.0] * 8; cnt = 0
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = 0.0
for it in range(30):
    torch.mul(g_prev, 0.9, out=g); g.add_(0.01)
    a.record()
    _lib.check(lib.mde_lbfgs_dev_step(h, _lib.ptr(g), _lib.ptr(g_prev), _lib.ptr(d), 0.3, _lib.ptr(d), _lib.ptr(board), _lib.ptr(work), st))
    b.record(); torch.cuda.synchronize()
    if it >= 15:
        w = work.view(torch.int64)[3072 + 4: 3072 + 4 + 9].tolist()   # verdict + 8 words = 4 doubles behind double 3072
        for k in range(8): acc[k] += (w[k + 1] - w[k]) / 100.0
        tot += a.elapsed_time(b) * 1e3; cnt += 1
print("k_lb_fused + k_lb_rescue, N = %d: %.1f us per step by events" % (N, tot / cnt))
for k in range(8): print("  %-32s %6.2f us" % (names[k], acc[k] / cnt))
print("  %-32s %6.2f us" % ("stamped total (workgroup 0)", sum(acc) / cnt))
