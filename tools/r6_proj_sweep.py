# r6_proj_sweep.py -- the constraint maps across embedding dimensions at n = 500k: Standardized tangent projection and
# retraction, Centered projection, the solver's vector statistics and one L-BFGS direction update; ms per call and the
# bytes-per-second they amount to (each reads / writes a handful of n x d arrays)
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pymde_amd
dev = torch.device("cuda", 0)
n = int(os.environ.get("PROJ_N", "500000"))
def tm(f, reps=20):
    for _ in range(3): f()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps
for d in [int(a) for a in sys.argv[1:]] or [2, 3, 8, 16, 24, 32, 48, 50, 64, 96, 100, 128, 200, 256]:
    std, cen = pymde_amd.Standardized(), pymde_amd.Centered()
    torch.manual_seed(0)
    X = std.initialization(n, d, device=dev)
    Z = torch.randn((n, d), device=dev)
    Y = X.clone() + 0.01 * Z
    t_tan = tm(lambda: std.project_onto_tangent_space(X, Z, inplace=True))
    t_ret = tm(lambda: std.project_onto_constraint(Y, inplace=True))
    t_cen = tm(lambda: cen.project_onto_constraint(Y, inplace=True))
    G = (Y.double().T @ Y.double() / n)
    err = float((G - torch.eye(d, device=dev, dtype=torch.float64)).abs().max())
    mb = n * d * 4 / 1e6
    print("d=%4d  %6.1f MB per array  tangent %.3f ms  retraction %.3f ms (|Y'Y/n - I| = %.1e)  centering %.3f ms  -> tangent %.2f TB/s (3 arrays), retraction %.2f TB/s (4 passes)" % (
        d, mb, t_tan, t_ret, err, t_cen, 3 * mb / t_tan / 1e6, 4 * mb / t_ret / 1e6), flush=True)
