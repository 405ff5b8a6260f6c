import sys; sys.path.insert(0, '/root/repo')
import torch, bench, pymde_amd
dev = torch.device('cuda', 0)
n, deg, d = 500_000, 40, 128
edges, w, _ = bench.make_workload(dev, n=n, deg=deg, d=2)
c = pymde_amd.Standardized()
mde = pymde_amd.MDE(n, d, edges, pymde_amd.penalties.Log1p(w * 1e5), constraint=c, device=dev)
torch.manual_seed(0)
X0 = c.initialization(n, d, device=dev).contiguous()
mde.embed(X=X0.clone(), max_iter=12, eps=0.0)
st = mde.solve_stats
print("E:", [round(v, 5) for v in st.average_distortions])
print("res:", [float('%.3g' % v) for v in st.residual_norms])
print("step%:", [float('%.3g' % v) for v in st.step_size_percents])
