"""Round 6: where the ring layout's feasibility rule puts a problem, and what each kernel costs there.

    python tools/r6_cliff_probe.py CASE [CASE ...]      CASE = name:n:deg:d:graph   graph = uniform | hub | ba

Every case is timed with MDE_PANEL unset (auto), 0 (CSR) and 1 (ring forced); prints one JSON line per (case, mode):
kernel ms (median of 30 launches, HIP events), ms per 1e8 half-edges, layout used, loss.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_graph(kind, n, deg, device, seed=0):
    import bench
    edges, w, _ = bench.make_workload(device, n=n, deg=deg, d=2, graph={"ba": "powerlaw"}.get(kind, kind))
    return edges, w


def main():
    import pymde_amd
    from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    modes = os.environ.get("PROBE_MODES", "auto,0,1").split(",")
    for case in sys.argv[1:]:
        name, n, deg, d, kind = case.split(":")
        n, deg, d = int(n), int(deg), int(d)
        edges, w = make_graph(kind, n, deg, device)
        gen = torch.Generator(device=device)
        gen.manual_seed(0)
        X = torch.randn((n, d), device=device, generator=gen)
        X -= X.mean(0)
        X = X.contiguous()
        degs = torch.bincount(edges.reshape(-1), minlength=n)
        ref = None
        for mode in modes:
            if mode == "auto":
                os.environ.pop("MDE_PANEL", None)
            else:
                os.environ["MDE_PANEL"] = mode
            try:
                plan = EdgePlan(n, edges)
                b = Binding(plan, pymde_amd.penalties.Log1p(w))
                buf = torch.zeros(n * d + 1, dtype=torch.float32, device=device)
                grad, loss = buf[:n * d].view(n, d), buf[n * d:]
                for _ in range(5):
                    fused_evaluate(b, X, grad, loss)
                torch.cuda.synchronize()
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
                for a, c in ev:
                    a.record()
                    fused_evaluate(b, X, grad, loss)
                    c.record()
                torch.cuda.synchronize()
                ms = float(np.median([a.elapsed_time(c) for a, c in ev]))
                g = grad.clone()
                if ref is None:
                    ref = g
                rec = {"case": name, "n": n, "deg": deg, "d": d, "graph": kind, "mode": mode, "kernel_ms": round(ms, 4),
                       "ms_per_1e8_half_edges": round(ms * 1e8 / plan.half_edges, 4), "layout": int(b.struct(d).layout),
                       "stream": b.stream_kind, "loss": float(loss.item()), "max_degree": int(degs.max()),
                       "max_abs_diff_vs_first_mode": float((g - ref).abs().max()), "grad_absmax": float(ref.abs().max())}
                del b, plan
            except Exception as exc:  # noqa: BLE001
                rec = {"case": name, "mode": mode, "error": repr(exc)[:300]}
            print(json.dumps(rec), flush=True)
        del edges, w, X
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
