"""Parity of the HIP path (through the C ABI) with the oracle and the reference's golden
vectors -- plan, fused kernel, edge-order evaluators, constraints.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from conftest import LOSS_RTOL, assert_grad_close, func_from_golden
from oracle import oracle

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _native_loaded():
    from pymde_amd import _lib
    _lib.load()
    maps = open("/proc/self/maps").read()
    assert "libmde_hip.so" in maps, "the HIP extension is not loaded in this process"


def _make_function(fd, device=DEV):
    """Build the pymde_amd function object equivalent to an oracle descriptor."""
    import pymde_amd
    pen, los = pymde_amd.penalties, pymde_amd.losses
    a0 = torch.tensor(fd["a0"], device=device)
    a1 = None if fd.get("a1") is None else torch.tensor(fd["a1"], device=device)
    s, sn = fd["scalars"], fd["scalars_neg"]
    single = {
        "LINEAR": lambda w, sc: pen.Linear(w), "QUADRATIC": lambda w, sc: pen.Quadratic(w),
        "CUBIC": lambda w, sc: pen.Cubic(w), "POWER": lambda w, sc: pen.Power(w, sc[0]),
        "HUBER": lambda w, sc: pen.Huber(w, sc[0]),
        "LOGISTIC": lambda w, sc: pen.Logistic(w, sc[0], sc[1]),
        "SIGMOID": lambda w, sc: pen.Sigmoid(w, sc[0], sc[1]),
        "HINGE": lambda w, sc: pen.Hinge(w, sc[0], sc[1]),
        "LOG1P": lambda w, sc: pen.Log1p(w, sc[0]), "LOG": lambda w, sc: pen.Log(w, sc[0]),
        "INVPOWER": lambda w, sc: pen.InvPower(w, sc[0]),
        "LOGRATIO": lambda w, sc: pen.LogRatio(w, sc[0]),
        "DEADZONE_QUADRATIC": lambda w, sc: pen._DeadzoneQuadratic(w, sc[0]),
        "DEADZONE_CUBIC": lambda w, sc: pen._DeadzoneCubic(w, sc[0]),
        "CLIPPED_QUADRATIC": lambda w, sc: pen._ClippedQuadratic(w, sc[0]),
    }
    kind, kind_neg = fd["kind"], fd.get("kind_neg", "NONE")
    if kind_neg != "NONE":
        import functools

        def mk(k, sc):
            return lambda w: single[k](w, sc)
        return pen.PushAndPull(a0, mk(kind, s), mk(kind_neg, sn))
    if kind in single:
        return single[kind](a0, s)
    losses = {
        "L_QUADRATIC": lambda: los.Quadratic(a0), "L_WEIGHTED_QUADRATIC": lambda: los.WeightedQuadratic(a0, a1),
        "L_HUBER": lambda: los.Huber(a0, s[0]), "L_CUBIC": lambda: los.Cubic(a0),
        "L_POWER": lambda: los.Power(a0, s[0]), "L_ABSOLUTE": lambda: los.Absolute(a0),
        "L_LOGISTIC": lambda: los.Logistic(a0), "L_FRACTIONAL": lambda: los.Fractional(a0),
        "L_SOFT_FRACTIONAL": lambda: los.SoftFractional(a0, s[0]),
        "L_CLIPPED_QUADRATIC": lambda: los._ClippedQuadratic(a0, s[0]),
        "L_WEIGHTED_POWER": lambda: los._WeightedPower(a0, s[0], a1),
        "L_LOG1P": lambda: los._Log1p(a0, s[0]),
    }
    return losses[kind]()


def _hip_eval(n, d, edges, f, X):
    import pymde_amd
    mde = pymde_amd.MDE(n, d, torch.tensor(edges, device=DEV), f)
    Xt = torch.tensor(X, device=DEV, requires_grad=True)
    E = mde.average_distortion(Xt)
    E.backward()
    return mde, float(E), Xt.grad.cpu().numpy()


def test_native_library_is_the_path():
    _native_loaded()


# ---------------------------------------------------------------- plan (integer work: bit exact)
@pytest.mark.parametrize("n,p,seed", [(5, 4, 0), (64, 500, 1), (1000, 20000, 2), (4097, 60001, 3)])
def test_plan_matches_oracle_bit_exact(n, p, seed):
    from pymde_amd.average_distortion import EdgePlan
    rng = np.random.default_rng(seed)
    pairs = np.stack(np.triu_indices(n, 1), axis=1)
    edges = pairs[rng.choice(len(pairs), p, replace=False)]  # unsorted on purpose
    flip = rng.random(p) < 0.5
    edges[flip] = edges[flip][:, ::-1]  # (j, i) orientation is legal too
    plan = EdgePlan(n, torch.tensor(edges, device=DEV))
    rowptr, nbr, eid = [t.cpu().numpy() for t in plan.csr()]
    w_rowptr, w_nbr, w_eid = oracle.plan_csr(n, edges)
    np.testing.assert_array_equal(rowptr, w_rowptr)
    np.testing.assert_array_equal(nbr, w_nbr)
    np.testing.assert_array_equal(eid, w_eid)
    # sharded plans tile the full plan; the bounds equal the numpy restatement
    from pymde_amd import distributed
    bounds = distributed.shard_bounds(n, torch.tensor(edges, device=DEV), 3)
    assert bounds == oracle.shard_bounds(n, edges, 3)
    for r in range(3):
        sp = EdgePlan(n, torch.tensor(edges, device=DEV), bounds[r], bounds[r + 1])
        s_rowptr, s_nbr, s_eid = [t.cpu().numpy() for t in sp.csr()]
        o = oracle.plan_csr(n, edges, bounds[r], bounds[r + 1])
        np.testing.assert_array_equal(s_rowptr, o[0])
        np.testing.assert_array_equal(s_nbr, o[1])
        np.testing.assert_array_equal(s_eid, o[2])


def test_plan_edge_cases():
    import pymde_amd
    from pymde_amd.average_distortion import EdgePlan
    # isolated vertices (empty rows) and duplicate edges
    edges = np.array([[0, 5], [0, 5], [2, 5], [7, 2]])
    plan = EdgePlan(9, torch.tensor(edges, device=DEV))
    rowptr, nbr, eid = [t.cpu().numpy() for t in plan.csr()]
    o = oracle.plan_csr(9, edges)
    np.testing.assert_array_equal(rowptr, o[0])
    np.testing.assert_array_equal(nbr, o[1])
    # self edges raise the reference's ValueError (problem.py:134-140, test_optim.py:157-170)
    bad = np.array([(0, 1), (0, 0), (0, 2), (1, 2), (1, 1)])
    with pytest.raises(ValueError, match=r"The edge list must not contain self edges.*"):
        pymde_amd.MDE(3, 3, bad, pymde_amd.penalties.Quadratic(torch.ones(5)))
    with pytest.raises(ValueError):
        EdgePlan(3, torch.tensor([[0, 3]], device=DEV))
    with pytest.raises(ValueError, match="more than"):
        pymde_amd.MDE(3, 2, np.array([[0, 1]] * 4), pymde_amd.penalties.Quadratic(torch.ones(4)))


# ---------------------------------------------------------------- the hot kernel
def test_known_answer_62_over_3():
    # pymde/test_optim.py:75-93
    import pymde_amd
    edges = np.array([(0, 1), (0, 2), (1, 2)])
    mde = pymde_amd.MDE(3, 2, edges, pymde_amd.penalties.Quadratic(torch.tensor([1.0, 2.0, 3.0])),
                        constraint=pymde_amd.Standardized())
    X = torch.tensor([[0.0, 0.0], [1.0, 1.0], [3.0, 3.0]], device=DEV)
    assert float(mde.average_distortion(X)) == pytest.approx(62.0 / 3, rel=1e-6)


def test_every_function_against_reference_golden(golden_functions):
    g = golden_functions
    edges, n = g["edges"], int(g["n"])
    for name in g["names"]:
        fd = func_from_golden(g, str(name))
        f = _make_function(fd)
        for tag in ("d1", "d2", "d3", "d8", "zero"):
            X = g["X_zero"] if tag == "zero" else g["X_" + tag]
            mde, E, grad = _hip_eval(n, X.shape[1], edges, f, X)
            want_E = float(g["%s__%s__loss" % (name, tag)])
            if np.isfinite(want_E):
                assert E == pytest.approx(want_E, rel=LOSS_RTOL, abs=1e-7), (name, tag)
            else:
                assert not np.isfinite(E) or abs(E) > 1e30, (name, tag)
            assert_grad_close(grad, g["%s__%s__grad" % (name, tag)])
            want = g["%s__%s__distortions" % (name, tag)]
            got = mde.distortions(torch.tensor(X, device=DEV)).cpu().numpy()
            fin = np.isfinite(want)
            np.testing.assert_allclose(got[fin], want[fin], rtol=2e-5, atol=1e-6, err_msg=str((name, tag)))


@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 7, 8, 16, 33, 64, 100, 128, 200])
def test_fused_kernel_against_oracle_all_dims(d):
    rng = np.random.default_rng(d)
    n, p = 3000, 40000
    pairs_i = rng.integers(0, n, p)
    pairs_j = (pairs_i + 1 + rng.integers(0, n - 1, p)) % n
    edges = np.stack([pairs_i, pairs_j], axis=1)
    X = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    w = np.where(rng.random(p) < 0.3, -1.0, rng.uniform(0.5, 2.0, p)).astype(np.float32)
    dev = rng.uniform(0.3, 2.0, p).astype(np.float32)
    cases = [oracle.func("LOG1P", np.abs(w), None, (1.5,)),
             oracle.func("LOG1P", w, None, (1.5,), "LOG", (1.0,)),
             oracle.func("QUADRATIC", np.abs(w)),
             oracle.func("L_ABSOLUTE", dev),
             oracle.func("L_HUBER", dev, None, (0.5,))]
    for fd in cases:
        f = _make_function(fd)
        _, E, grad = _hip_eval(n, d, edges, f, X)
        wE, wgrad = oracle.average_distortion(edges, X, fd)
        assert E == pytest.approx(wE, rel=LOSS_RTOL), (d, fd["kind"])
        assert_grad_close(grad, wgrad)


@pytest.mark.parametrize("group", ["4", "8", "16", "32", "64"])
def test_row_group_widths_agree(group, monkeypatch):
    """Every lanes-per-row variant of the small-d kernel gives the oracle's answer (run in a
    subprocess because the library reads MDE_GROUP once)."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from oracle import oracle
import pymde_amd
rng = np.random.default_rng(0)
n, p = 2000, 30000
i = rng.integers(0, n, p); j = (i + 1 + rng.integers(0, n - 1, p)) %% n
edges = np.stack([i, j], 1); X = rng.standard_normal((n, 2)).astype(np.float32)
w = rng.uniform(0.5, 2, p).astype(np.float32)
mde = pymde_amd.MDE(n, 2, torch.tensor(edges), pymde_amd.penalties.Log1p(torch.tensor(w)))
Xt = torch.tensor(X, device='cuda', requires_grad=True)
E = mde.average_distortion(Xt); E.backward()
wE, wg = oracle.average_distortion(edges, X, oracle.func('LOG1P', w, None, (1.5,)))
assert abs(float(E) - wE) < 1e-5 * abs(wE)
assert np.abs(Xt.grad.cpu().numpy() - wg).max() < 1e-4 * np.abs(wg).max()
print('ok')
""" % (str(__import__("conftest").ROOT),)
    import os
    env = dict(os.environ, MDE_GROUP=group, MDE_FLAT="0")  # (the row-per-group kernel, not the edge-balanced one)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("d", [1, 2, 3, 4])
def test_edge_balanced_kernel_on_skewed_graphs(d, monkeypatch):
    """The edge-balanced CSR kernel (tiles of 256 half-edge positions, segmented row sums, records for
    rows that cross tiles): hubs spanning a dozen tiles, isolated items, rows of one half-edge; against
    the oracle, against the row-per-group kernel, and bit-equal across vertex-range shards (tiles are
    aligned to global positions)."""
    import pymde_amd
    from pymde_amd import distributed
    from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate
    monkeypatch.setenv("MDE_PANEL", "0")
    rng = np.random.default_rng(40 + d)
    n = 5000
    hubs = np.array([7, 2500, 4999])
    parts = [np.stack([np.full(m, h), rng.choice(np.setdiff1d(np.arange(100, n - 100), hubs), m, replace=False)], 1)
             for h, m in zip(hubs, (3100, 700, 257))]
    body = rng.integers(100, n - 100, (12000, 2))        # items 8..99 and 4900..4998 stay isolated
    body = body[body[:, 0] != body[:, 1]]
    edges = np.unique(np.sort(np.concatenate(parts + [body]), 1), axis=0)
    edges = edges[rng.permutation(len(edges))]
    p = len(edges)
    X = rng.standard_normal((n, d)).astype(np.float32)
    w = np.where(rng.random(p) < 0.3, -1.0, rng.uniform(0.5, 2.0, p)).astype(np.float32)
    et, Xt = torch.tensor(edges, device=DEV), torch.tensor(X, device=DEV)
    for fd in (oracle.func("LOG1P", w, None, (1.5,), "LOG", (1.0,)), oracle.func("L_HUBER", np.abs(w), None, (0.5,))):
        f = _make_function(fd)
        wE, wgrad = oracle.average_distortion(edges, X, fd)
        out = {}
        for mode in ("2", "0"):
            monkeypatch.setenv("MDE_FLAT", mode)
            buf = torch.full((n * d + 1,), float("nan"), device=DEV)
            fused_evaluate(Binding(EdgePlan(n, et), f), Xt, buf[:n * d].view(n, d), buf[n * d:])
            out[mode] = buf.clone()
            assert float(buf[n * d]) == pytest.approx(wE, rel=LOSS_RTOL), (d, fd["kind"], mode)
            assert_grad_close(buf[:n * d].view(n, d).cpu().numpy(), wgrad)
        assert float(out["2"][8 * d:100 * d].abs().sum()) == 0.0  # isolated items: zero rows, not stale memory
        # the same launch twice: bitwise; three vertex-range shards: the rows they own, bitwise
        monkeypatch.setenv("MDE_FLAT", "2")
        again = torch.zeros(n * d + 1, device=DEV)
        fused_evaluate(Binding(EdgePlan(n, et), f), Xt, again[:n * d].view(n, d), again[n * d:])
        assert torch.equal(again, out["2"])
        bounds = distributed.shard_bounds(n, et, 3)
        total = torch.zeros(n * d + 1, device=DEV)
        for r in range(3):
            lo, hi = distributed.shard_range(bounds, r)
            part = torch.zeros(n * d + 1, device=DEV)
            fused_evaluate(Binding(EdgePlan(n, et, lo, hi), f), Xt, part[:n * d].view(n, d), part[n * d:])
            assert float(part[:lo * d].abs().sum()) == 0 and float(part[hi * d:n * d].abs().sum()) == 0
            total += part
        assert torch.equal(total[:n * d], out["2"][:n * d]), (d, fd["kind"])
        assert float(total[n * d]) == pytest.approx(wE, rel=LOSS_RTOL)


def test_zero_distance_edges_contribute_nothing():
    # average_distortion.py:81-88: g = NaN/Inf -> 1.0 but diff = 0, so the gradient stays finite
    import pymde_amd
    pen = pymde_amd.penalties
    n = 6
    edges = np.array([[0, 1], [1, 2], [3, 4], [0, 5]])
    X = np.array([[0, 0], [0, 0], [1, 0], [2, 2], [2, 2], [0.5, 1]], dtype=np.float32)
    w = torch.tensor([1.0, 2.0, -1.0, 1.0])
    for f, fd in ((pen.Log1p(w.abs()), oracle.func("LOG1P", w.abs().numpy(), None, (1.5,))),
                  (pen.PushAndPull(w, pen.Log1p, pen.Log), oracle.func("LOG1P", w.numpy(), None, (1.5,), "LOG", (1.0,))),
                  (pen.Linear(w.abs()), oracle.func("LINEAR", w.abs().numpy()))):
        _, E, grad = _hip_eval(n, 2, edges, f, X)
        wE, wgrad = oracle.average_distortion(edges, X, fd)
        assert np.isfinite(grad).all()
        assert_grad_close(grad, wgrad)
        assert (np.isinf(E) and np.isinf(wE)) or E == pytest.approx(wE, rel=1e-5)


def test_scalar_weight_broadcast_and_grad_output():
    import pymde_amd
    rng = np.random.default_rng(0)
    n, p = 500, 3000
    i = rng.integers(0, n, p)
    j = (i + 1 + rng.integers(0, n - 1, p)) % n
    edges = np.stack([i, j], 1)
    X = rng.standard_normal((n, 3)).astype(np.float32)
    f = pymde_amd.penalties.Huber(torch.tensor(2.0))  # weights.nelement() == 1 (penalties.py:224-227)
    mde = pymde_amd.MDE(n, 3, edges, f)
    Xt = torch.tensor(X, device=DEV, requires_grad=True)
    (3.0 * mde.average_distortion(Xt)).backward()  # upstream scalar (average_distortion.py:105)
    wE, wgrad = oracle.average_distortion(edges, X, oracle.func("HUBER", [2.0], None, (0.5,)), grad_output=3.0)
    assert_grad_close(Xt.grad.cpu().numpy(), wgrad)
    assert float(mde.average_distortion(Xt.detach())) == pytest.approx(wE, rel=1e-5)


def test_arbitrary_callable_takes_the_unfused_path():
    """Any Python callable on distances is a legal distortion function
    (docs_src/source/mde/index.rst:300-306): gather/scatter stay on the HIP kernels."""
    import pymde_amd
    rng = np.random.default_rng(1)
    n, p = 800, 6000
    i = rng.integers(0, n, p)
    j = (i + 1 + rng.integers(0, n - 1, p)) % n
    edges = np.stack([i, j], 1)
    X = rng.standard_normal((n, 2)).astype(np.float32)
    w = torch.tensor(rng.uniform(0.5, 2, p).astype(np.float32), device=DEV)

    def custom(distances):
        return w * torch.log1p(distances.pow(1.5))
    mde = pymde_amd.MDE(n, 2, edges, custom)
    assert not mde._binding().fused
    Xt = torch.tensor(X, device=DEV, requires_grad=True)
    E = mde.average_distortion(Xt)
    E.backward()
    wE, wgrad = oracle.average_distortion(edges, X, oracle.func("LOG1P", w.cpu().numpy(), None, (1.5,)))
    assert float(E) == pytest.approx(wE, rel=1e-5)
    assert_grad_close(Xt.grad.cpu().numpy(), wgrad)
    # composition of built-in functions differentiates through the element-wise HIP kernel
    f2 = pymde_amd.penalties.Log1p(w)
    mde2 = pymde_amd.MDE(n, 2, edges, lambda dd: 0.5 * f2(dd) + 0.5 * f2(dd))
    Xt2 = torch.tensor(X, device=DEV, requires_grad=True)
    mde2.average_distortion(Xt2).backward()
    assert_grad_close(Xt2.grad.cpu().numpy(), wgrad)


def test_distances_differences_and_norm_backward(golden_functions):
    import pymde_amd
    g = golden_functions
    edges, n = g["edges"], int(g["n"])
    for tag in ("d1", "d2", "d3", "d8", "zero"):
        X = g["X_zero"] if tag == "zero" else g["X_" + tag]
        mde = pymde_amd.MDE(n, X.shape[1], edges, pymde_amd.penalties.Quadratic(torch.ones(len(edges))))
        Xt = torch.tensor(X, device=DEV)
        np.testing.assert_allclose(mde.distances(Xt).cpu().numpy(), g[tag + "__distances"], rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(mde.differences(Xt).cpu().numpy(), g[tag + "__differences"])
        Xg = Xt.clone().requires_grad_(True)
        gout = torch.linspace(0.5, 1.5, len(edges), device=DEV)
        (mde.distances(Xg) * gout).sum().backward()
        want = oracle.distances_backward(edges, X, gout.cpu().numpy())
        assert_grad_close(Xg.grad.cpu().numpy(), want)
    # test_optim.py:57-71: sub-gradient 0 at coincident points
    mde = pymde_amd.MDE(3, 3, np.array([(0, 1)]), pymde_amd.penalties.Quadratic(torch.ones(1)))
    Xo = torch.ones((3, 3), requires_grad=True, device=DEV)
    mde.distances(Xo).backward()
    np.testing.assert_array_equal(Xo.grad.cpu().numpy(), g["norm_grad_zero"])


def test_high_distortion_pairs_sorted():
    import pymde_amd
    rng = np.random.default_rng(2)
    n = 100
    edges = np.stack(np.triu_indices(n, 1), 1)[:900]
    mde = pymde_amd.MDE(n, 2, edges, pymde_amd.penalties.Cubic(torch.ones(900)))
    X = torch.tensor(rng.standard_normal((n, 2)).astype(np.float32), device=DEV)
    pairs, dist = mde.high_distortion_pairs(X)
    dd = dist.cpu().numpy()
    assert (np.diff(dd) <= 0).all() and pairs.shape == (900, 2)


# ---------------------------------------------------------------- constraints
def test_constraints_against_reference_golden(golden_constraints):
    import pymde_amd
    g = golden_constraints
    std, cen = pymde_amd.Standardized(), pymde_amd.Centered()
    for n, d in g["shapes"]:
        tag = "%dx%d" % (n, d)
        X = torch.tensor(g["X_" + tag], device=DEV)
        Z = torch.tensor(g["Z_" + tag], device=DEV)
        P = std.project_onto_constraint(X, inplace=False)
        np.testing.assert_allclose(P.cpu().numpy(), g["std_retract_" + tag], rtol=2e-3, atol=5e-4)
        Pn = P.double().cpu().numpy()
        np.testing.assert_allclose(Pn.T @ Pn / n, np.eye(d), rtol=1e-4, atol=1e-5)  # test_util.py:20-71
        np.testing.assert_allclose(Pn.mean(0), 0, atol=1e-5)
        Pref = torch.tensor(g["std_retract_" + tag], device=DEV)
        T = std.project_onto_tangent_space(Pref, Z, inplace=False)
        np.testing.assert_allclose(T.cpu().numpy(), g["std_tangent_" + tag], rtol=1e-4, atol=1e-5)
        C = cen.project_onto_constraint(X, inplace=False)
        np.testing.assert_allclose(C.cpu().numpy(), g["centered_" + tag], rtol=1e-5, atol=1e-6)
        assert torch.equal(X, torch.tensor(g["X_" + tag], device=DEV))  # inplace=False leaves X alone
    anc = pymde_amd.Anchored(torch.tensor(g["anchors"]), torch.tensor(g["anchor_values"]))
    Z = torch.tensor(g["anchor_Z"], device=DEV)
    np.testing.assert_array_equal(anc.project_onto_tangent_space(None, Z, inplace=False).cpu().numpy(),
                                  g["anchor_tangent"])
    np.testing.assert_array_equal(anc.project_onto_constraint(Z, inplace=False).cpu().numpy(),
                                  g["anchor_retract"])


def test_sphere_constraint_against_reference_golden_and_oracle(golden_sphere):
    """`_Sphere` (private in the reference, constraints.py:203-231) on the HIP row kernel: the reference's own
    outputs, then the oracle on a ragged set of shapes (one thread per row up to d = 32, one wave per row above),
    and a solve through the generic path that stays on the sphere."""
    import pymde_amd
    from pymde_amd import constraints
    g = golden_sphere
    for n, d, radius in g["cases"]:
        tag = "%dx%d" % (int(n), int(d))
        c = constraints._Sphere(float(radius))
        Z = torch.tensor(g["Z_" + tag], device=DEV)
        X = torch.tensor(g["X_" + tag], device=DEV)
        R = c.project_onto_constraint(Z, inplace=False)
        np.testing.assert_allclose(R.cpu().numpy(), g["retract_" + tag], rtol=1e-5, atol=1e-6)
        T = c.project_onto_tangent_space(X, Z, inplace=False)
        np.testing.assert_allclose(T.cpu().numpy(), g["tangent_" + tag], rtol=1e-5, atol=1e-5)
        assert torch.equal(Z, torch.tensor(g["Z_" + tag], device=DEV))   # inplace=False leaves Z alone
        Zi = Z.clone()
        assert c.project_onto_constraint(Zi, inplace=True) is Zi and torch.equal(Zi, R)
    rng = np.random.default_rng(2)
    for n, d, radius in [(1, 1, 1.0), (3, 32, 2.0), (5, 33, 0.25), (100001, 2, 1.5), (4097, 65, 1.0), (50, 200, 7.0)]:
        c = constraints._Sphere(radius)
        Zn = rng.standard_normal((n, d)).astype(np.float32)
        Xn = oracle.sphere_retract(rng.standard_normal((n, d)), radius).astype(np.float32)
        R = c.project_onto_constraint(torch.tensor(Zn, device=DEV), inplace=False).cpu().numpy()
        np.testing.assert_allclose(R, oracle.sphere_retract(Zn, radius), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(np.linalg.norm(R.astype(np.float64), axis=1), radius, rtol=1e-5)
        T = c.project_onto_tangent_space(torch.tensor(Xn, device=DEV), torch.tensor(Zn, device=DEV), inplace=False)
        np.testing.assert_allclose(T.cpu().numpy(), oracle.sphere_tangent(Xn, Zn, radius), rtol=1e-5, atol=1e-5)
    X0 = constraints._Sphere(2.0).initialization(1000, 3, device=DEV)
    np.testing.assert_allclose(X0.norm(dim=1).cpu().numpy(), 2.0, rtol=1e-5)
    n, p = 400, 3000
    i = rng.integers(0, n, p)
    j = (i + 1 + rng.integers(0, n - 1, p)) % n
    w = torch.tensor(rng.uniform(0.5, 2, p).astype(np.float32), device=DEV)
    mde = pymde_amd.MDE(n, 3, np.stack([i, j], 1), pymde_amd.penalties.Quadratic(w), constraint=constraints._Sphere(1.0))
    torch.manual_seed(1)
    X = mde.embed(max_iter=30)
    E = mde.solve_stats.average_distortions
    assert E[-1] < E[0]
    np.testing.assert_allclose(X.norm(dim=1).cpu().numpy(), 1.0, rtol=1e-5)


@pytest.mark.parametrize("n,d", [(2, 2), (10, 3), (100, 3), (1000, 2), (1000, 3), (1000, 250), (5000, 128),
                                 (3000, 64), (777, 96), (600000, 2), (350001, 3),
                                 (4001, 5), (4001, 7), (4001, 10), (4001, 31), (4001, 33), (4001, 50), (4001, 65), (4001, 100),
                                 (4001, 127), (3001, 129), (3001, 200), (3001, 256), (2001, 300), (2001, 512)])
def test_proj_standardized_property(n, d):
    # pymde/test_util.py:20-71 (shapes incl. (1000, 250)); d = 5 .. 128 take the f32 MFMA Gram (register tiles of
    # 32 / 64 / 128 columns, padded), d = 129 .. 512 its column blocks of 128 (round 6); the two
    # tall shapes are vectors of more than 2^20 floats (many trips per thread of the reducing kernels)
    from pymde_amd import util
    torch.manual_seed(0)
    X = torch.randn((n, d), device=DEV)
    demean = n > 2
    P = util.proj_standardized(X, demean=demean).double().cpu().numpy()
    np.testing.assert_allclose(P.T @ P / n, np.eye(d), rtol=1e-4, atol=2e-5)
    if demean:
        np.testing.assert_allclose(P.mean(0), 0, atol=1e-5)
        want = oracle.proj_standardized(X.cpu().numpy(), demean=True)
        np.testing.assert_allclose(P, want, rtol=2e-3, atol=5e-4)


@pytest.mark.parametrize("n,d", [(1000, 2), (600000, 2), (350001, 3), (300000, 4)])
def test_centering_and_vector_statistics_on_long_vectors(n, d):
    """The reducing kernels on short and on long vectors (many trips per thread, 16-byte and tail loads)
    against float64 torch: Centered retraction of a line-search trial point, and the solver's statistics."""
    import ctypes
    import pymde_amd
    from pymde_amd import _lib, util
    lib = _lib.load()
    torch.manual_seed(3)
    X = torch.randn((n, d), device=DEV) + 0.25
    D = torch.randn((n, d), device=DEV)
    c = pymde_amd.Centered()
    Z = c.project_onto_constraint(X.clone(), inplace=True)
    want = (X.double() - X.double().mean(0)).float()
    assert float((Z - want).abs().max()) <= 2e-7 * float(X.abs().max())
    assert float(Z.double().mean(0).abs().max()) < 1e-7
    work = util.work_buffer(torch.device(DEV), d)
    out = torch.empty_like(X)
    _lib.check(lib.mde_center_step(n, d, _lib.ptr(X), _lib.ptr(D), ctypes.c_float(0.375), _lib.ptr(out), _lib.ptr(work),
                                   _lib.stream_ptr()))
    step = torch.addcmul(X, D, torch.tensor(0.375, device=DEV))  # fl(X + t D), as the kernel rounds it
    want = (step.double() - step.double().mean(0)).float()
    assert float((out - want).abs().max()) <= 2e-7 * float(step.abs().max())
    board = torch.zeros(64, dtype=torch.float64, device=DEV)
    g, dd, x = X.reshape(-1), D.reshape(-1), out.reshape(-1)
    _lib.check(lib.mde_vec_stats(n * d, _lib.ptr(g), _lib.ptr(dd), _lib.ptr(x), _lib.ptr(board), _lib.ptr(work),
                                 _lib.stream_ptr()))
    b = board.cpu().numpy()
    g64, d64, x64 = g.double(), dd.double(), x.double()
    ref = [float((g64 * d64).sum()), float((g64 * g64).sum()), float(g64.abs().sum()), float(g64.abs().max()), 0.0,
           float((d64 * d64).sum()), float(d64.abs().max()), float((x64 * x64).sum())]
    np.testing.assert_allclose(b[:8], ref, rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("n,d", [(4001, 5), (4001, 8), (4001, 24), (4001, 50), (4001, 100), (3001, 128), (3001, 130),
                                 (3001, 200), (3001, 256), (2001, 384)])
def test_standardized_tangent_projection_across_widths(n, d):
    """Z - X (Z^T X) / n [ref: constraints.py:186-192] at widths that take the padded / blocked MFMA kernels (round 6),
    against a float64 evaluation; in place and not."""
    import pymde_amd
    std = pymde_amd.Standardized()
    torch.manual_seed(d)
    X = std.initialization(n, d, device=DEV)
    Z = torch.randn((n, d), device=DEV)
    want = (Z.double() - X.double() @ (Z.double().T @ X.double()) / n)
    got = std.project_onto_tangent_space(X, Z, inplace=False)
    err = float((got.double() - want).abs().max())
    assert err <= 2e-5 * max(1.0, float(want.abs().max())), (d, err)
    Zc = Z.clone()
    got2 = std.project_onto_tangent_space(X, Zc, inplace=True)
    assert torch.equal(got2, got) and got2.data_ptr() == Zc.data_ptr()
    # the retraction, in place (d > 128: the blocked product reads its rows from a scratch copy)
    Y = (X + 0.05 * Z).contiguous()
    P = std.project_onto_constraint(Y.clone(), inplace=True).double()
    G = (P.T @ P / n).cpu().numpy()
    np.testing.assert_allclose(G, np.eye(d), rtol=1e-4, atol=3e-5)
    want_P = oracle.proj_standardized(Y.cpu().numpy(), demean=True)
    np.testing.assert_allclose(P.cpu().numpy(), want_P, rtol=2e-3, atol=5e-4)


def test_standardized_initialization():
    # pymde/test_optim.py:14-18
    import pymde_amd
    torch.manual_seed(0)
    X = pymde_amd.Standardized().initialization(5, 3, device=DEV).double().cpu().numpy()
    np.testing.assert_allclose(X.T @ X / 5, np.eye(3), rtol=1e-4, atol=1e-5)
    Xc = pymde_amd.Centered().initialization(1000, 2, device=DEV)
    assert abs(float(Xc.mean())) < 1e-6


@pytest.mark.parametrize("da,db", [(32, 32), (64, 128), (128, 128), (33, 7), (9, 9), (250, 250)])
def test_gram_mfma_and_generic_paths(da, db):
    import ctypes
    from pymde_amd import _lib, util
    lib = _lib.load()
    n = 4321
    torch.manual_seed(1)
    A = torch.randn((n, da), device=DEV)
    B = torch.randn((n, db), device=DEV) + 0.5
    out = torch.empty((da, db), dtype=torch.float64, device=DEV)
    work = util.work_buffer(torch.device(DEV), max(da, db))
    _lib.check(lib.mde_gram(n, da, db, _lib.ptr(A), _lib.ptr(B), _lib.ptr(out), _lib.ptr(work),
                            _lib.stream_ptr()))
    want = A.double().T @ B.double()
    # asymmetric inputs: a transposed C/D mapping of the MFMA tile would fail here
    np.testing.assert_allclose(out.cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=2e-3)


@pytest.mark.parametrize("d", [32, 64, 96, 128, 160, 40])
def test_right_multiply_mfma_and_generic_paths(d):
    """out = base + alpha A M: widths that are multiples of 32 (M <= 64 KB as fp32) take the f32
    MFMA tile kernel, the others the generic one; also in place, as the projections use it."""
    from pymde_amd import _lib
    lib = _lib.load()
    n = 4321
    torch.manual_seed(2)
    A = torch.randn((n, d), device=DEV)
    base = torch.randn((n, d), device=DEV)
    M = torch.randn((d, d), dtype=torch.float64, device=DEV) / np.sqrt(d)
    want = (base.double() - 0.25 * (A.double() @ M)).cpu().numpy()
    out = torch.empty((n, d), device=DEV)
    _lib.check(lib.mde_right_multiply_add(n, d, d, _lib.ptr(A), _lib.ptr(M), -0.25, _lib.ptr(base),
                                          _lib.ptr(out), _lib.stream_ptr()))
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    # in place on A (asymmetric M: a transposed operand mapping would fail here)
    A2 = A.clone()
    _lib.check(lib.mde_right_multiply(n, d, d, _lib.ptr(A2), _lib.ptr(M), _lib.ptr(A2), _lib.stream_ptr()))
    np.testing.assert_allclose(A2.cpu().numpy(), (A.double() @ M).cpu().numpy(), rtol=1e-4, atol=1e-4)
    # Standardized tangent projection Z - X (Z^T X) / n at this width
    import pymde_amd
    c = pymde_amd.Standardized()
    X = c.project_onto_constraint(A, inplace=False)
    T = c.project_onto_tangent_space(X, base, inplace=False).double().cpu().numpy()
    Xd, Zd = X.double().cpu().numpy(), base.double().cpu().numpy()
    np.testing.assert_allclose(T, Zd - Xd @ (Zd.T @ Xd) / n, rtol=1e-3, atol=2e-4)


# ---------------------------------------------------------------- determinism and invariants
def test_bitwise_reproducible_and_translation_invariant():
    import pymde_amd
    rng = np.random.default_rng(3)
    n, p = 20000, 400000
    i = rng.integers(0, n, p)
    j = (i + 1 + rng.integers(0, n - 1, p)) % n
    edges = torch.tensor(np.stack([i, j], 1), device=DEV)
    w = torch.tensor(np.where(rng.random(p) < 0.3, -1.0, 1.5).astype(np.float32), device=DEV)
    f = pymde_amd.penalties.PushAndPull(w, pymde_amd.penalties.Log1p, pymde_amd.penalties.Log)
    X = torch.tensor(rng.standard_normal((n, 2)).astype(np.float32), device=DEV)
    grads, losses = [], []
    for _ in range(3):
        mde = pymde_amd.MDE(n, 2, edges, f)  # a fresh plan every time
        Xt = X.clone().requires_grad_(True)
        E = mde.average_distortion(Xt)
        E.backward()
        grads.append(Xt.grad.clone())
        losses.append(E.detach().clone())
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])
    assert torch.equal(losses[0], losses[1])
    # E depends on differences only: the gradient rows sum to ~0
    s = grads[0].double().sum(0).abs().max().item()
    assert s < 1e-4 * grads[0].abs().max().item() * np.sqrt(n)


# ---------------------------------------------------------------- LDS column-panel kernel
_PANEL_CODE = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from oracle import oracle
import pymde_amd
from pymde_amd import _lib
rng = np.random.default_rng(7)
for (n, p, d, fname) in [(30000, 400000, 2, 'log1p'), (30000, 400000, 2, 'pushpull'), (50000, 300000, 3, 'quad'),
                         (30000, 400000, 2, 'log1p_cb'), (70001, 500003, 2, 'pushpull_cb'), (20000, 250000, 3, 'quad_cb'),
                         (20000, 250000, 1, 'absolute'), (70001, 500003, 2, 'pushpull_lr'), (9000, 200000, 4, 'huber'),
                         (30000, 400000, 2, 'huber'), (30000, 400000, 3, 'absolute'), (30000, 400000, 2, 'lquad'),
                         (30000, 400000, 2, 'log1p_scalar'), (30000, 400000, 2, 'quad_scalar')]:
    i = rng.integers(0, n, p); j = (i + 1 + rng.integers(0, n - 1, p)) %% n
    # hub vertices (degree ~ p/20 and p/50): rows with far more entries per tile than a wave has
    # iterations exercise the run-folding slow path; vertex n-1 is isolated (empty row)
    i[: p // 20] = 3
    i[p // 20: p // 20 + p // 50] = n // 2
    j[j == i] = (j[j == i] + 1) %% n
    j[j == n - 1] = 0; i[i == n - 1] = 1
    keep = i != j
    i, j = i[keep], j[keep]; p = len(i)
    edges = np.stack([i, j], 1)
    X = rng.standard_normal((n, d)).astype(np.float32)
    w = np.where(rng.random(p) < 0.3, -1.0, rng.uniform(0.5, 2.0, p)).astype(np.float32)
    dev = rng.uniform(0.3, 2.0, p).astype(np.float32)
    pen, los = pymde_amd.penalties, pymde_amd.losses
    if fname.endswith('_cb'):
        # few distinct weights (a k-NN graph's 1 / 2, -1 for repulsive pairs): at d = 2 the panel
        # layout streams them as a codebook index inside the packed word
        w = rng.choice(np.array([-1.0, 1.0, 2.0], dtype=np.float32), size=p, p=[0.3, 0.4, 0.3])
    wt, dt = torch.tensor(w, device='cuda'), torch.tensor(dev, device='cuda')
    f, fd = {
        'log1p_cb': (pen.Log1p(wt.abs()), oracle.func('LOG1P', np.abs(w), None, (1.5,))),
        'pushpull_cb': (pen.PushAndPull(wt, pen.Log1p, pen.Log), oracle.func('LOG1P', w, None, (1.5,), 'LOG', (1.0,))),
        'quad_cb': (pen.Quadratic(wt.abs()), oracle.func('QUADRATIC', np.abs(w))),
        'log1p': (pen.Log1p(wt.abs()), oracle.func('LOG1P', np.abs(w), None, (1.5,))),
        'pushpull': (pen.PushAndPull(wt, pen.Log1p, pen.Log), oracle.func('LOG1P', w, None, (1.5,), 'LOG', (1.0,))),
        'pushpull_lr': (pen.PushAndPull(wt), oracle.func('LOG1P', w, None, (1.5,), 'LOGRATIO', (2.0,))),
        'quad': (pen.Quadratic(wt.abs()), oracle.func('QUADRATIC', np.abs(w))),
        'absolute': (los.Absolute(dt), oracle.func('L_ABSOLUTE', dev)),
        'huber': (los.Huber(dt, 0.5), oracle.func('L_HUBER', dev, None, (0.5,))),
        'lquad': (los.Quadratic(dt), oracle.func('L_QUADRATIC', dev)),
        # ONE weight for every edge (a0_scalar = 1): the ring kernel's padding lanes must not carry it
        'log1p_scalar': (pen.Log1p(torch.tensor(2.0)), oracle.func('LOG1P', [2.0], None, (1.5,))),
        'quad_scalar': (pen.Quadratic(torch.tensor(0.5)), oracle.func('QUADRATIC', [0.5])),
    }[fname]
    mde = pymde_amd.MDE(n, d, torch.tensor(edges, device='cuda'), f)
    Xt = torch.tensor(X, device='cuda', requires_grad=True)
    E = mde.average_distortion(Xt); E.backward()
    assert mde._binding().struct(d).layout == %d, 'unexpected layout'
    assert mde._binding().codebook == (fname.endswith('_cb') and d in (2, 3) and mde._binding().struct(d).layout == 1), fname
    wE, wg = oracle.average_distortion(edges, X, fd)
    assert abs(float(E) - wE) <= 1e-5 * abs(wE), (fname, float(E), wE)
    err = np.abs(Xt.grad.cpu().numpy() - wg).max()
    assert err <= 1e-4 * np.abs(wg).max() , (fname, err)
    # forward only (no grad) gives the same value
    assert abs(float(mde.average_distortion(Xt.detach())) - wE) <= 1e-5 * abs(wE)
    # one writer per accumulator and a fixed summation order: repeated evaluations, also from a
    # freshly built plan, are bitwise identical
    Xt2 = torch.tensor(X, device='cuda', requires_grad=True)
    mde2 = pymde_amd.MDE(n, d, torch.tensor(edges, device='cuda'), f)
    E2 = mde2.average_distortion(Xt2); E2.backward()
    assert torch.equal(Xt2.grad, Xt.grad) and torch.equal(E2.detach(), E.detach())
# vertex-range shards (multi-GPU layout; several column groups per row block when panels are on)
from pymde_amd import distributed
from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate
n, p, d = 120000, 600000, 2
i = rng.integers(0, n, p); j = (i + 1 + rng.integers(0, n - 1, p)) %% n
edges = np.stack([i, j], 1); et = torch.tensor(edges, device='cuda')
w = rng.uniform(0.5, 2.0, p).astype(np.float32); X = rng.standard_normal((n, d)).astype(np.float32)
f = pymde_amd.penalties.Log1p(torch.tensor(w, device='cuda')); Xt = torch.tensor(X, device='cuda')
wE, wg = oracle.average_distortion(edges, X, oracle.func('LOG1P', w, None, (1.5,)))
for world in (1, 2, 8):
    bounds = distributed.shard_bounds(n, et, world)
    total = torch.zeros(n * d + 1, device='cuda')
    for r in range(world):
        lo, hi = distributed.shard_range(bounds, r)
        buf = torch.zeros(n * d + 1, device='cuda')
        b = Binding(EdgePlan(n, et, lo, hi), f)
        fused_evaluate(b, Xt, buf[:n * d].view(n, d), buf[n * d:])
        assert b.struct(d).layout == %d
        assert float(buf[:lo * d].abs().sum()) == 0 and float(buf[hi * d:n * d].abs().sum()) == 0
        total += buf
    assert abs(float(total[n * d]) - wE) <= 1e-5 * abs(wE), (world, float(total[n * d]), wE)
    err = np.abs(total[:n * d].view(n, d).cpu().numpy() - wg).max()
    assert err <= 1e-4 * np.abs(wg).max(), (world, err)
print('ok')
"""


@pytest.mark.parametrize("mode,bs", [("1", "1024"), ("0", "1024")])
def test_panel_kernel_against_oracle(mode, bs):
    """The LDS column-panel kernel (forced on with MDE_PANEL=1 at sizes the oracle checks in
    seconds) and the CSR kernel (MDE_PANEL=0) both reproduce the oracle."""
    import os
    import subprocess
    import sys
    code = _PANEL_CODE % (str(__import__("conftest").ROOT), int(mode), int(mode))
    env = dict(os.environ, MDE_PANEL=mode, MDE_PANEL_BS=bs)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "ok" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])


@pytest.mark.parametrize("d,fname", [(2, "huber"), (3, "quadratic"), (1, "log1p"), (4, "absolute")])
def test_ring_kernel_on_dense_graphs(monkeypatch, d, fname):
    """Many entries per (row, chunk): the layout orders the entries of a chunk by column (in CSR order
    an iteration -- distinct rows -- would find a handful of rows in its look-ahead), uses tall row
    blocks and several column groups of few chunks.  Forced on at a size the oracle checks in seconds."""
    import pymde_amd
    monkeypatch.setenv("MDE_PANEL", "1")
    rng = np.random.default_rng(13 + d)
    n, p = 12000, 2_500_000
    i = rng.integers(0, n, p)
    j = (i + 1 + rng.integers(0, n - 1, p)) % n
    key = np.unique(np.minimum(i, j).astype(np.int64) * n + np.maximum(i, j))
    edges = np.stack([key // n, key % n], 1)
    p = len(edges)
    X = rng.standard_normal((n, d)).astype(np.float32)
    a = rng.uniform(0.5, 2.0, p).astype(np.float32)
    at = torch.tensor(a, device=DEV)
    f, fd = {
        "huber": (pymde_amd.losses.Huber(at, 0.7), oracle.func("L_HUBER", a, None, (0.7,))),
        "quadratic": (pymde_amd.penalties.Quadratic(at), oracle.func("QUADRATIC", a)),
        "log1p": (pymde_amd.penalties.Log1p(at), oracle.func("LOG1P", a, None, (1.5,))),
        "absolute": (pymde_amd.losses.Absolute(at), oracle.func("L_ABSOLUTE", a)),
    }[fname]
    mde = pymde_amd.MDE(n, d, torch.tensor(edges, device=DEV), f)
    Xt = torch.tensor(X, device=DEV, requires_grad=True)
    E = mde.average_distortion(Xt)
    E.backward()
    assert mde._binding().struct(d).layout == 1
    from pymde_amd import _lib
    padded = int(_lib.load().mde_plan_layout_half_edges(mde._binding().plan.handle, 1))
    assert padded <= 1.35 * 2 * p   # the iterations are (nearly) full
    wE, wgrad = oracle.average_distortion(edges, X, fd)
    assert float(E) == pytest.approx(wE, rel=1e-5)
    assert_grad_close(Xt.grad.cpu().numpy(), wgrad)


def test_ring_kernel_on_dense_graphs_sharded(monkeypatch):
    """The dense-graph layout (entries ordered by column) on vertex-range shards: every rank's rows
    against the oracle's, W = 2 and 5 (ragged last shard)."""
    import pymde_amd
    from pymde_amd import distributed
    from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate
    monkeypatch.setenv("MDE_PANEL", "1")
    rng = np.random.default_rng(29)
    n, p, d = 12000, 2_000_000, 2
    i = rng.integers(0, n, p)
    j = (i + 1 + rng.integers(0, n - 1, p)) % n
    key = np.unique(np.minimum(i, j).astype(np.int64) * n + np.maximum(i, j))
    edges = np.stack([key // n, key % n], 1)
    p = len(edges)
    X = rng.standard_normal((n, d)).astype(np.float32)
    dev = (1.0 + rng.integers(0, 30, p)).astype(np.float32)
    et = torch.tensor(edges, device=DEV)
    f = pymde_amd.losses.Huber(torch.tensor(dev, device=DEV), 1.0)
    Xt = torch.tensor(X, device=DEV)
    wE, wgrad = oracle.average_distortion(edges, X, oracle.func("L_HUBER", dev, None, (1.0,)))
    for world in (2, 5):
        bounds = distributed.shard_bounds(n, et, world)
        total = torch.zeros(n * d + 1, device=DEV)
        for r in range(world):
            lo, hi = distributed.shard_range(bounds, r)
            buf = torch.zeros(n * d + 1, device=DEV)
            b = Binding(EdgePlan(n, et, lo, hi), f)
            fused_evaluate(b, Xt, buf[:n * d].view(n, d), buf[n * d:])
            assert b.struct(d).layout == 1
            assert float(buf[:lo * d].abs().sum()) == 0 and float(buf[hi * d:n * d].abs().sum()) == 0
            total += buf
        assert float(total[n * d]) == pytest.approx(wE, rel=1e-5), world
        assert_grad_close(total[:n * d].view(n, d).cpu().numpy(), wgrad)


def test_ring_layout_peels_hub_rows():
    """Auto layout choice at a size where the LDS-ring kernel runs (n d 4 >= 6 MB) on a graph with a hub vertex
    of half a million half-edges.  Rounds 3-5: the rows of a wave iteration are distinct, the hub would need one
    iteration per entry, and the builder gave the whole layout up for the CSR kernel.  Round 6: the hub row is
    PEELED -- the ring streams hold everybody else's entries, k_hub_rows / k_hub_finish evaluate the hub from the CSR
    plan behind the ring kernel -- same results, against the oracle, bitwise reproducible."""
    import pymde_amd
    rng = np.random.default_rng(11)
    n, p, hub = 800_000, 8_000_000, 500_000
    i = rng.integers(0, n, p)
    j = (i + 1 + rng.integers(0, n - 1, p)) % n
    i[:hub] = 12345
    j[:hub] = rng.choice(n - 1, hub, replace=False)
    j[:hub][j[:hub] >= 12345] += 1
    edges = np.stack([np.minimum(i, j), np.maximum(i, j)], 1)
    w = rng.choice(np.array([1.0, 2.0], dtype=np.float32), size=p)
    X = rng.standard_normal((n, 2)).astype(np.float32)
    mde = pymde_amd.MDE(n, 2, torch.tensor(edges, device=DEV), pymde_amd.penalties.Log1p(torch.tensor(w, device=DEV)))
    Xt = torch.tensor(X, device=DEV, requires_grad=True)
    E = mde.average_distortion(Xt)
    E.backward()
    assert mde._binding().struct(2).layout == 1
    info = mde._plan.ring_info()
    assert info["hub_rows"] == 1 and info["hub_half_edges"] >= hub and not info["permuted"], info
    wE, wgrad = oracle.average_distortion(edges, X, oracle.func("LOG1P", w, None, (1.5,)))
    assert float(E) == pytest.approx(wE, rel=1e-5)
    assert_grad_close(Xt.grad.cpu().numpy(), wgrad)
    Xt2 = torch.tensor(X, device=DEV, requires_grad=True)
    E2 = mde.average_distortion(Xt2)
    E2.backward()
    assert torch.equal(Xt2.grad, Xt.grad) and torch.equal(E2.detach(), E.detach())


def _skewed_graph(rng, n, p, hubs):
    """Random graph whose degrees fall along the vertex order (endpoint ~ n u^3) plus a few hub vertices."""
    i = np.minimum((n * rng.random(p) ** 3).astype(np.int64), n - 1)
    j = (i + 1 + rng.integers(0, n - 1, p)) % n
    for k, (v, deg) in enumerate(hubs):
        sel = rng.choice(p, deg, replace=False)
        i[sel] = v
        j[sel] = (v + 1 + rng.choice(n - 1, deg, replace=False)) % n
    key = np.unique(np.minimum(i, j).astype(np.int64) * n + np.maximum(i, j))
    return np.stack([key // n, key % n], 1)


@pytest.mark.parametrize("d", [1, 2, 3, 4])
@pytest.mark.parametrize("mode", ["peel", "deal", "peel+deal"])
def test_ring_layout_peeled_and_dealt_rows_against_oracle(monkeypatch, d, mode):
    """The round-6 layouts forced on at a size the oracle checks fast: hub rows peeled off to the hub kernel
    (threshold 300 half-edges), row blocks dealt by degree, both; d = 1..4 (run-time and compile-time functors);
    one GPU (Q = 2 at d = 2: the in-launch sum of the two column groups) and a 3-way vertex-range shard (more
    column groups: k_ring_combine behind the kernel) -- loss and gradient against the oracle, two evaluations
    bitwise equal.  Functions: Log1p on a codebook stream, PushAndPull on continuous weights (mde_func.e0 feeds the hub
    rows), WeightedQuadratic (two per-edge arrays: e0 and e1), Quadratic with ONE scalar weight."""
    import pymde_amd
    from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate
    pen, los = pymde_amd.penalties, pymde_amd.losses
    monkeypatch.setenv("MDE_PANEL", "1")
    monkeypatch.setenv("MDE_RING_HUB", "300" if "peel" in mode else "0")
    monkeypatch.setenv("MDE_RING_PERMUTE", "1" if "deal" in mode else "0")
    rng = np.random.default_rng(100 + d)
    n, p = 40000, 500000
    edges = _skewed_graph(rng, n, p, [(7, 20000), (n // 2, 3000), (n - 3, 900)])
    p = len(edges)
    et = torch.tensor(edges, device=DEV)
    X = (rng.standard_normal((n, d)) * 1.5).astype(np.float32)
    Xd = torch.tensor(X, device=DEV)
    w2 = rng.choice(np.array([1.0, 2.0], dtype=np.float32), size=p)
    wc = np.where(rng.random(p) < 0.3, -rng.uniform(0.5, 1.5, p), rng.uniform(0.5, 2.0, p)).astype(np.float32)
    dev_ = rng.uniform(0.5, 3.0, p).astype(np.float32)
    wq = rng.uniform(0.2, 1.0, p).astype(np.float32)
    t = lambda a: torch.tensor(a, device=DEV)
    cases = [
        ("log1p codebook", pen.Log1p(t(w2)), oracle.func("LOG1P", w2, None, (1.5,))),
        ("pushpull fp32", pen.PushAndPull(t(wc), pen.Log1p, pen.Log), oracle.func("LOG1P", wc, None, (1.5,), "LOG", (1.0,))),
        ("weighted quadratic", los.WeightedQuadratic(t(dev_), t(wq)), oracle.func("L_WEIGHTED_QUADRATIC", dev_, wq, ())),
        ("scalar weight", pen.Quadratic(t(np.array([0.7], dtype=np.float32))),
         oracle.func("QUADRATIC", np.full(p, 0.7, dtype=np.float32), None, ())),
    ]
    for name, f, of in cases:
        wE, wgrad = oracle.average_distortion(edges, X, of)
        plan = EdgePlan(n, et)
        b = Binding(plan, f)
        buf = torch.zeros(n * d + 1, device=DEV)
        fused_evaluate(b, Xd, buf[:n * d].view(n, d), buf[n * d:])
        assert b.struct(d).layout == 1, name
        info = plan.ring_info()
        assert (info["hub_rows"] >= 3) == ("peel" in mode) and info["permuted"] == ("deal" in mode), (name, info)
        assert float(buf[n * d]) == pytest.approx(wE, rel=2e-5), (name, mode)
        assert_grad_close(buf[:n * d].view(n, d).cpu().numpy(), wgrad)
        buf2 = torch.zeros_like(buf)
        fused_evaluate(b, Xd, buf2[:n * d].view(n, d), buf2[n * d:])
        assert torch.equal(buf2, buf), name
        # forward only (no gradient buffer)
        lonly = torch.zeros(1, device=DEV)
        fused_evaluate(b, Xd, None, lonly)
        assert float(lonly) == pytest.approx(wE, rel=2e-5), name
        if name != "log1p codebook":
            continue
        # 3-way vertex-range shard: every rank's plan peels / deals its own rows
        total = torch.zeros_like(buf)
        bounds = [0, n // 5, n // 2, n]
        for r in range(3):
            part = torch.zeros_like(buf)
            fused_evaluate(Binding(EdgePlan(n, et, bounds[r], bounds[r + 1]), f), Xd, part[:n * d].view(n, d), part[n * d:])
            total += part
        assert float(total[n * d]) == pytest.approx(wE, rel=2e-5)
        assert_grad_close(total[:n * d].view(n, d).cpu().numpy(), wgrad)


def test_ring_layout_peeled_and_dealt_rows_on_a_dense_graph(monkeypatch):
    """The same round-6 layouts on a DENSE graph (1300 half-edges per row on average: the layout builder orders the
    entries by column first, and rebuilds its per-position slot array behind that sort): hub rows above 1500
    half-edges peeled, blocks dealt by degree; Huber loss on a byte-index stream; against the oracle, bitwise twice."""
    import pymde_amd
    from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate
    monkeypatch.setenv("MDE_PANEL", "1")
    monkeypatch.setenv("MDE_RING_HUB", "1500")
    monkeypatch.setenv("MDE_RING_PERMUTE", "1")
    rng = np.random.default_rng(9)
    n, p, d = 3000, 2_000_000, 2
    i = np.minimum((n * rng.random(p) ** 1.5).astype(np.int64), n - 1)      # degrees fall along the vertex order
    j = (i + 1 + rng.integers(0, n - 1, p)) % n
    key = np.unique(np.minimum(i, j).astype(np.int64) * n + np.maximum(i, j))
    edges = np.stack([key // n, key % n], 1)
    p = len(edges)
    dev_ = (1.0 + rng.integers(0, 30, p)).astype(np.float32)
    X = (rng.standard_normal((n, d)) * 3).astype(np.float32)
    f = pymde_amd.losses.Huber(torch.tensor(dev_, device=DEV), 1.0)
    plan = EdgePlan(n, torch.tensor(edges, device=DEV))
    b = Binding(plan, f)
    Xd = torch.tensor(X, device=DEV)
    buf = torch.zeros(n * d + 1, device=DEV)
    fused_evaluate(b, Xd, buf[:n * d].view(n, d), buf[n * d:])
    info = plan.ring_info()
    assert b.struct(d).layout == 1 and info["permuted"] and info["hub_rows"] > 0, info
    wE, wgrad = oracle.average_distortion(edges, X, oracle.func("L_HUBER", dev_, None, (1.0,)))
    assert float(buf[n * d]) == pytest.approx(wE, rel=2e-5)
    assert_grad_close(buf[:n * d].view(n, d).cpu().numpy(), wgrad)
    buf2 = torch.zeros_like(buf)
    fused_evaluate(b, Xd, buf2[:n * d].view(n, d), buf2[n * d:])
    assert torch.equal(buf2, buf)


@pytest.mark.parametrize("d", [2, 3])
def test_ring_kernel_every_public_function(monkeypatch, d):
    """Every public penalty and loss has a compile-time functor on the ring kernel since round 5 (the units
    mde_ring_k_penalty2 / mde_ring_k_loss2 added Power, Logistic, Sigmoid, Hinge, InvPower, LogRatio and the
    losses Cubic, Power, Logistic, Fractional, SoftFractional): each against the oracle on the ring kernel
    (forced on), with continuous parameters (fp32 stream), 3 distinct values (codebook) and 60 (byte index)."""
    import pymde_amd
    pen, los = pymde_amd.penalties, pymde_amd.losses
    monkeypatch.setenv("MDE_PANEL", "1")
    rng = np.random.default_rng(31 + d)
    n, p = 30000, 400000
    i = rng.integers(0, n, p)
    j = (i + 1 + rng.integers(0, n - 1, p)) % n
    key = np.unique(np.minimum(i, j).astype(np.int64) * n + np.maximum(i, j))
    edges = np.stack([key // n, key % n], 1)
    p = len(edges)
    et = torch.tensor(edges, device=DEV)
    X = (rng.standard_normal((n, d)) * 1.5).astype(np.float32)
    params = {"fp32": rng.uniform(0.5, 2.0, p).astype(np.float32),
              "codebook": rng.choice(np.array([0.5, 1.0, 2.0], dtype=np.float32), size=p),
              "byte index": (0.5 + rng.integers(0, 60, p) / 40.0).astype(np.float32)}
    cases = [
        ("LOGISTIC", lambda a: pen.Logistic(a, 0.5, 3.0), (0.5, 3.0), +1),
        ("SIGMOID", lambda a: pen.Sigmoid(a, 1.0, 2.0), (1.0, 2.0), +1),
        ("HINGE", lambda a: pen.Hinge(a, 1.0), (1.0, 0.5), +1),
        ("POWER", lambda a: pen.Power(a, 2.5), (2.5,), +1),
        ("POWER", lambda a: pen.Power(a, 1.5), (1.5,), +1),
        ("POWER", lambda a: pen.Power(a, 0.5), (0.5,), +1),
        ("INVPOWER", lambda a: pen.InvPower(a, 1), (1.0,), -1),
        ("INVPOWER", lambda a: pen.InvPower(a, 2.5), (2.5,), -1),
        ("LOGRATIO", lambda a: pen.LogRatio(a, 2), (2.0,), -1),
        ("LOGRATIO", lambda a: pen.LogRatio(a, 1.5), (1.5,), -1),
        ("L_CUBIC", lambda a: los.Cubic(a), (), +1),
        ("L_POWER", lambda a: los.Power(a, 1.5), (1.5,), +1),
        ("L_LOGISTIC", lambda a: los.Logistic(a), (), +1),
        ("L_FRACTIONAL", lambda a: los.Fractional(a), (), +1),
        ("L_SOFT_FRACTIONAL", lambda a: los.SoftFractional(a, 10.0), (10.0,), +1),
    ]
    for kind, make, scal, sign in cases:
        for stream, a in params.items():
            if kind not in ("LOGISTIC", "POWER", "L_CUBIC", "LOGRATIO") and stream != "fp32" and d == 3:
                continue  # (keep the d = 3 sweep short: every kind on one stream, four kinds on all three)
            a = (sign * a).astype(np.float32)
            f = make(torch.tensor(a, device=DEV))
            mde = pymde_amd.MDE(n, d, et, f)
            Xt = torch.tensor(X, device=DEV, requires_grad=True)
            E = mde.average_distortion(Xt)
            E.backward()
            b = mde._binding()
            assert b.struct(d).layout == 1 and b.stream_kind == stream, (kind, stream, b.stream_kind)
            wE, wgrad = oracle.average_distortion(edges, X, oracle.func(kind, a, None, scal))
            assert float(E.detach()) == pytest.approx(wE, rel=2e-5), (kind, scal, stream)
            assert_grad_close(Xt.grad.cpu().numpy(), wgrad)


def test_pushpull_on_the_ring_kernel_at_near_coincident_points(monkeypatch):
    """The merged PushAndPull(Log1p, Log) of the ring kernel keeps f'/d finite by construction and skips the
    NaN / Inf fix-up for codebook streams (mde_functions.h).  Repulsive AND attractive edges whose endpoints
    are 0, 1e-12, 1e-9, 1e-6 and 1e-3 apart, next to ordinary ones, against the oracle (the reference's rule:
    NaN / Inf -> 1, which at d = 0 multiplies x_i - x_j = 0): the gradient at the kernel tolerance; the loss
    with the coincident repulsive pairs left out (log(1 - exp(-0)) = -inf there, in the reference too)."""
    import pymde_amd
    pen = pymde_amd.penalties
    monkeypatch.setenv("MDE_PANEL", "1")
    rng = np.random.default_rng(77)
    n, p, d = 30000, 400000, 2
    i = rng.integers(0, n, p)
    j = (i + 1 + rng.integers(0, n - 1, p)) % n
    key = np.unique(np.minimum(i, j).astype(np.int64) * n + np.maximum(i, j))
    edges = np.stack([key // n, key % n], 1)
    p = len(edges)
    X = rng.standard_normal((n, d)).astype(np.float32) * 0.5
    w = rng.choice(np.array([-1.0, 1.0, 2.0], dtype=np.float32), size=p, p=[0.3, 0.4, 0.3])
    # near the origin float32 resolves tiny separations: pairs of vertices placed `gap` apart, joined by an edge
    gaps = [0.0, 1e-12, 1e-9, 1e-6, 1e-3]
    special = []
    for k, gap in enumerate(gaps):
        for sgn in (-1.0, 1.0):
            a, b = 2 * (2 * k + (sgn > 0)), 2 * (2 * k + (sgn > 0)) + 1   # vertices 0..19, distinct pairs
            X[a] = (1e-3 * (k + 1), 0.0)
            X[b] = (np.float32(1e-3 * (k + 1)) + np.float32(gap), 0.0) if gap >= 1e-9 else X[a]
            if 0.0 < gap < 1e-9:
                X[a] = (0.0, 0.0)
                X[b] = (np.float32(gap), 0.0)
            special.append((a, b, sgn, gap))
    have = set(map(tuple, edges.tolist()))
    add = np.array([[a, b] for a, b, _, _ in special if (a, b) not in have], dtype=np.int64)
    addw = np.array([sgn for a, b, sgn, _ in special if (a, b) not in have], dtype=np.float32)
    edges = np.concatenate([edges, add])
    w = np.concatenate([w, addw])
    f = pen.PushAndPull(torch.tensor(w, device=DEV), pen.Log1p, pen.Log)
    mde = pymde_amd.MDE(n, d, torch.tensor(edges, device=DEV), f)
    Xt = torch.tensor(X, device=DEV, requires_grad=True)
    E = mde.average_distortion(Xt)
    E.backward()
    b = mde._binding()
    assert b.struct(d).layout == 1 and b.codebook
    fd = oracle.func("LOG1P", w, None, (1.5,), "LOG", (1.0,))
    _, wgrad = oracle.average_distortion(edges, X, fd)
    got = Xt.grad.cpu().numpy()
    assert np.isfinite(got).all()
    # the rows of the near-coincident pairs, one by one (their entries span 20 orders of magnitude), then the rest
    for a, bb, sgn, gap in special:
        for v in (a, bb):
            np.testing.assert_allclose(got[v], wgrad[v], rtol=2e-4, atol=2e-5 * np.abs(wgrad[v]).max() + 1e-12,
                                       err_msg="gap %g, weight %g" % (gap, sgn))
    rest = np.ones(n, bool)
    rest[:20] = False
    assert_grad_close(got[rest], wgrad[rest])
    # the loss: without the coincident repulsive pair (-inf x w there, here and in the reference)
    keep = ~((np.linalg.norm(X[edges[:, 0]] - X[edges[:, 1]], axis=1) == 0) & (w < 0))
    assert np.isinf(float(E.detach())) or not (~keep).any()
    mde2 = pymde_amd.MDE(n, d, torch.tensor(edges[keep], device=DEV),
                         pen.PushAndPull(torch.tensor(w[keep], device=DEV), pen.Log1p, pen.Log))
    E2 = float(mde2.average_distortion(torch.tensor(X, device=DEV)))
    wE2, _ = oracle.average_distortion(edges[keep], X, oracle.func("LOG1P", w[keep], None, (1.5,), "LOG", (1.0,)), want_grad=False)
    assert E2 == pytest.approx(wE2, rel=1e-5)


def test_ring_layout_and_fold_options_agree_bitwise():
    """Round-5 options of the ring path against the default on one problem with two column groups per row block
    (where the groups' rows are added inside the launch): `MDE_RING_FOLD=0` (the k_ring_combine launch instead) and
    `MDE_RING_ASSIGN=1` (the sweep-balanced row -> wave map: a row's entries still reach its accumulator in chunk
    order, whichever wave owns it) must give the same bits, and the oracle's numbers."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import pymde_amd
rng = np.random.default_rng(3)
n, p, d = 600000, 6000000, 2
i = rng.integers(0, n, p); j = (i + 1 + rng.integers(0, n - 1, p)) %% n
e = np.stack([np.minimum(i, j), np.maximum(i, j)], 1)
w = (1.0 + (rng.random(p) < 0.3)).astype(np.float32)
X = rng.standard_normal((n, d)).astype(np.float32)
mde = pymde_amd.MDE(n, d, torch.tensor(e, device='cuda'), pymde_amd.penalties.Log1p(torch.tensor(w, device='cuda')))
Xt = torch.tensor(X, device='cuda', requires_grad=True)
E = mde.average_distortion(Xt); E.backward()
assert mde._binding().struct(d).layout == 1
np.save(sys.argv[1], np.concatenate([Xt.grad.cpu().numpy().ravel(), [float(E)]]))
"""
    root = str(__import__("conftest").ROOT)
    outs = {}
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        for name, env in (("default", {}), ("nofold", {"MDE_RING_FOLD": "0"}), ("assign", {"MDE_RING_ASSIGN": "1"})):
            path = os.path.join(td, name + ".npy")
            out = subprocess.run([sys.executable, "-c", code % root, path], env=dict(os.environ, MDE_PANEL="1", **env),
                                 capture_output=True, text=True, timeout=600)
            assert out.returncode == 0, (name, out.stderr[-2000:])
            outs[name] = np.load(path)
    assert np.array_equal(outs["default"], outs["nofold"]), "in-launch sum of the two column groups != k_ring_combine"
    assert np.array_equal(outs["default"][:-1], outs["assign"][:-1]), "balanced row -> wave map changed a gradient bit"
    np.testing.assert_allclose(outs["assign"][-1], outs["default"][-1], rtol=1e-6)
    # ... and they are the oracle's numbers
    rng = np.random.default_rng(3)
    n, p, d = 600000, 6000000, 2
    i = rng.integers(0, n, p)
    j = (i + 1 + rng.integers(0, n - 1, p)) % n
    e = np.stack([np.minimum(i, j), np.maximum(i, j)], 1)
    w = (1.0 + (rng.random(p) < 0.3)).astype(np.float32)
    X = rng.standard_normal((n, d)).astype(np.float32)
    wE, wgrad = oracle.average_distortion(e, X, oracle.func("LOG1P", w, None, (1.5,)))
    assert outs["default"][-1] == pytest.approx(wE, rel=1e-5)
    assert_grad_close(outs["default"][:-1].reshape(n, d), wgrad)


# ---------------------------------------------------------------- general-d kernel, pipelined form + processing order (round 6)
def _band_graph(rng, n, deg, window, shuffle):
    src = np.repeat(np.arange(n), deg)
    dst = (src + rng.integers(1, window + 1, n * deg)) % n
    e = np.stack([np.minimum(src, dst), np.maximum(src, dst)], 1)
    e = np.unique(e, axis=0)
    if shuffle:
        perm = rng.permutation(n)
        e = perm[e]
        e = np.stack([e.min(1), e.max(1)], 1)
    return e


@pytest.mark.parametrize("d", [5, 6, 7, 8, 9, 10, 11, 12, 13, 16, 17, 20, 24, 31, 32, 33, 40, 48, 50, 64, 96, 99, 100, 101,
                               127, 128, 129, 200, 255, 256, 257, 388, 511, 512])
def test_pipelined_wide_kernel_on_ragged_graphs(d, monkeypatch):
    """k_fused_wide4p (every d = 5 .. 512: 1 .. 32 lanes x 2 .. 4 float4s per half-edge at any float of a row -- 4-byte aligned 16-byte accesses --, the row's last column shifted back and masked where d is not a multiple of 4): empty rows, rows of one to three half-edges, a hub,
    odd and even step counts -- against the oracle, against the unpipelined kernel (MDE_WIDE_P=0), three runs bitwise."""
    import pymde_amd
    from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate
    rng = np.random.default_rng(d)
    n, p = 6007, 90000
    i = rng.integers(0, n, p)
    j = rng.integers(0, n, p)
    i[: p // 10] = 3                                   # a hub
    keep = (i != j) & (i % 7 != 5) & (j % 7 != 5)      # a seventh of the rows is empty
    e = np.unique(np.stack([np.minimum(i, j), np.maximum(i, j)], 1)[keep], axis=0)
    pp = e.shape[0]
    X = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    w = rng.uniform(0.5, 2.0, pp).astype(np.float32)
    fd = oracle.func("LOG1P", w, None, (1.5,))
    wE, wgrad = oracle.average_distortion(e, X, fd)
    et = torch.tensor(e, device=DEV)
    Xt = torch.tensor(X, device=DEV)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MDE_WIDE_P", mode)
        b = Binding(EdgePlan(n, et), pymde_amd.penalties.Log1p(torch.tensor(w, device=DEV)))
        runs = []
        for _ in range(3):
            buf = torch.zeros(n * d + 1, device=DEV)
            fused_evaluate(b, Xt, buf[:n * d].view(n, d), buf[n * d:])
            runs.append(buf)
        assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
        g = runs[0][:n * d].view(n, d).cpu().numpy()
        assert float(runs[0][n * d]) == pytest.approx(wE, rel=LOSS_RTOL)
        assert_grad_close(g, wgrad)
        assert np.all(g[np.arange(n) % 7 == 5] == 0.0)
        outs[mode] = g
    # the two kernels add a row's half-edges in different groupings: equal to rounding, not bit for bit
    assert_grad_close(outs["1"], outs["0"])


@pytest.mark.parametrize("window,shuffle", [(60, True), (60, False), (0, True)])
def test_row_processing_order(window, shuffle, monkeypatch):
    """mde_plan_row_order: on a band graph under a random renumbering the breadth-first order is adopted and brings
    the ends of an edge together; on the same graph as given, and on a random graph, it is not.  Whatever it does, the
    gradient is the gradient BIT FOR BIT (a row's sum does not depend on when the row is evaluated), and forcing an
    order (mode 2) on a vertex-range shard gives the single plan's rows."""
    import pymde_amd
    from pymde_amd.average_distortion import Binding, EdgePlan, fused_evaluate
    rng = np.random.default_rng(7)
    n, d, deg = 30011, 128, 12
    if window:
        e = _band_graph(rng, n, deg, window, shuffle)
    else:
        i = rng.integers(0, n, n * deg)
        j = (i + 1 + rng.integers(0, n - 1, n * deg)) % n
        e = np.unique(np.stack([np.minimum(i, j), np.maximum(i, j)], 1), axis=0)
    pp = e.shape[0]
    X = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    w = rng.uniform(0.5, 2.0, pp).astype(np.float32)
    et, Xt, wt = torch.tensor(e, device=DEV), torch.tensor(X, device=DEV), torch.tensor(w, device=DEV)

    def run(plan):
        b = Binding(plan, pymde_amd.penalties.Log1p(wt))
        buf = torch.zeros(n * d + 1, device=DEV)
        fused_evaluate(b, Xt, buf[:n * d].view(n, d), buf[n * d:])
        return buf

    monkeypatch.setenv("MDE_ROW_ORDER", "0")
    plain = EdgePlan(n, et)
    ref = run(plain)
    assert not plain.row_order(0)["in_use"]
    monkeypatch.delenv("MDE_ROW_ORDER")
    plan = EdgePlan(n, et)
    out = run(plan)                       # (the first evaluation at d = 128 builds the order)
    info = plan.row_order(1)
    if window and shuffle:
        assert info["in_use"] and info["mean_distance_after"] < 4 * window < info["mean_distance_before"] / 10
    else:
        assert not info["in_use"]
        if window:
            # rejected (the graph as given is as local as the search makes it): mode 2 overrides, the bits stay
            assert plan.row_order(2)["in_use"]
            assert torch.equal(run(plan)[:n * d], ref[:n * d])
            assert not plan.row_order(0)["in_use"]
    assert torch.equal(out[:n * d], ref[:n * d])
    assert float(out[n * d]) == pytest.approx(float(ref[n * d]), rel=1e-6)
    wE, wgrad = oracle.average_distortion(e, X, oracle.func("LOG1P", w, None, (1.5,)))
    assert float(out[n * d]) == pytest.approx(wE, rel=LOSS_RTOL)
    assert_grad_close(out[:n * d].view(n, d).cpu().numpy(), wgrad)
    # a forced order on two vertex-range shards: the rows of the single plan, bit for bit
    lo = 0
    for hi in (n // 3, n):
        shard = EdgePlan(n, et, row_lo=lo, row_hi=hi)
        # (a random graph's search is abandoned -- its third level holds most of the rows --: nothing to force)
        assert shard.row_order(2)["in_use"] == (bool(window) and hi - lo >= 8192)
        part = run(shard)
        assert torch.equal(part[lo * d:hi * d], ref[lo * d:hi * d])
        lo = hi
