"""Generate golden vectors by RUNNING THE REFERENCE (cvxgrp/pymde) on the CPU.

Run in the build container (it reads /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

It copies the reference to a scratch directory, builds its one Cython extension, stubs the two
imports that are not installed here (torchvision, pynndescent), imports it as ``pymde`` and
writes small ``.npz`` fixtures next to this file.  Everything is seeded; torch runs with one
thread so the reference's own fp32 summation order is fixed.

Fixtures
  functions.npz     (loss, grad, distortions) of every public penalty / loss, d in {1,2,3,8},
                    incl. coincident points (d_k = 0)         -> pins average_distortion.py:62-106
  constraints.npz   Centered / Standardized / Anchored maps   -> pins constraints.py, util.py:129-171
  trajectories.npz  per-iteration SolveStats of short embed() runs (optim.py / lbfgs.py)
  trajectories_mid.npz  the same at n = 20k, p ~ 300k (two problems x 4 perturbation levels, 8 iterations)
  trajectories_clusters.npz  the same on a structured problem: n = 100k in 100 planted clusters, p ~ 1.8M,
                    PushAndPull(Log1p, Log), Standardized (3 perturbation levels, 6 iterations)
  spectral.npz      quadratic.spectral on small graphs         -> pins quadratic.py
  preprocess.npz    deduplicate_edges / sample_edges of the reference
  cycle.npz         BASELINE config 1 scaled down: preserve_distances on a cycle graph,
                    losses.Quadratic, all pairs
  linesearch.npz    trial sequences of _strong_wolfe on scalar problems (lbfgs.py:44-253)
  sphere.npz        the private _Sphere constraint's two maps (constraints.py:203-231)
  api.npz           pca / align / rotate, k-NN with max_distance, graph k-NN (shortest-path and
                    direct), the graphs behind laplacian_embedding and preserve_neighbors(Graph)
"""
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"


def import_reference():
    scratch = os.path.join(tempfile.gettempdir(), "pymde_reference_scratch")
    if not os.path.exists(os.path.join(scratch, "pymde")):
        os.makedirs(scratch, exist_ok=True)
        for item in ("pymde", "setup.py", "README.md"):
            src = os.path.join(REFERENCE, item)
            dst = os.path.join(scratch, item)
            if os.path.isdir(src):
                shutil.copytree(src, dst)
            else:
                shutil.copy(src, dst)
        subprocess.check_call(["chmod", "-R", "u+w", scratch])
        subprocess.check_call([sys.executable, "setup.py", "build_ext", "--inplace"], cwd=scratch,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for name in ("torchvision", "torchvision.datasets", "torchvision.datasets.utils", "pynndescent"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
    sys.modules["torchvision.datasets"].utils = sys.modules["torchvision.datasets.utils"]
    sys.path.insert(0, scratch)
    import pymde
    return pymde


def random_graph(rng, n, p):
    """p distinct edges (i < j) of a random graph on n vertices."""
    all_pairs = np.stack(np.triu_indices(n, 1), axis=1)
    idx = rng.choice(all_pairs.shape[0], size=p, replace=False)
    return all_pairs[np.sort(idx)].astype(np.int64)


# name -> (constructor from (pymde, weights/deviations tensors), kind, per-edge array role,
#          scalars, kind_neg, scalars_neg)
def function_cases(pymde, torch, rng, p):
    pen, los = pymde.penalties, pymde.losses
    w_pos = torch.tensor(rng.uniform(0.5, 2.0, p).astype(np.float32))
    w_neg = -w_pos
    w_mix = torch.tensor(np.where(rng.random(p) < 0.4, -1.0, rng.uniform(0.5, 2.0, p)).astype(np.float32))
    dev = torch.tensor(rng.uniform(0.3, 2.5, p).astype(np.float32))
    w2 = torch.tensor(rng.uniform(0.2, 1.5, p).astype(np.float32))
    cases = [
        ("pen_linear", pen.Linear(w_pos), "LINEAR", w_pos, None, (), "NONE", ()),
        ("pen_quadratic", pen.Quadratic(w_pos), "QUADRATIC", w_pos, None, (), "NONE", ()),
        ("pen_cubic", pen.Cubic(w_pos), "CUBIC", w_pos, None, (), "NONE", ()),
        ("pen_power_2p5", pen.Power(w_pos, 2.5), "POWER", w_pos, None, (2.5,), "NONE", ()),
        ("pen_power_1", pen.Power(w_pos, 1.0), "POWER", w_pos, None, (1.0,), "NONE", ()),
        ("pen_huber", pen.Huber(w_pos, 0.5), "HUBER", w_pos, None, (0.5,), "NONE", ()),
        ("pen_logistic", pen.Logistic(w_pos, 0.2, 3.0), "LOGISTIC", w_pos, None, (0.2, 3.0), "NONE", ()),
        ("pen_sigmoid", pen.Sigmoid(w_pos, 0.5, 2.0), "SIGMOID", w_pos, None, (0.5, 2.0), "NONE", ()),
        ("pen_hinge", pen.Hinge(w_mix, 1.0, 0.25), "HINGE", w_mix, None, (1.0, 0.25), "NONE", ()),
        ("pen_log1p", pen.Log1p(w_pos), "LOG1P", w_pos, None, (1.5,), "NONE", ()),
        ("pen_log1p_e2", pen.Log1p(w_pos, 2.0), "LOG1P", w_pos, None, (2.0,), "NONE", ()),
        ("pen_log1p_e0p7", pen.Log1p(w_pos, 0.7), "LOG1P", w_pos, None, (0.7,), "NONE", ()),
        ("pen_log", pen.Log(w_neg), "LOG", w_neg, None, (1.0,), "NONE", ()),
        ("pen_log_e2", pen.Log(w_neg, 2.0), "LOG", w_neg, None, (2.0,), "NONE", ()),
        ("pen_invpower", pen.InvPower(w_neg), "INVPOWER", w_neg, None, (1.0,), "NONE", ()),
        ("pen_invpower_e2", pen.InvPower(w_neg, 2), "INVPOWER", w_neg, None, (2.0,), "NONE", ()),
        ("pen_logratio", pen.LogRatio(w_neg), "LOGRATIO", w_neg, None, (2.0,), "NONE", ()),
        ("pen_pushpull_default", pen.PushAndPull(w_mix), "LOG1P", w_mix, None, (1.5,), "LOGRATIO", (2.0,)),
        ("pen_pushpull_log", pen.PushAndPull(w_mix, pen.Log1p, pen.Log), "LOG1P", w_mix, None, (1.5,),
         "LOG", (1.0,)),
        ("pen_pushpull_quad_invpower", pen.PushAndPull(w_mix, pen.Quadratic, pen.InvPower), "QUADRATIC",
         w_mix, None, (), "INVPOWER", (1.0,)),
        ("loss_quadratic", los.Quadratic(dev), "L_QUADRATIC", dev, None, (), "NONE", ()),
        ("loss_weighted_quadratic", los.WeightedQuadratic(dev), "L_WEIGHTED_QUADRATIC", dev,
         1.0 / dev.pow(2), (), "NONE", ()),
        ("loss_weighted_quadratic_w", los.WeightedQuadratic(dev, w2), "L_WEIGHTED_QUADRATIC", dev, w2,
         (), "NONE", ()),
        ("loss_huber", los.Huber(dev, 0.4), "L_HUBER", dev, None, (0.4,), "NONE", ()),
        ("loss_cubic", los.Cubic(dev), "L_CUBIC", dev, None, (), "NONE", ()),
        ("loss_power", los.Power(dev, 1.7), "L_POWER", dev, None, (1.7,), "NONE", ()),
        ("loss_absolute", los.Absolute(dev), "L_ABSOLUTE", dev, None, (), "NONE", ()),
        ("loss_logistic", los.Logistic(dev), "L_LOGISTIC", dev, None, (), "NONE", ()),
        ("loss_fractional", los.Fractional(dev), "L_FRACTIONAL", dev, None, (), "NONE", ()),
        ("loss_soft_fractional", los.SoftFractional(dev, 5.0), "L_SOFT_FRACTIONAL", dev, None, (5.0,),
         "NONE", ()),
        # the reference's private kinds (penalties.py:134-160,174-188; losses.py:90-98,151-163,232-239)
        ("pen_deadzone_quadratic", pen._DeadzoneQuadratic(w_pos, 1.0), "DEADZONE_QUADRATIC", w_pos, None,
         (1.0,), "NONE", ()),
        ("pen_deadzone_cubic", pen._DeadzoneCubic(w_pos, 1.2), "DEADZONE_CUBIC", w_pos, None, (1.2,),
         "NONE", ()),
        ("pen_clipped_quadratic", pen._ClippedQuadratic(w_pos, torch.tensor(0.4)), "CLIPPED_QUADRATIC", w_pos, None,
         (0.4,), "NONE", ()),
        ("loss_clipped_quadratic", los._ClippedQuadratic(dev, torch.tensor(0.1)), "L_CLIPPED_QUADRATIC", dev, None,
         (0.1,), "NONE", ()),
        ("loss_weighted_power", los._WeightedPower(dev, 1.7, w2), "L_WEIGHTED_POWER", dev, w2, (1.7,),
         "NONE", ()),
        ("loss_weighted_power_default", los._WeightedPower(dev, 2.0), "L_WEIGHTED_POWER", dev,
         1.0 / dev.pow(2), (2.0,), "NONE", ()),
        # (an even exponent: the reference's (d - delta).pow(e) is NaN for d < delta otherwise)
        ("loss_log1p_e2", los._Log1p(dev, 2.0), "L_LOG1P", dev, None, (2.0,), "NONE", ()),
    ]
    return cases


def gen_functions(pymde, torch):
    rng = np.random.default_rng(1234)
    n, p = 40, 160
    edges = random_graph(rng, n, p)
    out = {"edges": edges, "n": n}
    names = []
    for d in (1, 2, 3, 8):
        X = rng.standard_normal((n, d)).astype(np.float32)
        out["X_d%d" % d] = X
    # a variant with coincident points (rows 0..3 equal -> some d_k = 0 exactly)
    Xz = rng.standard_normal((n, 2)).astype(np.float32)
    Xz[edges[:6, 1]] = Xz[edges[:6, 0]]
    out["X_zero"] = Xz
    cases = function_cases(pymde, torch, np.random.default_rng(99), p)
    for name, f, kind, a0, a1, sc, kind_neg, sc_neg in cases:
        names.append(name)
        out[name + "__kind"] = np.array(kind)
        out[name + "__kind_neg"] = np.array(kind_neg)
        out[name + "__a0"] = a0.numpy()
        if a1 is not None:
            out[name + "__a1"] = a1.numpy()
        out[name + "__scalars"] = np.array(sc, dtype=np.float64)
        out[name + "__scalars_neg"] = np.array(sc_neg, dtype=np.float64)
        for tag in ("d1", "d2", "d3", "d8", "zero"):
            X = out["X_zero"] if tag == "zero" else out["X_" + tag]
            d = X.shape[1]
            mde = pymde.MDE(n, d, torch.tensor(edges), f)
            Xt = torch.tensor(X, requires_grad=True)
            E = mde.average_distortion(Xt)
            E.backward()
            out["%s__%s__loss" % (name, tag)] = np.array(E.item(), dtype=np.float64)
            out["%s__%s__grad" % (name, tag)] = Xt.grad.numpy().copy()
            with torch.no_grad():
                out["%s__%s__distortions" % (name, tag)] = mde.distortions(torch.tensor(X)).numpy()
                if name == "pen_quadratic":
                    out["%s__distances" % tag] = mde.distances(torch.tensor(X)).numpy()
                    out["%s__differences" % tag] = mde.differences(torch.tensor(X)).numpy()
    out["names"] = np.array(names)
    # the reference's own known-answer test (pymde/test_optim.py:75-93): 62/3
    e3 = np.array([[0, 1], [0, 2], [1, 2]])
    mde = pymde.MDE(3, 2, e3, pymde.penalties.Quadratic(torch.tensor([1.0, 2.0, 3.0])),
                    constraint=pymde.Standardized())
    out["kat_62_3"] = np.array(
        mde.average_distortion(torch.tensor([[0.0, 0.0], [1.0, 1.0], [3.0, 3.0]])).item())
    # distances backward at coincident points (test_optim.py:57-71)
    Xo = torch.ones((3, 3), requires_grad=True)
    mde = pymde.MDE(3, 3, np.array([[0, 1]]), pymde.penalties.Quadratic(torch.ones(1)))
    mde.distances(Xo).backward()
    out["norm_grad_zero"] = Xo.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "functions.npz"), **out)
    print("functions.npz:", len(names), "functions")


def gen_constraints(pymde, torch):
    rng = np.random.default_rng(7)
    out = {}
    shapes = [(5, 3), (10, 3), (100, 3), (500, 2), (300, 8), (400, 32), (200, 40)]
    out["shapes"] = np.array(shapes)
    for (n, d) in shapes:
        X = rng.standard_normal((n, d)).astype(np.float32) * rng.uniform(0.5, 2.0, d).astype(np.float32)
        X += rng.uniform(-1, 1, d).astype(np.float32)
        Z = rng.standard_normal((n, d)).astype(np.float32)
        tag = "%dx%d" % (n, d)
        out["X_" + tag] = X
        out["Z_" + tag] = Z
        P = pymde.Standardized().project_onto_constraint(torch.tensor(X), inplace=False)
        out["std_retract_" + tag] = P.numpy()
        out["std_tangent_" + tag] = pymde.Standardized().project_onto_tangent_space(
            P, torch.tensor(Z), inplace=False).numpy()
        out["centered_" + tag] = pymde.Centered().project_onto_constraint(
            torch.tensor(X), inplace=False).numpy()
    n, d = 50, 2
    anchors = torch.tensor(rng.choice(n, 7, replace=False))
    values = torch.tensor(rng.standard_normal((7, d)).astype(np.float32))
    Z = rng.standard_normal((n, d)).astype(np.float32)
    c = pymde.Anchored(anchors, values)
    out["anchors"] = anchors.numpy()
    out["anchor_values"] = values.numpy()
    out["anchor_Z"] = Z
    out["anchor_tangent"] = c.project_onto_tangent_space(None, torch.tensor(Z), inplace=False).numpy()
    out["anchor_retract"] = c.project_onto_constraint(torch.tensor(Z), inplace=False).numpy()
    np.savez_compressed(os.path.join(HERE, "constraints.npz"), **out)
    print("constraints.npz written")


def gen_trajectories(pymde, torch):
    rng = np.random.default_rng(2024)
    pen, los = pymde.penalties, pymde.losses
    n, p = 300, 3000
    edges = random_graph(rng, n, p)
    w_pos = rng.uniform(0.5, 2.0, p).astype(np.float32)
    w_mix = np.where(np.arange(p) < 2 * p // 3, w_pos, -1.0).astype(np.float32)
    dev = rng.uniform(0.5, 2.0, p).astype(np.float32)
    anchors = rng.choice(n, 20, replace=False)
    anchor_vals = rng.standard_normal((20, 2)).astype(np.float32)
    out = {"edges": edges, "n": n, "w_pos": w_pos, "w_mix": w_mix, "dev": dev, "anchors": anchors,
           "anchor_values": anchor_vals}
    problems = [
        ("quad_std", 2, "QUADRATIC", lambda: pen.Quadratic(torch.tensor(w_pos)), "standardized"),
        ("log1p_centered", 2, "LOG1P", lambda: pen.Log1p(torch.tensor(w_pos)), "centered"),
        ("pushpull_std", 2, "LOG1P|LOG", lambda: pen.PushAndPull(torch.tensor(w_mix), pen.Log1p, pen.Log),
         "standardized"),
        ("pushpull_centered_d3", 3, "LOG1P|LOGRATIO", lambda: pen.PushAndPull(torch.tensor(w_mix)), "centered"),
        ("absolute_centered", 2, "L_ABSOLUTE", lambda: los.Absolute(torch.tensor(dev)), "centered"),
        ("huber_std", 2, "L_HUBER", lambda: los.Huber(torch.tensor(dev), 0.5), "standardized"),
        ("quadloss_anchored", 2, "L_QUADRATIC", lambda: los.Quadratic(torch.tensor(dev)), "anchored"),
    ]
    names = []
    # The reference's line search is chaotic at fp32 noise level: on the first iterations the
    # step along d is tiny, the cubic-interpolation discriminant (lbfgs.py:31-32) is ~0 and its
    # sign -- i.e. whether the next trial is 10 t or the bisection 5.5 t -- is decided by
    # rounding.  Perturbing X0 by 1e-7 .. 1e-5 (relative) makes the reference itself hop between a few
    # distinct trajectories, so the fixture records a small ENSEMBLE of reference runs
    # (trial 0 = unperturbed) and parity means "matches one member".
    TRIALS = 12
    NOISE = [0.0, 1e-7, 1e-6, 1e-5]  # relative perturbation of X0 per trial (trial % 4)
    for name, d, kinds, make_f, cname in problems:
        torch.manual_seed(0)
        if cname == "standardized":
            c = pymde.Standardized()
        elif cname == "centered":
            c = pymde.Centered()
        else:
            c = pymde.Anchored(torch.tensor(anchors), torch.tensor(anchor_vals))
        X0 = c.initialization(n, d)
        names.append(name)
        out[name + "__d"] = d
        out[name + "__kinds"] = np.array(kinds)
        out[name + "__constraint"] = np.array(cname)
        out[name + "__X0"] = X0.numpy()
        E, R, S, F = [], [], [], []
        for trial in range(TRIALS):
            gen = torch.Generator().manual_seed(100 + trial)
            Xs = X0 * (1 + NOISE[trial % 4] * torch.randn(X0.shape, generator=gen))
            if cname == "anchored":
                Xs[torch.tensor(anchors)] = torch.tensor(anchor_vals)
            mde = pymde.MDE(n, d, torch.tensor(edges), make_f(), constraint=c)
            mde.embed(X=Xs, max_iter=12, eps=1e-9, memory_size=5)
            s = mde.solve_stats
            pad = lambda v: np.pad(np.array(v, dtype=np.float64), (0, 12 - len(v)), constant_values=np.nan)
            E.append(pad(s.average_distortions))
            R.append(pad(s.residual_norms))
            S.append(pad(s.step_size_percents))
            mde2 = pymde.MDE(n, d, torch.tensor(edges), make_f(), constraint=c)
            mde2.embed(X=Xs, max_iter=150, eps=1e-6, memory_size=10)
            F.append(mde2.value)
        out[name + "__distortions"] = np.stack(E)
        out[name + "__residuals"] = np.stack(R)
        out[name + "__steps"] = np.stack(S)
        out[name + "__final_value_150"] = np.array(F)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "trajectories.npz"), **out)
    print("trajectories.npz:", names)


def mid_problem_arrays(which):
    """Numpy-seeded inputs of the two mid-size trajectory cases (the test regenerates them with the same
    calls; the fixture stores only X0, the reference's statistics and a checksum of the edges).
    'neighbors': the config-2 stand-in scaled to n = 20k -- 10 'neighbour' edges (weights 1 / 2) and 5
    'dissimilar' edges (weight -1) per item, PushAndPull(Log1p, Log), Standardized.  'distances': a
    preserve_distances-shaped problem, 300k random pairs with target distances, losses.Huber(0.5),
    Centered."""
    n = 20000
    rng = np.random.default_rng(4242 if which == "neighbors" else 4343)
    deg = 15
    src = np.repeat(np.arange(n), deg)
    dst = rng.integers(0, n - 1, n * deg)
    dst += dst >= src
    e = np.stack([np.minimum(src, dst), np.maximum(src, dst)], 1)
    key = np.unique(e[:, 0].astype(np.int64) * n + e[:, 1])
    edges = np.stack([key // n, key % n], 1)
    p = len(edges)
    if which == "neighbors":
        w = np.where(rng.random(p) < 2.0 / 3.0, 1.0 + (rng.random(p) < 0.3), -1.0).astype(np.float32)
        return n, edges, w
    dev = rng.uniform(0.5, 3.0, p).astype(np.float32)
    return n, edges, dev


def gen_trajectories_mid(pymde, torch):
    """First 8 iterations of the reference's embed() on two problems of realistic size (n = 20k,
    p ~ 300k), from X0 and from X0 perturbed by 1e-7 / 1e-6 / 1e-5 (relative): SURVEY 8c's
    "first iterations within rtol 1e-3 of the oracle run with identical init" at scale."""
    torch.set_num_threads(8)  # (the summation order of scatter_add_ on 300k edges is thread-count dependent only in the last bits)
    out = {}
    NOISE = [0.0, 1e-7, 1e-6, 1e-5]
    for which in ("neighbors", "distances"):
        n, edges, par = mid_problem_arrays(which)
        out[which + "__edge_checksum"] = np.array([int(edges[:, 0].sum()), int(edges[:, 1].sum()), len(edges)])
        out[which + "__param_checksum"] = np.array([float(np.abs(par).astype(np.float64).sum())])
        torch.manual_seed(0)
        if which == "neighbors":
            c = pymde.Standardized()
            make_f = lambda: pymde.penalties.PushAndPull(torch.tensor(par), pymde.penalties.Log1p, pymde.penalties.Log)
        else:
            c = pymde.Centered()
            make_f = lambda: pymde.losses.Huber(torch.tensor(par), 0.5)
        X0 = c.initialization(n, 2)
        out[which + "__X0"] = X0.numpy()
        E, R, S = [], [], []
        for trial, noise in enumerate(NOISE):
            gen = torch.Generator().manual_seed(200 + trial)
            Xs = X0 * (1 + noise * torch.randn(X0.shape, generator=gen))
            mde = pymde.MDE(n, 2, torch.tensor(edges), make_f(), constraint=c)
            mde.embed(X=Xs, max_iter=8, eps=1e-12, memory_size=10)
            st = mde.solve_stats
            pad = lambda v: np.pad(np.array(v, dtype=np.float64), (0, 8 - len(v)), constant_values=np.nan)
            E.append(pad(st.average_distortions))
            R.append(pad(st.residual_norms))
            S.append(pad(st.step_size_percents))
            print(which, "noise", noise, E[-1][:4])
        out[which + "__distortions"] = np.stack(E)
        out[which + "__residuals"] = np.stack(R)
        out[which + "__steps"] = np.stack(S)
    out["noise"] = np.array(NOISE)
    np.savez_compressed(os.path.join(HERE, "trajectories_mid.npz"), **out)
    print("trajectories_mid.npz written")


def cluster_problem_arrays():
    """A STRUCTURED problem of config-4 kind at a size the reference runs in seconds: n = 100k items in 100 planted
    clusters of 1000; every item draws 12 'neighbour' edges inside its cluster (weights 1 / 2) and 6 'dissimilar'
    edges anywhere (weight -1); PushAndPull(Log1p, Log), Standardized.  (The uniform-random graph of SURVEY 8d has
    nothing to learn under unit covariance -- its Standardized solve is flat; this one separates the clusters.)"""
    n, csize = 100_000, 1000
    rng = np.random.default_rng(777)
    src = np.repeat(np.arange(n), 12)
    dst = (src // csize) * csize + rng.integers(0, csize - 1, src.size)
    dst += dst >= src
    rs = np.repeat(np.arange(n), 6)
    rd = rng.integers(0, n - 1, rs.size)
    rd += rd >= rs
    a = np.concatenate([src, rs])
    b = np.concatenate([dst, rd])
    w = np.concatenate([1.0 + (rng.random(src.size) < 0.3), -np.ones(rs.size)]).astype(np.float32)
    key = np.minimum(a, b).astype(np.int64) * n + np.maximum(a, b)
    _, first = np.unique(key, return_index=True)   # (a pair drawn twice keeps its first weight)
    first.sort()
    edges = np.stack([np.minimum(a, b)[first], np.maximum(a, b)[first]], 1)
    return n, edges, w[first]


def gen_trajectories_clusters(pymde, torch):
    """First 6 iterations of the reference's embed() on the planted-cluster problem (n = 100k, p ~ 1.8M), from X0
    and from X0 perturbed by 1e-7 / 1e-6.  X0 is regenerated by the test (numpy seed, Standardized by the float64
    recipe below); the fixture keeps a checksum of it."""
    torch.set_num_threads(8)
    n, edges, w = cluster_problem_arrays()
    X0 = cluster_X0(n)
    out = {"edge_checksum": np.array([int(edges[:, 0].sum()), int(edges[:, 1].sum()), len(edges)]),
           "param_checksum": np.array([float(np.abs(w).astype(np.float64).sum())]),
           "X0_checksum": np.array([float(np.abs(X0).astype(np.float64).sum())])}
    NOISE = [0.0, 1e-7, 1e-6]
    E, R, S = [], [], []
    for trial, noise in enumerate(NOISE):
        gen = torch.Generator().manual_seed(300 + trial)
        Xs = torch.tensor(X0) * (1 + noise * torch.randn((n, 2), generator=gen))
        f = pymde.penalties.PushAndPull(torch.tensor(w), pymde.penalties.Log1p, pymde.penalties.Log)
        mde = pymde.MDE(n, 2, torch.tensor(edges), f, constraint=pymde.Standardized())
        mde.embed(X=Xs, max_iter=6, eps=1e-12, memory_size=10)
        st = mde.solve_stats
        pad = lambda v: np.pad(np.array(v, dtype=np.float64), (0, 6 - len(v)), constant_values=np.nan)
        E.append(pad(st.average_distortions))
        R.append(pad(st.residual_norms))
        S.append(pad(st.step_size_percents))
        print("clusters noise", noise, E[-1])
    out.update(distortions=np.stack(E), residuals=np.stack(R), steps=np.stack(S), noise=np.array(NOISE))
    np.savez_compressed(os.path.join(HERE, "trajectories_clusters.npz"), **out)
    print("trajectories_clusters.npz written")


def cluster_X0(n):
    """A Standardized starting point from a numpy seed: centred, then X (X^T X / n)^{-1/2} in float64."""
    rng = np.random.default_rng(778)
    X = rng.standard_normal((n, 2))
    X -= X.mean(0)
    lam, Q = np.linalg.eigh(X.T @ X / n)
    return (X @ (Q / np.sqrt(lam)) @ Q.T).astype(np.float32)


def gen_spectral(pymde, torch):
    from pymde import quadratic
    out = {}
    # the reference's own test problem (pymde/test_quadratic.py:67-109): n = 12, d = 3
    torch.manual_seed(0)
    np.random.seed(0)
    n, m = 12, 3
    edges = pymde.all_edges(n)
    weights = torch.tensor(np.random.default_rng(5).uniform(0.1, 1.0, edges.shape[0]).astype(np.float32))
    emb = quadratic.spectral(n, m, edges, weights)
    out["small_edges"], out["small_weights"], out["small_emb"] = edges.numpy(), weights.numpy(), emb.numpy()
    rng = np.random.default_rng(11)
    n, m = 400, 2
    e = random_graph(rng, n, 3200)
    w = rng.uniform(0.5, 1.5, e.shape[0]).astype(np.float32)
    emb = quadratic.spectral(n, m, torch.tensor(e), torch.tensor(w))
    out["mid_edges"], out["mid_weights"], out["mid_emb"] = e, w, emb.numpy()
    mde = pymde.MDE(n, m, torch.tensor(e), pymde.penalties.Quadratic(torch.tensor(w)),
                    constraint=pymde.Standardized())
    out["mid_value"] = np.array(mde.average_distortion(emb).item())
    np.savez_compressed(os.path.join(HERE, "spectral.npz"), **out)
    print("spectral.npz written")


def gen_cycle(pymde, torch):
    """BASELINE config 1 scaled down (n = 300 instead of 5000 so the fixture stays small):
    preserve_distances on a cycle graph, Quadratic loss, all pairs, d = 2."""
    n = 300
    cyc = np.array([[i, (i + 1) % n] for i in range(n)])
    graph = pymde.Graph.from_edges(torch.tensor(cyc))
    torch.manual_seed(0)
    mde = pymde.preserve_distances(graph, embedding_dim=2, loss=pymde.losses.Quadratic)
    X0 = mde.constraint.initialization(n, 2)
    E, R, F = [], [], []
    for trial in range(12):  # ensemble of reference runs, see gen_trajectories
        gen = torch.Generator().manual_seed(200 + trial)
        Xs = X0 * (1 + [0.0, 1e-7, 1e-6, 1e-5][trial % 4] * torch.randn(X0.shape, generator=gen))
        mde.embed(X=Xs, max_iter=40, eps=1e-8)
        E.append(np.pad(np.array(mde.solve_stats.average_distortions), (0, 40 - mde.solve_stats.iterations),
                        constant_values=np.nan))
        R.append(np.pad(np.array(mde.solve_stats.residual_norms), (0, 40 - mde.solve_stats.iterations),
                        constant_values=np.nan))
        F.append(mde.value)
    out = {"n": n, "edges": mde.edges.numpy(), "deviations": mde.distortion_function.deviations.numpy(),
           "X0": X0.numpy(), "distortions": np.stack(E), "residuals": np.stack(R),
           "final_value": np.array(F)}
    np.savez_compressed(os.path.join(HERE, "cycle.npz"), **out)
    print("cycle.npz: p =", out["edges"].shape[0], "final", out["final_value"])


def gen_preprocess(pymde, torch):
    """deduplicate_edges outputs (exact) and sample_edges invariants of the reference (f1)."""
    from pymde.preprocess import preprocess as pp
    rng = np.random.default_rng(77)
    n = 500
    e = rng.integers(0, n, (6000, 2))
    e = e[e[:, 0] != e[:, 1]]
    e = np.concatenate([e, e[:700][:, ::-1], e[:300]])     # flipped and repeated rows
    out = {"n": n, "edges": e, "dedup": pp.deduplicate_edges(torch.tensor(e)).numpy()}
    excl = out["dedup"][:2000]
    s = pp.sample_edges(n, 5000, exclude=torch.tensor(excl), seed=3).numpy()
    out["exclude"] = excl
    out["ref_sample_count"] = np.array(len(s))
    out["ref_sample"] = s
    # preserve_distances on a small data matrix with every pair retained (deterministic)
    data = rng.standard_normal((40, 5)).astype(np.float32)
    for cname, c in (("centered", None), ("standardized", pymde.Standardized())):
        mde = pymde.preserve_distances(torch.tensor(data), embedding_dim=2, loss=pymde.losses.Absolute,
                                       constraint=c)
        out["pd_edges_" + cname] = mde.edges.numpy()
        out["pd_deviations_" + cname] = mde.distortion_function.deviations.numpy()
    out["pd_data"] = data
    # k-NN graph of the reference (sklearn brute-force branch, n < 10 000); the stub for the
    # un-installed pynndescent only has to be importable
    sys.modules["pynndescent"].NNDescent = None
    kd = rng.standard_normal((600, 20)).astype(np.float32)
    kg = pymde.preprocess.k_nearest_neighbors(torch.tensor(kd), k=15)
    out["knn_data"] = kd
    out["knn_edges"] = kg.edges.numpy()
    out["knn_weights"] = kg.weights.numpy()
    np.savez_compressed(os.path.join(HERE, "preprocess.npz"), **out)
    print("preprocess.npz: dedup", out["dedup"].shape, "sampled", len(s))


def gen_api(pymde, torch):
    """The remaining public helpers around the hot path: pca / align / rotate (util.py,
    quadratic.py:16-44) and the neighbour graphs of the recipes (graph.py:502-587,
    data_matrix.py:91-178 with max_distance, recipes.py:221-503)."""
    import scipy.sparse as sp
    from pymde.preprocess import data_matrix, graph as rgraph
    sys.modules["pynndescent"].NNDescent = None
    rng = np.random.default_rng(2024)
    out = {}
    Y = (rng.standard_normal((50, 6)) * np.array([5, 3, 2, 1, 0.5, 0.2])).astype(np.float32)
    out["pca_Y"] = Y
    out["pca_out"] = pymde.pca(torch.tensor(Y), 3).numpy()
    src = rng.standard_normal((40, 3)).astype(np.float32) * np.array([2.0, 1.0, 0.5], dtype=np.float32) + 3.0
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    tgt = ((src - src.mean(0)) @ q + 0.05 * rng.standard_normal((40, 3))).astype(np.float32)
    out["align_source"], out["align_target"] = src, tgt
    out["align_out"] = pymde.align(torch.tensor(src), torch.tensor(tgt)).numpy()
    out["procrustes_out"] = pymde.util.procrustes(torch.tensor(src), torch.tensor(tgt)).numpy()
    X2 = rng.standard_normal((30, 2)).astype(np.float32)
    X3 = rng.standard_normal((30, 3)).astype(np.float32)
    out["rot_X2"], out["rot_X3"] = X2, X3
    out["rot2_out"] = pymde.rotate(torch.tensor(X2), torch.tensor(30.0)).numpy()
    out["rot3_out"] = pymde.rotate(torch.tensor(X3), torch.tensor([10.0, 20.0, 30.0])).numpy()
    # k-NN of a data matrix with a radius
    kd = rng.standard_normal((60, 5)).astype(np.float32)
    g = data_matrix.k_nearest_neighbors(kd, k=4, max_distance=1.6)
    out["knnr_data"] = kd
    out["knnr_edges"], out["knnr_weights"] = g.edges.numpy(), g.distances.numpy()
    # k-NN on a weighted graph with distinct lengths (no distance ties)
    n = 40
    rows = rng.integers(0, n, 140)
    cols = rng.integers(0, n, 140)
    keep = rows != cols
    rows, cols = rows[keep], cols[keep]
    A = sp.coo_matrix((rng.uniform(0.5, 2.0, rows.size), (rows, cols)), shape=(n, n)).tocsr()
    A = A.maximum(A.T)
    G = pymde.Graph(A.copy())
    out["g_n"] = np.array(n)
    out["g_edges"], out["g_lengths"] = G.edges.numpy(), G.distances.numpy()
    for tag, kw in (("sp", {"graph_distances": True}),
                    ("spr", {"graph_distances": True, "max_distance": 1.5}),
                    ("direct", {"graph_distances": False}),
                    ("directr", {"graph_distances": False, "max_distance": 1.2})):
        # (a fresh Graph per call: the reference's max_distance masking writes into the adjacency)
        kg = rgraph.k_nearest_neighbors(pymde.Graph(A.copy()), k=3, **kw)
        out["gknn_%s_edges" % tag] = kg.edges.numpy()
        out["gknn_%s_weights" % tag] = kg.distances.numpy()
    # the graphs the recipes build
    lap = pymde.laplacian_embedding(torch.tensor(kd), embedding_dim=2, n_neighbors=5, init="random")
    out["lap_edges"] = lap.edges.numpy()
    out["lap_weights"] = lap.distortion_function.weights.numpy()
    out["lap_constraint"] = np.array(type(lap.constraint).__name__)
    pn = pymde.preserve_neighbors(pymde.Graph(A.copy()), embedding_dim=2, n_neighbors=3, init="random")
    w = pn.distortion_function.weights.numpy()
    out["png_edges_pos"] = pn.edges.numpy()[w > 0]
    out["png_weights_pos"] = w[w > 0]
    out["png_n_neg"] = np.array(int((w < 0).sum()))
    np.savez_compressed(os.path.join(HERE, "api.npz"), **out)
    print("api.npz:", {k: np.asarray(v).shape for k, v in out.items() if k.endswith("edges")})


def linesearch_problems():
    """Scalar problems phi(t) -> (f, f') for the strong-Wolfe trace; shared with the test."""
    nan = float("nan")

    def quartic(t):
        return (t - 0.25) ** 2 * (t + 1) ** 2, 2 * (t - 0.25) * (t + 1) * (2 * t + 0.75)

    def more_thuente(t):  # -t / (t^2 + 2): long extrapolation phase from a tiny first step
        return -t / (t * t + 2.0), (t * t - 2.0) / (t * t + 2.0) ** 2

    def steep(t):  # Armijo fails at t = 1, zoom with the 10 % safeguard
        return 1.0 - t + 40.0 * t ** 4, -1.0 + 160.0 * t ** 3

    def nan_beyond(t):  # not finite for t > 0.3: the halving prologue
        return ((t - 0.2) ** 2, 2 * (t - 0.2)) if t <= 0.3 else (nan, nan)

    def nan_gap(t):  # finite at t >= 1 and t <= 0.01 only: zoom walks into NaNs, 0.8 back-off recovers
        if t >= 1.0:
            return 10.0, 1.0
        if t <= 0.01:
            return 1.0 - t, -1.0
        return nan, nan

    def nan_below(t):  # finite at t = 0 and t >= 1 only: the back-off ends at t = 0
        if t >= 1.0:
            return 10.0, 1.0
        if t == 0.0:
            return 1.0, -1.0
        return nan, nan

    return [("quartic", quartic, 1.0), ("more_thuente_small", more_thuente, 1e-3),
            ("more_thuente_large", more_thuente, 10.0), ("steep", steep, 1.0),
            ("nan_beyond", nan_beyond, 1.0), ("nan_gap", nan_gap, 1.0), ("nan_below", nan_below, 1.0)]


def gen_linesearch(pymde, torch):
    """Trial sequences of the reference's _strong_wolfe (lbfgs.py:44-253) on scalar problems
    (x = 0, d = 1, so the trial point IS t), in float64."""
    from pymde import lbfgs as ref_lbfgs
    out = {"names": np.array([p[0] for p in linesearch_problems()])}
    for name, fn, t0 in linesearch_problems():
        trials = []

        def obj_func(x, t, d):
            trials.append(float(t))
            f, g = fn(float(t))
            return f, torch.tensor([g], dtype=torch.float64)
        f0, g0 = fn(0.0)
        x = [torch.zeros(1, dtype=torch.float64)]
        d = torch.ones(1, dtype=torch.float64)
        g = torch.tensor([g0], dtype=torch.float64)
        f_new, g_new, t, n_evals = ref_lbfgs._strong_wolfe(obj_func, x, t0, d, f0, g, g.dot(d))
        out[name + "__t0"] = np.array(t0)
        out[name + "__trials"] = np.array(trials, dtype=np.float64)
        out[name + "__result"] = np.array([float(f_new), float(t), float(n_evals)], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "linesearch.npz"), **out)
    print("linesearch.npz:", {n: len(out[n + "__trials"]) for n in out["names"]})


def gen_sphere(pymde, torch):
    """The private `_Sphere` constraint (constraints.py:203-231): tangent projection and retraction."""
    from pymde import constraints
    rng = np.random.default_rng(11)
    out = {}
    cases = [(7, 2, 1.0), (300, 3, 2.5), (200, 40, 0.5), (100, 128, 3.0)]
    out["cases"] = np.array(cases)
    for (n, d, radius) in cases:
        c = constraints._Sphere(radius)
        Z = rng.standard_normal((n, d)).astype(np.float32)
        X = c.project_onto_constraint(torch.tensor(rng.standard_normal((n, d)).astype(np.float32)), inplace=False)
        tag = "%dx%d" % (n, d)
        out["Z_" + tag] = Z
        out["X_" + tag] = X.numpy()
        out["retract_" + tag] = c.project_onto_constraint(torch.tensor(Z), inplace=False).numpy()
        out["tangent_" + tag] = c.project_onto_tangent_space(X, torch.tensor(Z), inplace=False).numpy()
    np.savez_compressed(os.path.join(HERE, "sphere.npz"), **out)
    print("sphere.npz written")


def main():
    pymde = import_reference()
    import torch
    torch.set_num_threads(1)
    only = sys.argv[1:]
    if only:  # e.g. `make_golden.py functions linesearch`
        for name in only:
            globals()["gen_" + name](pymde, torch)
        return
    gen_functions(pymde, torch)
    gen_linesearch(pymde, torch)
    gen_constraints(pymde, torch)
    gen_trajectories(pymde, torch)
    gen_trajectories_mid(pymde, torch)
    gen_trajectories_clusters(pymde, torch)
    gen_spectral(pymde, torch)
    gen_cycle(pymde, torch)
    gen_preprocess(pymde, torch)
    gen_api(pymde, torch)
    gen_sphere(pymde, torch)


if __name__ == "__main__":
    main()
