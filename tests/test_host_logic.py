"""Host-side logic that needs no GPU: the C-ABI library exports what include/mde_hip.h
declares, the line search / L-BFGS coefficient recursion, function descriptors, and the
loud failure when there is no GPU."""
import ctypes
import math
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT
from pymde_amd import _lib, lbfgs

GOLDEN = os.path.join(ROOT, "tests", "golden")


# ---------------------------------------------------------------- the C ABI
def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mde_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mde_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()  # loads (and builds if needed) without a GPU; no compute calls here
    declared = _declared_symbols()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(lib, name), "libmde_hip.so does not export %s" % name
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.SYMBOLS) == declared
    assert lib.mde_abi_version() == 2
    assert lib.mde_work_doubles(2) > 0


def test_ctypes_signatures_match_the_header_prototypes():
    """Every prototype of include/mde_hip.h against the ctypes table: same number of arguments, and for each
    one the same kind (pointer / int32 / int64 / float / double) -- an argument added on one side only would
    otherwise show up as a corrupted call, not as an error."""
    text = open(os.path.join(ROOT, "include", "mde_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = re.findall(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(mde_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", text)
    assert len(protos) >= 60

    def kind_c(t):
        t = t.strip()
        if "*" in t:
            return "ptr"
        base = re.sub(r"\b(const|unsigned|struct)\b", "", t).split()
        base = base[0] if base else ""
        return {"int64_t": "i64", "uint64_t": "u64", "int32_t": "i32", "int": "i32", "float": "f32", "double": "f64"}[base]

    def kind_py(t):
        if t in (ctypes.c_void_p, ctypes.c_char_p) or (isinstance(t, type) and issubclass(t, ctypes._Pointer)):
            return "ptr"
        return {ctypes.c_int64: "i64", ctypes.c_uint64: "u64", ctypes.c_int32: "i32", ctypes.c_float: "f32",
                ctypes.c_double: "f64"}[t]

    seen = set()
    for ret, name, params in protos:
        params = params.strip()
        plist = [] if params in ("", "void") else [q.strip() for q in params.split(",")]
        # drop the parameter name: the type is everything up to the last identifier
        kinds = [kind_c(re.sub(r"[A-Za-z_][A-Za-z0-9_]*(\[\d*\])?$", "", q) if not q.endswith("*") else q) for q in plist]
        restype, argtypes = _lib.SYMBOLS[name]
        assert [kind_py(t) for t in argtypes] == kinds, name
        rk = "ptr" if "*" in ret else {"int64_t": "i64", "int32_t": "i32", "int": "i32", "void": "void"}[
            re.sub(r"\b(const|extern)\b", "", ret).split()[-1]]
        assert (kind_py(restype) if restype is not None else "void") == rk, name
        seen.add(name)
    assert seen == set(_lib.SYMBOLS)


def test_mde_func_struct_layout_matches_header(tmp_path):
    """struct mde_func as gcc lays it out from include/mde_hip.h against the ctypes mirror (ABI 2: e0 / e1 at the end)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fields = [f for f, _ in _lib.MdeFunc._fields_]
    src = tmp_path / "f.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mde_hip.h"\nint main(void){\n'
                   'printf("%zu", sizeof(mde_func));\n'
                   + "".join('printf(" %%zu", offsetof(mde_func, %s));\n' % f for f in fields)
                   + 'return 0;}\n')
    exe = tmp_path / "f"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert got[0] == ctypes.sizeof(_lib.MdeFunc) == 80
    assert got[1:] == [getattr(_lib.MdeFunc, f).offset for f in fields]
    assert _lib.MdeFunc.a0.offset == 8 and _lib.MdeFunc.s0.offset == 32 and _lib.MdeFunc.layout.offset == 56
    assert _lib.MdeFunc.e0.offset == 64 and _lib.MdeFunc.e1.offset == 72


def test_turn_desc_layout_matches_header(tmp_path):
    """struct mde_turn_desc as gcc lays it out from include/mde_hip.h against the ctypes mirror."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.c"
    fields = [f for f, _ in _lib.MdeTurnDesc._fields_]
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mde_hip.h"\nint main(void){\n'
                   'printf("%zu", sizeof(mde_turn_desc));\n'
                   + "".join('printf(" %%zu", offsetof(mde_turn_desc, %s));\n' % f for f in fields)
                   + 'return 0;}\n')
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert got[0] == ctypes.sizeof(_lib.MdeTurnDesc)
    assert got[1:] == [getattr(_lib.MdeTurnDesc, f).offset for f in fields]


def test_ring_kernel_issues_its_lds_accesses_in_protocol_order(tmp_path):
    """The hand-shake of the LDS-ring kernel (csrc/mde_ring_kernel.h) relies on the ORDER in which a wave
    issues its LDS instructions: a consumer's operand reads in front of the store that releases the ring
    slots, a producer's chunk stores in front of the store that publishes the chunk.  The source forces
    both (volatile operand reads, a compiler barrier + a draining poll); this reads the gfx950 ISA of EVERY
    unit that instantiates the kernel -- Log1p, PushAndPull, the penalties, the losses, the run-time functor
    -- at every dimension it is built for (d = 2, 3; the run-time unit also d = 1, 4), codebook / fp32 / scalar
    parameter forms alike, and checks that the build did what the source says (round 4: at d = 3 it had not,
    and only the Log1p unit was looked at)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    units = ["mde_ring_k_log1p", "mde_ring_k_pushpull", "mde_ring_k_penalty", "mde_ring_k_penalty2", "mde_ring_k_loss",
             "mde_ring_k_loss2", "mde_ring_k_runtime"]
    procs = []
    for u in units:
        out = tmp_path / (u + ".s")
        procs.append((u, out, subprocess.Popen(
            [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"),
             "-ffp-contract=fast", "-S", "--cuda-device-only",
             os.path.join(ROOT, "pymde_amd", "csrc", u + ".hip"), "-o", str(out)],
            stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    # ring_ctrl_off(d) = (row cap + 32) * 4 d: prog[] at +0, LANDED at +64, accumulators from +256
    ctrl_of = {1: (12288 + 32) * 4, 2: (7816 + 32) * 8, 3: (5200 + 32) * 12, 4: (3936 + 32) * 16}
    words_of = {"ds_read_b32": 1, "ds_read2_b32": 2, "ds_read_b64": 2, "ds_read2_b64": 4, "ds_read_b96": 3, "ds_read_b128": 4,
                "ds_read_u8": 1, "ds_read_u16": 1}
    seen = {}
    for u, out, pr in procs:
        log, _ = pr.communicate()
        assert pr.returncode == 0, log.decode(errors="replace")[-2000:]
        lines = out.read_text().split("\n")
        starts = [i for i, l in enumerate(lines) if re.match(r"^_Z12k_fused_ringILi\dE", l)]
        ends = starts[1:] + [len(lines)]
        for s0, e0 in zip(starts, ends):
            dim = int(lines[s0][len("_Z12k_fused_ringILi")])
            ctrl = ctrl_of[dim]
            prog, pub, acc = ctrl, ctrl + 64, ctrl + 256
            releases = publishes = 0
            after_release = after_publish = False
            words = 0
            for l in lines[s0:e0]:
                t = l.split(";")[0].strip()
                if re.match(r"^\.LBB\d+_\d+:", t) or t.startswith("s_cbranch") or t.startswith("s_branch"):
                    after_release = after_publish = False   # (a basic block ends: the next one starts clean)
                    continue
                if t.startswith("ds_write_b32") and ("offset:%d" % prog) in t:
                    after_release, releases, words = True, releases + 1, 0
                elif t.startswith("ds_write_b32") and ("offset:%d" % pub) in t:
                    after_publish, publishes = True, publishes + 1
                elif t.startswith("ds_write_b128"):
                    assert not after_publish, "%s d = %d: a chunk store behind the producer's publish: %s" % (u, dim, t)
                elif after_release and t.startswith("ds_read"):
                    # behind a release the block may still read the accumulators of two entries (the pair's second
                    # and the next pair's first: 2 x d words) -- never an operand.  Operand and accumulator reads
                    # cannot be told apart by their address registers, so the WORDS read are counted (codebook
                    # and control words, whose immediate offset lies between prog[] and the accumulators, apart):
                    # the d = 3 kernel of round 4 read 10 where 6 are allowed.
                    offs = [int(v) for v in re.findall(r"offset:(\d+)", t)]
                    if offs and ctrl <= offs[0] < acc:
                        continue
                    words += words_of[t.split()[0]]
                    assert words <= 2 * dim, "%s d = %d: an operand read behind the consumer's release: %s" % (u, dim, t)
            # every kernel has the prologue's and the loop's releases; forward-only kernels too
            assert releases >= 3 and publishes >= 1, (u, dim, releases, publishes, lines[s0][:80])
            seen[(u, dim)] = seen.get((u, dim), 0) + 1
    for u in units[:6]:
        assert seen.get((u, 2), 0) >= 4 and seen.get((u, 3), 0) >= 4, (u, seen)
    for dim in (1, 2, 3, 4):
        assert seen.get(("mde_ring_k_runtime", dim), 0) >= 2, seen


# ---------------------------------------------------------------- no GPU -> loud failure
@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly():
    import pymde_amd
    edges = torch.tensor([[0, 1], [1, 2]])
    with pytest.raises(RuntimeError, match="GPU|cuda"):
        pymde_amd.MDE(3, 2, edges, pymde_amd.penalties.Quadratic(torch.ones(2)))
    with pytest.raises(RuntimeError, match="no CPU path|CPU"):
        pymde_amd.MDE(3, 2, edges, pymde_amd.penalties.Quadratic(torch.ones(2)), device="cpu")
    with pytest.raises(RuntimeError):
        pymde_amd.penalties.Quadratic(torch.ones(2))(torch.ones(2))


# ---------------------------------------------------------------- function descriptors
def test_function_descriptors():
    from pymde_amd import losses, penalties
    from pymde_amd.functions.function import KIND
    w = torch.tensor([1.0, -1.0, 2.0, 0.0])
    f = penalties.PushAndPull(w, penalties.Log1p, penalties.Log)
    s = f._hip_spec()
    assert (s.kind, s.kind_neg) == (KIND["LOG1P"], KIND["LOG"])
    assert s.scalars[0] == 1.5 and s.scalars_neg[0] == 1.0
    assert f.pos_idx.tolist() == [True, False, True, True]  # zero weight is attractive
    assert f.attractive_penalty.weights.tolist() == [1.0, 2.0, 0.0]
    assert f.repulsive_penalty.weights.tolist() == [-1.0]
    assert set(dict(f.named_buffers())) >= {"weights", "pos_idx"}
    h = penalties.Huber(torch.ones(3))
    assert h._hip_spec().scalars[0] == 0.5
    with pytest.raises(ValueError):
        penalties.Huber(torch.ones(3), threshold=-1)
    with pytest.raises(ValueError):
        penalties.InvPower(torch.ones(3))
    with pytest.raises(ValueError):
        penalties.PushAndPull(torch.ones(1))
    q = losses.WeightedQuadratic(torch.tensor([1.0, 2.0]))
    np.testing.assert_allclose(q.weights.numpy(), [1.0, 0.25])
    assert q._hip_spec().kind == KIND["L_WEIGHTED_QUADRATIC"] and q._hip_spec().a1 is q.weights
    assert losses.SoftFractional(torch.ones(2))._hip_spec().scalars[0] == 10.0
    with pytest.raises(ValueError):
        losses.SoftFractional(torch.ones(2), gamma=0.0)
    # every public class of the reference surface exists
    for name in ("Linear Quadratic Cubic Power Huber Logistic Sigmoid Hinge Log1p Log InvPower "
                 "LogRatio PushAndPull").split():
        assert hasattr(penalties, name)
    for name in ("Quadratic WeightedQuadratic Huber Cubic Power Absolute Logistic Fractional "
                 "SoftFractional").split():
        assert hasattr(losses, name)


# ---------------------------------------------------------------- line search
def test_cubic_interpolate():
    # minimiser of f(x) = (x - 1)^2 from the points 0 and 3
    assert lbfgs.cubic_interpolate(0.0, 1.0, -2.0, 3.0, 4.0, 4.0) == pytest.approx(1.0)
    # negative discriminant -> bisection
    assert lbfgs.cubic_interpolate(0.0, 0.0, 4.0, 1.0, 3.0, 4.0) == pytest.approx(0.5)
    # clipped to the bounds
    assert lbfgs.cubic_interpolate(0.0, 1.0, -2.0, 3.0, 4.0, 4.0, bounds=(1.5, 2.0)) == 1.5


def _wolfe_ok(phi, t, f0, g0, c1=1e-4, c2=0.9):
    f, g, _ = phi(t)
    return f <= f0 + c1 * t * g0 and abs(g) <= -c2 * g0


@pytest.mark.parametrize("case", ["quadratic", "quartic", "steep", "far"])
def test_strong_wolfe_satisfies_the_conditions(case):
    fns = {
        "quadratic": (lambda t: ((t - 2.0) ** 2, 2 * (t - 2.0)), 1.0),
        "quartic": (lambda t: ((t - 0.3) ** 4 + 0.1 * t, 4 * (t - 0.3) ** 3 + 0.1), 1.0),
        "steep": (lambda t: (math.exp(5 * t) - 20 * t, 5 * math.exp(5 * t) - 20), 1.0),
        "far": (lambda t: (0.5 * (t - 50.0) ** 2, t - 50.0), 0.01),
    }
    fn, t0 = fns[case]
    calls = []

    def phi(t):
        f, g = fn(t)
        calls.append(t)
        return f, g, True

    f0, g0 = fn(0.0)
    f_t, t, n = lbfgs.strong_wolfe(phi, t0, f0, g0, d_norm=1.0)
    assert _wolfe_ok(phi, t, f0, g0)
    assert f_t == pytest.approx(fn(t)[0])
    assert n <= 26


def test_strong_wolfe_backs_off_non_finite_trials():
    def phi(t):
        if t > 0.3:
            return float("nan"), float("nan"), False
        return (t - 0.2) ** 2, 2 * (t - 0.2), True
    f_t, t, _ = lbfgs.strong_wolfe(phi, 1.0, 0.04, -0.4, d_norm=1.0)
    assert 0 < t <= 0.3 and f_t < 0.04

    def always_nan(t):
        return float("nan"), 0.0, True
    with pytest.raises(lbfgs.LineSearchError):
        lbfgs.strong_wolfe(always_nan, 1.0, 1.0, -1.0, d_norm=1.0)


def test_strong_wolfe_reproduces_the_reference_trial_sequences():
    """tests/golden/linesearch.npz: every trial step and the returned (f, t, n_evals) of the
    reference's _strong_wolfe (lbfgs.py:44-253), recorded by running it in the build container on
    scalar problems that hit the extrapolation phase, the zoom with its 10 % safeguard, the
    NaN-halving prologue, and the 0.8 back-off (once recovering, once ending at t = 0)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    g = np.load(os.path.join(GOLDEN, "linesearch.npz"))
    assert len(g["names"]) >= 6
    for name, fn, t0 in mg.linesearch_problems():
        want_trials, want = g[name + "__trials"], g[name + "__result"]
        seen = []

        def phi(t):
            seen.append(t)
            f, gd = fn(t)
            return f, gd, not (math.isnan(gd) or math.isinf(gd))
        f0, g0 = fn(0.0)
        f_t, t, n = lbfgs.strong_wolfe(phi, float(t0), f0, g0, d_norm=1.0)
        assert len(seen) == len(want_trials), (name, seen, want_trials)
        np.testing.assert_allclose(seen, want_trials, rtol=1e-6, atol=1e-12, err_msg=name)
        np.testing.assert_allclose([f_t, t], want[:2], rtol=1e-6, atol=1e-12, err_msg=name)
        assert n == int(want[2]), name


# ---------------------------------------------------------------- two-loop recursion
def _two_loop(g, S, Y, H):
    """Explicit recursion of lbfgs.py:490-507 on numpy vectors."""
    q = -g.copy()
    m = len(S)
    ro = [1.0 / Y[i].dot(S[i]) for i in range(m)]
    al = [0.0] * m
    for i in range(m - 1, -1, -1):
        al[i] = S[i].dot(q) * ro[i]
        q -= al[i] * Y[i]
    r = q * H
    for i in range(m):
        be = Y[i].dot(r) * ro[i]
        r += (al[i] - be) * S[i]
    return r


def test_lbfgs_memory_matches_explicit_two_loop():
    rng = np.random.default_rng(0)
    N, hist = 50, 4
    mem = lbfgs.LbfgsMemory(hist)
    A = rng.standard_normal((N, N))
    A = A @ A.T + N * np.eye(N)  # SPD quadratic -> y.s > 0
    S, Y = [], []
    x = rng.standard_normal(N)
    g_prev = A @ x
    for step in range(9):
        d = -rng.standard_normal(N) * 0.1 - 0.05 * g_prev
        t = 0.7
        x = x + t * d
        g = A @ x
        y, s = g - g_prev, t * d
        # the dots the device would return for (y*, s*) against the stored pairs
        dots = [y.dot(s), y.dot(y), s.dot(g), y.dot(g)]
        for sj, yj in zip(S, Y):
            dots += [sj.dot(y), yj.dot(y), s.dot(yj), sj.dot(g), yj.dot(g)]
        accepted, Sg, Yg = mem.absorb(np.array(dots))
        assert accepted
        S.append(s)
        Y.append(y)
        if len(S) > hist:
            S.pop(0)
            Y.pop(0)
        assert mem.count == len(S)
        c_g, cs, cy = mem.direction_coefficients(Sg, Yg)
        got = c_g * g + sum(c * v for c, v in zip(cs, S)) + sum(c * v for c, v in zip(cy, Y))
        want = _two_loop(g, S, Y, y.dot(s) / y.dot(y))
        np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-12)
        g_prev = g
    # a pair with y.s <= 1e-10 is rejected and leaves the memory untouched (lbfgs.py:472)
    before = mem.SY.copy()
    dots = [1e-12, 1.0, 0.0, 0.0] + [0.0] * (5 * mem.count)
    accepted, Sg, Yg = mem.absorb(np.array(dots))
    assert not accepted and mem.count == hist and np.array_equal(before, mem.SY)
    assert len(Sg) == hist
