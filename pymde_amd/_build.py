"""Build libmde_hip.so (gfx950) in-tree with hipcc.

The shared library is the product's only compute path; it is rebuilt when any source under
``csrc/`` or ``include/`` is newer than the library.  There is no fallback: if hipcc is
missing and no prebuilt library exists, importing the kernels fails loudly.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libmde_hip.so")
SOURCES = ["mde_plan.hip", "mde_distortion.hip", "mde_ring.hip", "mde_ring_k_log1p.hip", "mde_ring_k_pushpull.hip",
           "mde_ring_k_penalty.hip", "mde_ring_k_penalty2.hip", "mde_ring_k_loss.hip", "mde_ring_k_loss2.hip", "mde_ring_k_runtime.hip", "mde_vec.hip", "mde_mfma.hip", "mde_edges.hip", "mde_knn.hip", "mde_graph.hip"]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link libmde_hip.so.  Returns the path."""
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    if hipcc is None:
        if os.path.exists(LIB):
            return LIB  # GPU box without a toolchain: use the library that travelled
        raise RuntimeError("hipcc not found and %s does not exist" % LIB)
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE,
             "-Wno-unused-result", "-ffp-contract=fast"] + os.environ.get("MDE_EXTRA_FLAGS", "").split()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        srcp = os.path.join(CSRC, src)
        hdr_t = max(os.path.getmtime(os.path.join(d, f)) for d in (CSRC, INCLUDE)
                    for f in os.listdir(d) if f.endswith(".h"))
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(srcp)
                and os.path.getmtime(obj) > hdr_t):
            continue
        cmd = [hipcc] + flags + ["-c", srcp, "-o", obj]
        if verbose:
            print("[pymde_amd build]", " ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError("hipcc failed on %s" % src)
        elif verbose and out.strip():
            print(out.decode(errors="replace"))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print("[pymde_amd build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
