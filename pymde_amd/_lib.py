"""ctypes binding of libmde_hip.so -- the only compute path of this package.

There is deliberately no CPU or PyTorch fallback: if the HIP library cannot be loaded, or
a call is made without a GPU, the error is raised to the caller.  ``include/mde_hip.h`` is
the authoritative declaration of every symbol bound here.
"""
import ctypes
import os
import threading

import torch  # noqa: F401  (must be imported first: libmde_hip.so binds to torch's libamdhip64)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmde_hip.so")

c_i32, c_i64, c_f32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float
c_vp = ctypes.c_void_p

MDE_OK = 0
MDE_E_INVALID, MDE_E_SELF_EDGE, MDE_E_RANGE = -1, -2, -3
MDE_E_TOO_LARGE, MDE_E_HIP, MDE_E_UNSUPPORTED = -4, -5, -6


class MdeFunc(ctypes.Structure):
    """Mirror of ``struct mde_func`` (include/mde_hip.h)."""
    _fields_ = [("kind", c_i32), ("kind_neg", c_i32), ("a0", c_vp), ("a1", c_vp),
                ("a0_scalar", c_i32), ("a1_scalar", c_i32),
                ("s0", c_f32), ("s1", c_f32), ("s2", c_f32),
                ("n0", c_f32), ("n1", c_f32), ("n2", c_f32), ("layout", c_i32),
                ("e0", c_vp), ("e1", c_vp)]


class MdeTurnDesc(ctypes.Structure):
    """Mirror of ``struct mde_turn_desc`` (include/mde_hip.h)."""
    _fields_ = [("plan", c_vp), ("func", c_vp), ("n", c_i64), ("d", c_i32), ("kind", c_i32),
                ("X", c_vp * 2), ("g", c_vp), ("g_prev", c_vp), ("dir", c_vp), ("loss_dev", c_vp),
                ("board", c_vp), ("work", c_vp), ("status", c_vp), ("lbfgs", c_vp),
                ("host_dst", c_vp), ("tail_src", c_vp), ("read_bytes", c_i64),
                ("host_loss", c_vp), ("host_status", c_vp), ("host_board", c_vp), ("seq", ctypes.c_double),
                ("pre_id", ctypes.c_double)]


# every exported symbol of include/mde_hip.h: name -> (restype, argtypes)
SYMBOLS = {
    "mde_last_error": (ctypes.c_char_p, []),
    "mde_abi_version": (c_i32, []),
    "mde_plan_create": (c_i32, [c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, ctypes.POINTER(c_vp)]),
    "mde_plan_destroy": (c_i32, [c_vp]),
    "mde_plan_n": (c_i64, [c_vp]),
    "mde_plan_p": (c_i64, [c_vp]),
    "mde_plan_half_edges": (c_i64, [c_vp]),
    "mde_plan_row_lo": (c_i64, [c_vp]),
    "mde_plan_row_hi": (c_i64, [c_vp]),
    "mde_plan_rowptr": (c_vp, [c_vp]),
    "mde_plan_nbr": (c_vp, [c_vp]),
    "mde_plan_eid": (c_vp, [c_vp]),
    "mde_plan_export": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mde_shard_bounds": (c_i32, [c_i64, c_i64, c_vp, c_i32, ctypes.POINTER(c_i64), c_vp]),
    "mde_plan_expand": (c_i32, [c_vp, c_vp, c_vp, c_vp]),
    "mde_plan_layout": (c_i32, [c_vp, c_i32, c_vp]),
    "mde_plan_layout_half_edges": (c_i64, [c_vp, c_i32]),
    "mde_plan_ring_info": (c_i32, [c_vp, ctypes.POINTER(c_i64)]),
    "mde_plan_row_order": (c_i32, [c_vp, c_i32, c_vp, ctypes.POINTER(ctypes.c_double)]),
    "mde_plan_function_hint": (c_i32, [c_vp, c_i32, c_i32]),
    "mde_plan_loss_double": (c_i32, [c_vp, c_vp, c_vp]),
    "mde_center_step_begin": (c_i32, [c_i64, c_i32, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp]),
    "mde_center_step_end": (c_i32, [c_i64, c_i32, c_vp, c_vp, ctypes.c_double, c_vp]),
    "mde_lbfgs_dev_stage": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp]),
    "mde_lbfgs_dev_finish": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mde_lbfgs_dev_dots": (c_i32, [c_vp]),
    "mde_rank_reduce": (c_i32, [c_i32, c_i32, ctypes.c_uint64, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "mde_plan_expand_codebook": (c_i32, [c_vp, c_vp, c_vp, ctypes.POINTER(c_i32), c_vp]),
    "mde_plan_expand_bytes": (c_i32, [c_vp, c_vp, c_vp, ctypes.POINTER(c_i32), c_vp]),
    "mde_plan_expand_layout": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp]),
    "mde_edges_deduplicate": (c_i32, [c_i64, c_i64, c_vp, c_vp, ctypes.POINTER(c_i64), c_vp]),
    "mde_edges_count_unique": (c_i32, [c_i64, c_i64, c_vp, c_vp, c_vp, ctypes.POINTER(c_i64), c_vp]),
    "mde_knn": (c_i32, [c_i64, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "mde_knn_pairs": (c_i32, [c_i64, c_i32, c_vp, c_vp, c_f32, c_vp, c_vp]),
    "mde_graph_knn": (c_i32, [c_vp, c_vp, c_f32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "mde_graph_shortest_paths": (c_i32, [c_vp, c_vp, c_f32, ctypes.c_double, ctypes.c_uint64, c_i64,
                                         c_vp, c_vp, ctypes.POINTER(c_i64), c_vp]),
    "mde_sample_edges": (c_i32, [c_i64, c_i64, ctypes.c_uint64, c_vp, c_i64, c_vp,
                                 ctypes.POINTER(c_i64), c_vp]),
    "mde_average_distortion": (c_i32, [c_vp, c_vp, c_i32, ctypes.POINTER(MdeFunc), c_f32, c_vp,
                                       c_vp, c_vp]),
    "mde_differences": (c_i32, [c_i64, c_i64, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "mde_distances": (c_i32, [c_i64, c_i64, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "mde_distances_backward": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "mde_distortions": (c_i32, [c_i64, c_vp, ctypes.POINTER(MdeFunc), c_vp, c_vp, c_vp]),
    "mde_scatter": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp, c_f32, c_vp, c_vp]),
    "mde_center": (c_i32, [c_i64, c_i32, c_vp, c_vp, c_vp]),
    "mde_anchor_rows": (c_i32, [c_i64, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "mde_sphere_rows": (c_i32, [c_i64, c_i32, c_vp, c_vp, c_f32, c_vp]),
    "mde_std_tangent": (c_i32, [c_i64, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "mde_std_retract": (c_i32, [c_i64, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "mde_center_step": (c_i32, [c_i64, c_i32, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp]),
    "mde_std_retract_step": (c_i32, [c_i64, c_i32, c_vp, c_vp, c_f32, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "mde_std_tangent_stats": (c_i32, [c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mde_gram": (c_i32, [c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mde_right_multiply": (c_i32, [c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "mde_right_multiply_add": (c_i32, [c_i64, c_i32, c_i32, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp]),
    "mde_shift_rows": (c_i32, [c_i64, c_i32, c_vp, c_vp, c_vp]),
    "mde_row_scale": (c_i32, [c_i64, c_i32, c_vp, c_vp, c_vp]),
    "mde_weighted_degree": (c_i32, [c_vp, c_vp, c_vp, c_vp]),
    "mde_work_doubles": (c_i64, [c_i32]),
    "mde_axpy": (c_i32, [c_i64, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "mde_vec_stats": (c_i32, [c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mde_lbfgs_create": (c_i32, [c_i64, c_i32, ctypes.POINTER(c_vp)]),
    "mde_lbfgs_destroy": (c_i32, [c_vp]),
    "mde_lbfgs_reset": (c_i32, [c_vp]),
    "mde_lbfgs_count": (c_i32, [c_vp]),
    "mde_lbfgs_stage": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp]),
    "mde_lbfgs_commit": (c_i32, [c_vp, c_i32]),
    "mde_lbfgs_dev_reset": (c_i32, [c_vp, c_vp]),
    "mde_lbfgs_dev_step": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "mde_lbfgs_debug_knobs": (c_i32, [c_i32, c_i32, c_i32, c_i32]),
    "mde_lbfgs_dev_info": (c_i32, [c_vp, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32), c_vp]),
    "mde_lbfgs_combine": (c_i32, [c_vp, c_vp, c_f32, ctypes.POINTER(c_f32),
                                  ctypes.POINTER(c_f32), c_vp, c_vp, c_vp, c_vp]),
    "mde_copy_to_host": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "mde_turn_enqueue": (c_i32, [c_vp, c_i32, c_f32, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_i32, c_vp]),
    "mde_turn_wait": (c_i32, [c_vp, c_i32, ctypes.c_double, c_i32, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                             c_vp, c_vp]),
}

_lib = None
_lock = threading.Lock()


class MdeHipError(RuntimeError):
    """A call into libmde_hip.so failed."""

    def __init__(self, code, message):
        super().__init__("libmde_hip error %d: %s" % (code, message))
        self.code = code
        self.message = message


def load():
    """Load (building if a toolchain is present and sources are newer) and bind the library."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        from pymde_amd import _build
        if (not os.path.exists(LIB_PATH) or os.environ.get("PYMDE_AMD_REBUILD")
                or (_build._stale() and _build._hipcc() is not None)):
            _build.build(verbose=bool(os.environ.get("PYMDE_AMD_VERBOSE")))
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "pymde_amd: %s is missing and could not be built; the HIP extension is the "
                "only compute path (no CPU fallback)" % LIB_PATH)
        # (design experiments on the GPU box: another build of the same library, tools/build_variant.sh)
        lib = ctypes.CDLL(os.environ.get("PYMDE_AMD_LIB_VARIANT") or LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if lib.mde_abi_version() != 2:
            raise ImportError("pymde_amd: ABI version mismatch in %s" % LIB_PATH)
        _lib = lib
    return _lib


def last_error():
    msg = load().mde_last_error()
    return msg.decode(errors="replace") if msg else ""


def check(rc):
    if rc != MDE_OK:
        raise MdeHipError(rc, last_error())


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError(
            "pymde_amd needs an AMD GPU (MI355X / gfx950): torch.cuda.is_available() is False "
            "and there is no CPU path")


def ptr(t):
    """Device pointer of a tensor (or None)."""
    return None if t is None else c_vp(t.data_ptr())


def stream_ptr(device=None):
    return c_vp(torch.cuda.current_stream(device).cuda_stream)
