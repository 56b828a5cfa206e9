"""ctypes binding of libfeasthip.so -- one prototype per symbol of include/feasthip.h."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class FeastHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"feasthip error {code}: {msg}")
        self.code = code


class Stats(C.Structure):
    _fields_ = [
        ("n", C.c_int64), ("nsuper", C.c_int64), ("nnz_l", C.c_int64), ("nnz_u", C.c_int64),
        ("fact_elems", C.c_int64), ("max_front", C.c_int64), ("max_p", C.c_int64), ("depth", C.c_int64),
        ("flops_factor", C.c_double), ("analyze_seconds", C.c_double), ("factor_ms", C.c_double),
        ("solve_ms", C.c_double), ("rr_ms", C.c_double), ("solve_bytes", C.c_double), ("solve_flops", C.c_double),
        ("perturbed_pivots", C.c_int64), ("local_nodes", C.c_int32), ("m0", C.c_int32), ("ldw", C.c_int32),
        ("dense", C.c_int32), ("alloc_seconds", C.c_double),
        ("factor_units", C.c_int32), ("folded", C.c_int32), ("solve_bytes_streamed", C.c_double), ("symmetric", C.c_int32), ("rr_device_sweeps", C.c_int32), ("growth", C.c_double), ("proj_ms", C.c_double),
        ("early_sweep_levels", C.c_int32), ("reserved0", C.c_int32),
    ]


VP, I, I64, D = C.c_void_p, C.c_int, C.c_int64, C.c_double
PVP = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/feasthip.h one to one
PROTOTYPES = {
    "feasthip_version": (I, []),
    "feasthip_device_count": (I, []),
    "feasthip_create": (I, [PVP, I]),
    "feasthip_destroy": (I, [VP]),
    "feasthip_trim_memory": (I, []),
    "feasthip_last_error": (C.c_char_p, [VP]),
    "feasthip_set_stream": (I, [VP, VP]),
    "feasthip_set_dense": (I, [VP, I64, I, PVP, C.POINTER(I), C.POINTER(D)]),
    "feasthip_set_csc": (I, [VP, I64, I, PVP, PVP, PVP, C.POINTER(I), C.POINTER(D), I]),
    "feasthip_set_contour": (I, [VP, I, VP, VP, I, VP]),
    "feasthip_set_node_shard": (I, [VP, I, I]),
    "feasthip_analyze": (I, [VP]),
    "feasthip_factorize": (I, [VP]),
    "feasthip_set_x": (I, [VP, VP, I64, I]),
    "feasthip_get_x": (I, [VP, VP]),
    "feasthip_get_block": (I, [VP, I, VP]),
    "feasthip_comm_unique_id": (I, [VP]),
    "feasthip_comm_init": (I, [VP, VP, I, I]),
    "feasthip_comm_destroy": (I, [VP]),
    "feasthip_rr_step": (I, [VP, VP, VP, VP]),
    "feasthip_small_eig_herm": (I, [VP, I, VP, VP, VP, VP, VP]),
    "feasthip_set_rr_only": (I, [VP, I]),
    "feasthip_contour_step": (I, [VP, I]),
    "feasthip_contour_step_begin": (I, [VP, I, I]),
    "feasthip_moment_chunk_wait": (I, [VP, I, C.POINTER(I64), C.POINTER(I64)]),
    "feasthip_contour_step_end": (I, [VP]),
    "feasthip_nl_contour_step": (I, [VP, I]),
    "feasthip_nl_rr_step": (I, [VP, VP, VP, VP]),
    "feasthip_nl_orthonormalize_x": (I, [VP]),
    "feasthip_nl_finalize_x": (I, [VP]),
    "feasthip_q_device": (I, [VP, PVP, C.POINTER(I64), C.POINTER(I64), C.POINTER(I)]),
    "feasthip_set_q_buffer": (I, [VP, VP, C.c_size_t]),
    "feasthip_feast_run": (I, [VP, VP, I64, I, I, D, I, VP, VP, VP, C.POINTER(I)]),
    "feasthip_dual_feast_run": (I, [VP, VP, VP, I64, I, I, D, I, VP, VP, VP, VP, C.POINTER(I)]),
    "feasthip_nlfeast_run": (I, [VP, VP, I64, I, I, D, D, VP, VP, VP, C.POINTER(I)]),
    "feasthip_set_precision": (I, [VP, I]),
    "feasthip_set_conjugate_fold": (I, [VP, I]),
    "feasthip_set_lazy_conversion": (I, [VP, I]),
    "feasthip_shard_plan": (I, [I, VP, I, I, I, VP, VP, VP, VP]),
    "feasthip_set_coef_functions": (I, [VP, VP, VP]),
    "feasthip_set_node_coefficients": (I, [VP, VP, I, I]),
    "feasthip_nl_beyn_step": (I, [VP, VP, VP]),
    "feasthip_nl_set_r": (I, [VP, VP]),
    "feasthip_nlm_begin": (I, [VP, VP, I64, I, I]),
    "feasthip_nlm_contour_step": (I, [VP, I]),
    "feasthip_nlm_candidates": (I, [VP, VP, VP]),
    "feasthip_nlm_finish": (I, [VP, VP, VP, VP, VP, VP]),
    "feasthip_nlm_get_y": (I, [VP, VP]),
    "feasthip_nlfeast_moments_run": (I, [VP, VP, I64, I, I, I, D, D, VP, VP, VP, VP, C.POINTER(I)]),
    "feasthip_estimate_count": (I, [VP, VP, I64, I, C.POINTER(D)]),
    "feasthip_lu_dense": (I, [I, VP, I64, PVP]),
    "feasthip_lu_dense_c64": (I, [I, VP, I64, PVP]),
    "feasthip_lu_csc": (I, [I, I64, VP, VP, VP, I, I, PVP]),
    "feasthip_ldiv": (I, [VP, VP, I64, VP, I64, I]),
    "feasthip_ldiv_adjoint": (I, [VP, VP, I64, VP, I64, I]),
    "feasthip_fact_free": (I, [VP]),
    "feasthip_fact_info": (I, [VP, C.POINTER(I), C.POINTER(D), C.POINTER(I64)]),
    "feasthip_get_stats": (I, [VP, C.POINTER(Stats)]),
    "feasthip_profile_solve": (I, [VP, VP]),
    "feasthip_profile_detail": (I, [VP, VP, I]),
    "feasthip_level_table": (I, [VP, VP, I]),
    "feasthip_mm_last_error": (C.c_char_p, []),
    "feasthip_mm_info": (I, [C.c_char_p, C.POINTER(I64), C.POINTER(I64), C.POINTER(I64), C.POINTER(I)]),
    "feasthip_mm_read_csc": (I, [C.c_char_p, VP, VP, VP, I64]),
    "feasthip_mm_write_csc": (I, [C.c_char_p, I64, I64, VP, VP, VP, I, I]),
    "feasthip_symbolic_probe": (I, [I64, VP, VP, I, I, VP, VP, VP, VP]),
    "feasthip_symbolic_contains": (I, [I64, VP, VP, I, I, VP, VP, I64]),
    "feasthip_small_eig": (I, [I, VP, VP, VP, VP]),
}


def library_path():
    return os.path.join(_HERE, "libfeasthip.so")


# void (*feasthip_coef_fn)(const double* z, int count, double* f, void* user)
COEF_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_void_p)


def load_library():
    """Load libfeasthip.so (built in-tree by __graft_entry__.build()).  Raises if it is
    missing: there is no Python/CPU fallback for the hot path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch (if it is going to be used in this process: bench.py, distributed.py) bundles its own
    # libamdhip64.so with the same SONAME as /opt/rocm's.  Whichever is loaded first wins, and torch does not
    # find its GPUs on top of a foreign runtime -- so let torch load first when it is installed.
    if os.environ.get("FEASTHIP_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    path = library_path()
    if not os.path.exists(path):
        raise FeastHipError(-6, f"{path} not found: build it with `make -C feastsolver.jl_amd/csrc` "
                                "(python -c 'import __graft_entry__ as g; g.build()')")
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)     # AttributeError if the library does not export the symbol
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(lib, handle, code, allow_positive=True):
    if code < 0 or (code > 0 and not allow_positive):
        msg = lib.feasthip_last_error(handle)
        raise FeastHipError(code, msg.decode() if msg else "")
    return code
