"""ctypes loader for libhipfact.so (the C ABI declared in include/hipfact.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C
sleqp_amd/csrc``.  There is no fallback: if the shared object is missing the
import of any numeric entry point fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HIPFACT_LIBRARY: an alternative build of the same library (instrumented copies, scripts/timeline.py)
LIB_PATH = os.environ.get("HIPFACT_LIBRARY") or os.path.join(_HERE, "csrc", "libhipfact.so")

HIPFACT_OK = 0
ERRORS = {
    -1: "HIPFACT_EINVAL",
    -2: "HIPFACT_EDEVICE",
    -3: "HIPFACT_ESINGULAR",
    -4: "HIPFACT_ENOMEM",
    -5: "HIPFACT_ESTATE",
    -6: "HIPFACT_EINTERNAL",
}

# every symbol include/hipfact.h declares
SYMBOLS = [
    "hipfact_create", "hipfact_free", "hipfact_retain", "hipfact_last_error", "hipfact_last_warning", "hipfact_set_matrix", "hipfact_solve_sparse",
    "hipfact_solve_dense", "hipfact_solution", "hipfact_solution_view", "hipfact_condition", "hipfact_refactor_device",
    "hipfact_solve_device", "hipfact_solution_device", "hipfact_synchronize", "hipfact_check", "hipfact_stream",
    "hipfact_assemble_kkt", "hipfact_reduced_matrix", "hipfact_spmat_create", "hipfact_spmat_update_values", "hipfact_spmat_free",
    "hipfact_spmat_mult_vec", "hipfact_spmat_mult_vec_trans", "hipfact_spmat_mult_vec_sym",
    "hipfact_spmat_mult_device", "hipfact_steihaug_solve", "hipfact_tr_solve", "hipfact_tr_solve_ex", "hipfact_tridiag_tr", "hipfact_set_option", "hipfact_get_info", "hipfact_debug_copy", "hipfact_debug_pool_selftest", "hipfact_plan_create",
    "hipfact_plan_free", "hipfact_plan_error", "hipfact_plan_array", "hipfact_plan_scalar",
]

_lib = None


class HipfactError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{ERRORS.get(code, code)}: {message}")
        self.code = code


def load() -> C.CDLL:
    """Loads libhipfact.so; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C sleqp_amd/csrc)")
    lib = C.CDLL(LIB_PATH)
    vp, ci, cd = C.c_void_p, C.c_int, C.c_double
    lib.hipfact_create.argtypes = [C.POINTER(vp), ci]
    lib.hipfact_free.argtypes = [C.POINTER(vp)]
    lib.hipfact_retain.argtypes = [vp]
    lib.hipfact_last_error.argtypes = [vp]
    lib.hipfact_last_error.restype = C.c_char_p
    lib.hipfact_last_warning.argtypes = [vp]
    lib.hipfact_last_warning.restype = C.c_char_p
    lib.hipfact_set_matrix.argtypes = [vp, ci, vp, vp, vp]
    lib.hipfact_solve_sparse.argtypes = [vp, ci, ci, vp, vp]
    lib.hipfact_solve_dense.argtypes = [vp, vp]
    lib.hipfact_solution.argtypes = [vp, vp, ci, ci]
    lib.hipfact_solution_view.argtypes = [vp, C.POINTER(vp), ci, ci]
    lib.hipfact_condition.argtypes = [vp, C.POINTER(cd)]
    lib.hipfact_refactor_device.argtypes = [vp, vp]
    lib.hipfact_solve_device.argtypes = [vp, vp, vp]
    lib.hipfact_solution_device.argtypes = [vp, C.POINTER(vp)]
    lib.hipfact_synchronize.argtypes = [vp]
    lib.hipfact_check.argtypes = [vp]
    lib.hipfact_stream.argtypes = [vp, C.POINTER(vp)]
    lib.hipfact_assemble_kkt.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, ci, C.POINTER(ci), vp, vp, vp]
    lib.hipfact_reduced_matrix.argtypes = [vp, C.POINTER(ci), vp, vp, vp]
    lib.hipfact_spmat_create.argtypes = [vp, ci, ci, vp, vp, vp, C.POINTER(vp)]
    lib.hipfact_spmat_update_values.argtypes = [vp, vp]
    lib.hipfact_spmat_free.argtypes = [C.POINTER(vp)]
    lib.hipfact_spmat_mult_vec.argtypes = [vp, vp, vp]
    lib.hipfact_spmat_mult_vec_trans.argtypes = [vp, vp, vp]
    lib.hipfact_spmat_mult_vec_sym.argtypes = [vp, vp, vp]
    lib.hipfact_spmat_mult_device.argtypes = [vp, ci, vp, vp]
    lib.hipfact_steihaug_solve.argtypes = [vp, vp, vp, cd, cd, ci, vp, C.POINTER(cd), C.POINTER(ci)]
    lib.hipfact_tr_solve.argtypes = [vp, ci, vp, vp, vp, vp, cd, cd, ci, vp, C.POINTER(cd), C.POINTER(ci)]
    lib.hipfact_tr_solve_ex.argtypes = [vp, ci, vp, vp, vp, vp, cd, cd, ci, vp, C.POINTER(cd), C.POINTER(ci), vp]
    lib.hipfact_tridiag_tr.argtypes = [ci, vp, vp, cd, cd, vp, C.POINTER(cd)]
    lib.hipfact_set_option.argtypes = [vp, C.c_char_p, cd]
    lib.hipfact_debug_copy.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
    lib.hipfact_debug_pool_selftest.argtypes = [ci, ci]
    lib.hipfact_get_info.argtypes = [vp, C.c_char_p, C.POINTER(cd)]
    lib.hipfact_plan_create.argtypes = [ci, vp, vp, vp, C.POINTER(vp)]
    lib.hipfact_plan_free.argtypes = [C.POINTER(vp)]
    lib.hipfact_plan_error.argtypes = [vp]
    lib.hipfact_plan_error.restype = C.c_char_p
    lib.hipfact_plan_array.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(ci)]
    lib.hipfact_plan_scalar.argtypes = [vp, C.c_char_p, C.POINTER(cd)]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int or fn.restype is None:
            pass
    lib.hipfact_plan_free.restype = None
    _lib = lib
    return lib


def kernel_sources_sha16() -> str:
    """Hash of everything the device translation unit is compiled from (csrc/*.hip, *.inc, *.h): recorded with the
    rocprofv3 --pmc summaries under profiles/, so that bench.py only quotes HBM traffic measured on the sources it runs."""
    import glob
    import hashlib

    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha256()
    for name in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.inc")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(name).encode())
        h.update(open(name, "rb").read())
    return h.hexdigest()[:16]
