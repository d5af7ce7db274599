"""The C-ABI library loads, exports every symbol include/hipfact.h declares, and
fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, has_gpu


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "hipfact.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hipfact_[a-z_]+)\s*\(", text)))


def test_header_symbols_exported(hipfact_lib):
    from sleqp_amd import _lib

    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(hipfact_lib, name), f"{name} declared in include/hipfact.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared


def test_no_cpu_fallback_without_gpu(hipfact_lib):
    if has_gpu():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = hipfact_lib.hipfact_create(C.byref(h), 0)
    assert rc == -2 and not h  # HIPFACT_EDEVICE
    assert b"no CPU fallback" in hipfact_lib.hipfact_last_error(None)
    from sleqp_amd.fact import HipFact
    from sleqp_amd import HipfactError

    with pytest.raises(HipfactError):
        HipFact()


def test_null_handle_calls_are_rejected(hipfact_lib):
    assert hipfact_lib.hipfact_set_matrix(None, 0, None, None, None) == -1
    assert hipfact_lib.hipfact_solve_dense(None, None) == -1
    assert hipfact_lib.hipfact_solution(None, None, 0, 0) == -1
    p = C.c_void_p()
    assert hipfact_lib.hipfact_free(C.byref(p)) == 0
