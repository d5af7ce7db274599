import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hipfact_lib():
    """libhipfact.so via ctypes; builds it when missing (hipcc cross-compiles without a GPU)."""
    from sleqp_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import subprocess

        subprocess.check_call(["make", "-C", os.path.join(ROOT, "sleqp_amd", "csrc")])
    return _lib.load()


def has_gpu() -> bool:
    import ctypes as C

    from sleqp_amd import _lib

    try:
        lib = _lib.load()
    except Exception:
        return False
    h = C.c_void_p()
    rc = lib.hipfact_create(C.byref(h), 0)
    if rc == 0:
        lib.hipfact_free(C.byref(h))
        return True
    return False
