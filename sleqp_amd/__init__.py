"""sleqp_amd — MI355X-native KKT linear-algebra backend for SLEQP (hipfact).

Python-side plumbing over the C ABI in include/hipfact.h: ctypes bindings, the
mirror of the reference's SleqpFact / SleqpAugJac interfaces used by the parity
tests and the bench, and the seeded synthetic-problem generators.
"""
from ._lib import HipfactError, LIB_PATH, SYMBOLS, load  # noqa: F401
from .sparse import SleqpMat, SleqpVec  # noqa: F401


def __getattr__(name):
    if name in ("HipFact", "SpMat", "StandardAugJac"):
        from . import fact

        return getattr(fact, name)
    raise AttributeError(name)
