"""mckpp_f90_amd - MI355X-native column physics for MC-KPP.

The product is the gfx950 shared library ``libmckpp_hip.so`` (HIP kernels +
C-ABI, see ``include/mckpp_hip.h``) and the Fortran host layer under
``fortran/``.  This Python package is thin plumbing over the same C-ABI for
tests, benchmarks and multi-GPU launch; it holds no physics and has no CPU
fallback - loading fails loudly if the library is missing.
"""
import ctypes as _C
import os as _os

_HERE = _os.path.dirname(_os.path.abspath(__file__))
# MCKPP_HIP_LIBRARY names another build of the same library (a profiling or A/B build made by tools/); the
# default is the in-tree product library
LIB_PATH = _os.environ.get("MCKPP_HIP_LIBRARY") or _os.path.join(_HERE, "libmckpp_hip.so")
INCLUDE_DIR = _os.path.join(_os.path.dirname(_HERE), "include")

_lib = None


def load_library():
    """dlopen the in-tree gfx950 library (build it with ``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        if not _os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C mckpp_f90_amd/csrc`). "
                "There is no CPU fallback."
            )
        _lib = _C.CDLL(LIB_PATH)
    return _lib


from .api import (  # noqa: E402,F401
    KppConstFields,
    Kpp3dFields,
    MckppHip,
    MckppHipMulti,
    MckppHipError,
    host_shard_mask,
    mckpp_initialize_ocean_model,
    mckpp_physics_driver,
    mckpp_physics_lookup,
)
