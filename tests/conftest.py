import os
import sys

import pytest
# torch before anything loads libmckpp_hip.so: torch brings its own HIP runtime, the library links the image's; whichever
# is loaded first serves both, and torch does not see the device through the other one (a run of a subset of the test
# files - without tests/test_dist_cpu.py, which imports torch at collection - otherwise fails in the first fixture that
# asks torch.cuda.is_available() after a test has used the library)
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


@pytest.fixture(autouse=True)
def _collect_contexts_between_tests():
    """Device contexts that a test leaves to the garbage collector (mckpp_initialize_ocean_model keeps one on its
    kpp_const_fields: a reference cycle) are finalized - their host arrays un-pinned, their device memory freed - when
    the test ends, not at some allocation in the middle of a later test."""
    yield
    if os.environ.get("MCKPP_TEST_NO_GC"):   # (the incident analysis of profiles/r05/incident: how much stays alive without it)
        return
    import gc

    gc.collect()


def pytest_sessionfinish(session, exitstatus):
    """MCKPP_TEST_REPORT_HELD=1: the device contexts still open at the end of the session and the host arrays they keep pinned."""
    if not os.environ.get("MCKPP_TEST_REPORT_HELD"):
        return
    import gc

    import numpy as np

    try:
        from mckpp_f90_amd import api
    except Exception:   # noqa: BLE001
        return

    def nbytes(o, seen):
        if id(o) in seen:
            return 0
        seen.add(id(o))
        if isinstance(o, np.ndarray):
            return o.nbytes if o.nbytes >= 256 * 1024 else 0
        return sum(nbytes(v, seen) for v in vars(o).values() if isinstance(v, np.ndarray) or hasattr(v, "__dict__")) if hasattr(o, "__dict__") else 0

    live = [o for o in gc.get_objects() if isinstance(o, (api.MckppHip, api.MckppHipMulti)) and getattr(o, "_h", None)]
    seen = set()
    total = sum(nbytes(v, seen) for o in live for v in o._held.values())
    print(f"\n[mckpp tests] {len(live)} device contexts open at the end of the session, {total / 2**20:.0f} MiB of host arrays held for them")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Everything compiled (HIP library, oracle)."""
    import __graft_entry__ as g

    lib = os.path.join(ROOT, "mckpp_f90_amd", "libmckpp_hip.so")
    orc_lib = os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(lib) and os.path.exists(orc_lib)):
        g.build()
    return True
