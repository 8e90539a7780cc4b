import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# the GPU box exposes 256 logical CPUs behind a 16-CPU cgroup quota: keep OpenMP modest
os.environ.setdefault("OMP_NUM_THREADS", "8")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def gpu_available():
    try:
        import ctypes
        from unified_cvo_amd import _capi
        L = _capi.lib()
        ctx = ctypes.c_void_p()
        rc = L.cvo_ctx_create(0, ctypes.byref(ctx))
        if rc == 0:
            L.cvo_ctx_destroy(ctx)
        return rc == 0
    except Exception:
        return False
