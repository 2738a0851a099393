import os
import sys

import pytest

# torch bundles its own HIP runtime (same soname as /opt/rocm's): whichever is loaded first serves
# the whole process, and loading torch's second fails ("No HIP GPUs").  Tests that mix torch tensors
# with the C ABI therefore need torch imported before librt_amd.so, as bench.py does.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


@pytest.fixture(scope="session")
def capi():
    from raytracer_amd import capi as c
    return c


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def gpu_ctx(capi):
    ctx = capi.Context(0)
    yield ctx
    ctx.close()
