import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def ref():
    """The reference compiled for the CPU (oracle/_ref); skipped where it has not been built / did not travel."""
    import pyref

    lib = pyref.ref_lib()
    if lib is None:
        pytest.skip("oracle/_ref/libmifx_ref.so not available")
    return lib


@pytest.fixture(scope="session")
def oracle():
    import pyref

    return pyref.oracle_lib()


@pytest.fixture(scope="session")
def mifx_lib():
    """libmifx.so, built on demand (hipcc cross-compiles without a GPU)."""
    from diligentfx_amd import binding

    if not os.path.exists(binding.LIB_PATH):
        from diligentfx_amd import build

        build.build()
    return binding.load()
