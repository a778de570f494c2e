import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The checker libraries built from the reference tree (oracle/_ref: its shaders, and its host classes against the recording device) are git-ignored build products:
    # where the reference is mounted they are (re)built here when missing or stale -- stamp-checked, seconds when up to date, a minute or two on a fresh checkout --
    # so that a fresh clone runs the tests that hold the checker to the reference instead of skipping them.  On the GPU box the prebuilt files travel.
    try:
        import importlib.util

        spec = importlib.util.spec_from_file_location("_oracle_build", os.path.join(ROOT, "oracle", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        if os.path.isdir(os.path.join(mod.REFERENCE_ROOT, "Shaders")):
            mod.build_ref()
            mod.build_refhost()
    except Exception as e:  # noqa: BLE001 -- (the tests that need the libraries skip or fail with their own message)
        sys.stderr.write(f"conftest: oracle/_ref was not (re)built: {e}\n")


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
