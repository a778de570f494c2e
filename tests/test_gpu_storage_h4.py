"""The RGBA16_FLOAT storage build of the library (libmifx_h4.so: every 4-channel image -- G-buffer colour / normal / material inputs, radiance, the SSR targets, the TAA
accumulation buffers, Bloom, the frame -- stored as the reference stores its colour targets) against the checker with format emulation.  The storage
mode is a property of the loaded library, so the checks run in a process of their own (tests/h4_checks.py) with MIFX_STORAGE=h4."""
import os
import subprocess
import sys

import pytest

from util import ROOT


def test_h4_library_is_built_and_reports_its_mode(mifx_lib):
    import ctypes

    from diligentfx_amd import build

    assert os.path.exists(build.OUT_H4), "libmifx_h4.so missing: python -m diligentfx_amd.build"
    h4 = ctypes.CDLL(build.OUT_H4)
    h4.mifx_storage_mode.restype = ctypes.c_uint32
    assert h4.mifx_storage_mode() == 1 and mifx_lib.mifx_storage_mode() == 0
    assert h4.mifx_abi_version() == mifx_lib.mifx_abi_version()


@pytest.mark.gpu
@pytest.mark.parametrize("section", ["chain", "fusion", "dof", "dof_chain", "dof_passes", "half_precision_depth", "sharded", "layers"])
def test_native_storage_build_against_the_format_emulating_checker(section):
    """chain: six frames of the whole chain, every effect's output against the checker with the reference's target formats; fusion: every fusion switch bit-identical;
    dof: depth of field end to end; dof_passes: its eleven passes one by one (R16_FLOAT / R16_UNORM circle-of-confusion targets); half_precision_depth: FEATURE_FLAG_HALF_PRECISION_DEPTH of PostFX / SSAO
    (R16_UNORM depth targets); sharded: two ranks bit-identical."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "h4_checks.py"), section], cwd=ROOT, env=dict(os.environ, MIFX_STORAGE="h4"), capture_output=True, text=True, timeout=600)
    print(r.stdout[-4000:])
    assert r.returncode == 0 and "h4 checks OK" in r.stdout and f"h4 section {section} OK" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
