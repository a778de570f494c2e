"""CPU-side checks of the drop-in boundary: libmifx.so loads and exports every symbol include/mifx.h declares,
struct layouts agree between the header (C), the ctypes mirror and the reference sizes."""
import ctypes
import os
import re

import pytest

from util import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mifx.h")).read()
    return sorted(set(re.findall(r"MIFX_API\s+[\w\s\*]+?\b(mifx_\w+)\s*\(", text)))


def test_header_declares_the_pass_interfaces():
    syms = declared_symbols()
    for s in ["mifx_ssao_execute", "mifx_ssr_execute", "mifx_bloom_execute", "mifx_taa_execute", "mifx_tonemap_execute", "mifx_pbr_shade_execute",
              "mifx_postfx_execute", "mifx_composite_execute", "mifx_chain_execute"]:
        assert s in syms
    assert len(syms) >= 45


def test_library_exports_every_declared_symbol(mifx_lib):
    missing = [s for s in declared_symbols() if not hasattr(mifx_lib, s)]
    assert not missing, missing


def test_struct_sizes_match_reference_and_ctypes(mifx_lib):
    from diligentfx_amd import binding as B

    reference_sizes = {"camera_attribs": 576, "tone_mapping_attribs": 48, "ssao_attribs": 48, "ssr_attribs": 48, "bloom_attribs": 32, "dof_attribs": 32,
                       "taa_attribs": 16, "pbr_light_attribs": 64}
    for name, cls in B.SIZEOF_NAMES.items():
        n = mifx_lib.mifx_sizeof(name.encode())
        assert n == ctypes.sizeof(cls), (name, n, ctypes.sizeof(cls))
        if name in reference_sizes:
            assert n == reference_sizes[name], name
    assert mifx_lib.mifx_sizeof(b"no_such_struct") == 0


def test_status_strings_and_version(mifx_lib):
    assert mifx_lib.mifx_status_string(0) == b"MIFX_OK"
    assert mifx_lib.mifx_status_string(-5) == b"MIFX_ERR_NOT_IMPLEMENTED"
    assert mifx_lib.mifx_abi_version() >= 1


def test_reverse_exp_tone_map_host_helper(mifx_lib):
    """Components/src/ToneMapping.cpp:43-83: grey input, saturation 1 => exact inverse of the EXP operator."""
    import math

    ldr = (ctypes.c_float * 3)(0.5, 0.5, 0.5)
    out = (ctypes.c_float * 3)()
    assert mifx_lib.mifx_reverse_exp_tone_map(ldr, ctypes.c_float(0.18), ctypes.c_float(0.3), out) == 0
    lum_scale = 0.18 / 0.3
    assert abs(out[0] - (-math.log(0.5) / lum_scale)) < 1e-5
    fwd = 1.0 - math.exp(-out[0] * lum_scale)
    assert abs(fwd - 0.5) < 1e-5


def test_product_has_no_cpu_fallback():
    """The product path must fail loudly without the HIP library and must not reference the oracle."""
    for root, _, files in os.walk(os.path.join(ROOT, "diligentfx_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) and f != "synth.py":
                text = open(os.path.join(root, f)).read()
                assert "pyref" not in text and "mifx_oracle" not in text and "libmifx_ref" not in text, os.path.join(root, f)
