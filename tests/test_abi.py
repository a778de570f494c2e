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


def test_pbr_frame_attribs_block_is_read_byte_for_byte(mifx_lib):
    """mifx_pbr_shade_attribs_from_frame_attribs: the reference's PBRFrameAttribs block (Camera | PrevCamera | PBRRendererShaderParameters | Lights[N] | ShadowMaps[M],
    RenderPBR_Structures.fxh:11-24, PBR_Structures.fxh:126-180) parsed where the renderer's limits put its members; wrong sizes and limits are refused."""
    from diligentfx_amd import binding as B

    assert mifx_lib.mifx_sizeof(b"pbr_renderer_shader_parameters") == 144 == ctypes.sizeof(B.PBRRendererShaderParameters)
    assert mifx_lib.mifx_sizeof(b"pbr_material_basic_attribs") == 96 == ctypes.sizeof(B.PBRMaterialBasicAttribs)
    cam, prev = B.CameraAttribs(), B.CameraAttribs()
    cam.f4ViewportSize[:] = [640.0, 360.0, 1 / 640.0, 1 / 360.0]
    cam.fExposure = 1.25
    r = B.PBRRendererShaderParameters()
    r.IBLScale[:] = [1.1, 0.9, 1.0, 1.0]
    r.OcclusionStrength, r.EmissionScale, r.PrefilteredCubeLastMip, r.LightCount = 0.8, 1.5, 7.0, 2
    lights = [B.PBRLightAttribs(1, 0, 0, 0, 0.3, -0.9, 0.2, 1, 3.0, 2.5, 2.0, 0, 0, 0, 0, 0), B.PBRLightAttribs(3, 2.0, 6.0, -3.0, -0.2, -0.9, 0.3, -1, 40.0, 35.0, 30.0, 160000.0, 8.0, -6.8, 0, 0)]
    mat = B.PBRMaterialBasicAttribs()
    mat.Workflow = 1
    shadow = (ctypes.c_float * 24)(*range(24))
    block = B.pbr_frame_attribs(cam, prev, r, lights, 4, [shadow], 2)
    assert len(block) == 2 * 576 + 144 + 4 * 64 + 2 * 96
    out_a, out_cam, out_sm = B.PBRShadeAttribs(), B.CameraAttribs(), (ctypes.c_float * 48)()
    call = lambda blk, ml, ms, m=ctypes.byref(mat): mifx_lib.mifx_pbr_shade_attribs_from_frame_attribs(blk, ctypes.c_uint64(len(blk)), ctypes.c_uint32(ml), ctypes.c_uint32(ms), m,  # noqa: E731
                                                                                                       ctypes.byref(out_a), ctypes.byref(out_cam), out_sm)
    assert call(block, 4, 2) == 0
    assert bytes(out_cam) == bytes(cam) and out_a.LightCount == 2 and out_a.Workflow == 1 and abs(out_a.PrefilteredCubeLastMip - 7.0) < 1e-9
    assert list(out_a.IBLScale) == list(r.IBLScale) and abs(out_a.OcclusionStrength - 0.8) < 1e-7 and abs(out_a.EmissionScale - 1.5) < 1e-7
    assert bytes(out_a.Lights[0]) == bytes(lights[0]) and bytes(out_a.Lights[1]) == bytes(lights[1]) and list(out_sm)[:24] == list(range(24))
    assert call(block, 4, 2, None) == 0 and out_a.Workflow == 0  # no material block: metallic-roughness
    assert call(block, 3, 2) < 0 and b"bytes" in mifx_lib.mifx_last_error()       # the block does not have the size of these limits
    assert call(block[:-96], 4, 2) < 0
    assert call(block, 17, 0) < 0                                                 # beyond PBR_Renderer's maximum
    r.LightCount = 5
    assert call(B.pbr_frame_attribs(cam, prev, r, lights, 4, [shadow], 2), 4, 2) < 0  # more lights than the block holds
    r.LightCount, r.DebugView = 2, 3
    assert call(B.pbr_frame_attribs(cam, prev, r, lights, 4, [shadow], 2), 4, 2) < 0  # debug views are outside this path


def test_product_has_no_cpu_fallback():
    """The product path must fail loudly without the HIP library and must not reference the oracle."""
    for root, _, files in os.walk(os.path.join(ROOT, "diligentfx_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) and f != "synth.py":
                text = open(os.path.join(root, f)).read()
                assert "pyref" not in text and "mifx_oracle" not in text and "libmifx_ref" not in text, os.path.join(root, f)


def test_pbr_layers_from_the_reference_material_block(mifx_lib):
    """mifx_pbr_layers_from_material_info reads Iridescence.IOR and Anisotropy.Rotation at the offsets, and demands the size, that the reference's own PBRMaterialShaderInfo
    (Shaders/PBR/public/PBR_Structures.fxh:291-317) has for the layer set -- the offsets come from that header compiled per permutation (oracle/ref/ref_pl_body.inc); host only."""
    import struct

    import pyref
    from diligentfx_amd import binding as B

    ref = pyref.ref_lib()
    if ref is None or not ref.has("ref_pbr_shade_layers_all_material_layout"):
        pytest.skip("needs oracle/_ref with the layered permutations")
    fn = mifx_lib.mifx_pbr_layers_from_material_info
    fn.restype = ctypes.c_int
    for perm, flags in (("clearcoat", 1), ("sheen", 2), ("anisotropy", 4), ("iridescence", 8), ("transmission", 16), ("all", 31)):
        lay = (ctypes.c_int * 5)()
        f = getattr(ref.lib, f"ref_pbr_shade_layers_{perm}_material_layout")
        f.restype = ctypes.c_int
        assert f(lay) == 0
        size, off_rot, off_ior, n_tex, off_tex = list(lay)
        assert n_tex == 3 and off_tex == size - 48 * n_tex
        block = bytearray(struct.pack(f"<{size // 4}f", *[0.001 * i for i in range(size // 4)]))
        struct.pack_into("<i", block, 48, 1)  # Basic.Workflow = PBR_WORKFLOW_SPECULAR_GLOSSINESS
        if off_rot >= 0:
            struct.pack_into("<f", block, off_rot, 0.75)
        if off_ior >= 0:
            struct.pack_into("<f", block, off_ior, 1.45)
        buf = (ctypes.c_char * size).from_buffer(block)
        layers, basic = B.PBRLayers(), (ctypes.c_float * 24)()
        layers.iridescence_ior, layers.anisotropy_rotation = -1.0, -1.0
        assert fn(buf, ctypes.c_uint64(size), ctypes.c_uint32(flags), 0, ctypes.c_uint32(n_tex), ctypes.byref(layers), basic) == 0, mifx_lib.mifx_last_error()
        assert layers.flags == flags
        assert layers.anisotropy_rotation == (pytest.approx(0.75) if off_rot >= 0 else -1.0)
        assert layers.iridescence_ior == (pytest.approx(1.45) if off_ior >= 0 else -1.0)
        assert struct.unpack_from("<i", bytes(basic), 48)[0] == 1 and basic[0] == 0.0 and basic[1] == pytest.approx(0.001)
        # a block of another size (one texture block less, a volume block that is not there) is refused
        assert fn(buf, ctypes.c_uint64(size), ctypes.c_uint32(flags), 0, ctypes.c_uint32(n_tex - 1), ctypes.byref(layers), None) == -1
        assert fn(buf, ctypes.c_uint64(size), ctypes.c_uint32(flags), 1, ctypes.c_uint32(n_tex), ctypes.byref(layers), None) == -1
    assert fn(None, ctypes.c_uint64(96), ctypes.c_uint32(0), 0, ctypes.c_uint32(0), None, None) == -1
