"""GPU parity of the PBR shading entry (P1-P9) and of the SSR/SSAO composite (M1)."""
import numpy as np
import os

import pytest
import torch

from util import assert_close, blue_noise_tables, to_np

pytestmark = pytest.mark.gpu


def checker(symbol):
    import pyref

    r = pyref.ref_lib()
    if r is not None:
        return r, "ref_"
    o = pyref.oracle_lib()
    if not o.has("oracle_" + symbol):
        pytest.skip("no checker available for " + symbol)
    return o, "oracle_"


@pytest.fixture(scope="module")
def ibl_np():
    import chain_util

    lib, pfx = checker("ibl_brdf_lut")
    return chain_util.make_ibl(lib, pfx)


def ibl_to_device(ibl_np, device):
    from diligentfx_amd import api

    return api.IBLResources(torch.from_numpy(ibl_np["lut"]).to(device), [torch.from_numpy(m).to(device) for m in ibl_np["irradiance"]],
                            [torch.from_numpy(m).to(device) for m in ibl_np["prefiltered"]])


@pytest.mark.parametrize("size", [(160, 96), (131, 77)])
@pytest.mark.parametrize("extras", [False, True])
def test_pbr_shade(mifx_lib, ibl_np, size, extras):
    import chain_util
    from diligentfx_amd import api, synth

    lib, pfx = checker("pbr_shade")
    w, h = size
    ctx = api.PostFXContext(0)
    f = synth.make_frame(synth.Scene(), 4, w, h, ctx.device)
    g = {k: f[k] for k in ("base_color", "normal", "material", "depth")}
    if extras:
        gen = torch.Generator(device="cpu").manual_seed(3)
        g["emissive"] = (torch.rand(h, w, 4, generator=gen) * 0.3).to(ctx.device)
        g["occlusion"] = (0.3 + 0.7 * torch.rand(h, w, generator=gen)).to(ctx.device)
    sa = chain_util.shade_attribs(len(ibl_np["prefiltered"]) - 1)
    sa.OcclusionStrength, sa.EmissionScale = 0.8, 1.5
    sa.IBLScale[:] = [1.1, 0.9, 1.0, 1.0]
    # a spot light exercises the third light type
    from diligentfx_amd import binding as B

    sa.Lights[sa.LightCount] = B.PBRLightAttribs(3, 2.0, 6.0, -3.0, -0.2, -0.9, 0.3, -1, 40.0, 35.0, 30.0, 20.0 ** 4, 8.0, -6.8, 0.0, 0.0)
    sa.LightCount += 1
    bg = (0.02, 0.03, 0.05, 0.0)
    rad, spec = api.pbr_shade(ctx, g, f["camera"], sa, ibl_to_device(ibl_np, ctx.device), background=bg)
    wr, ws = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    gn = {k: to_np(v) for k, v in g.items()}
    lib.call(pfx + "pbr_shade", [gn["base_color"], gn["normal"], gn["material"], gn["depth"], gn.get("emissive"), gn.get("occlusion"), ibl_np["lut"],
                                 ibl_np["irradiance"], ibl_np["prefiltered"]], [wr, ws], cam0=bytes(f["camera"]), attribs=bytes(sa), fval=list(bg))
    # cube-face selection and the nearest-texel re-projection at face edges are discontinuous in the direction
    assert_close(to_np(rad), wr, max_outlier_frac=0.0, what="radiance")
    assert_close(to_np(spec), ws, max_outlier_frac=0.0, what="specular IBL")
    assert float(to_np(rad)[..., :3].max()) > 0.5 and np.isfinite(to_np(rad)).all()
    # no specular-IBL target requested: same radiance
    rad2, none = api.pbr_shade(ctx, g, f["camera"], sa, ibl_to_device(ibl_np, ctx.device), background=bg, want_specular_ibl=False)
    assert none is None and torch.allclose(rad2, rad, rtol=1e-5, atol=1e-6)
    ctx.close()


def test_pbr_shade_on_the_reference_frame_block(mifx_lib, ibl_np):
    """mifx_pbr_shade_execute_frame_attribs: the shade fed with the renderer's own PBRFrameAttribs / PBRMaterialBasicAttribs bytes gives the texels of
    mifx_pbr_shade_execute with the equivalent mifx_pbr_shade_attribs, bit for bit."""
    import ctypes

    import chain_util
    from diligentfx_amd import api, binding as B, synth

    w, h = 160, 96
    ctx = api.PostFXContext(0)
    f = synth.make_frame(synth.Scene(), 4, w, h, ctx.device)
    g = {k: f[k] for k in ("base_color", "normal", "material", "depth")}
    sa = chain_util.shade_attribs(len(ibl_np["prefiltered"]) - 1)
    sa.OcclusionStrength, sa.EmissionScale = 0.8, 1.5
    sa.IBLScale[:] = [1.1, 0.9, 1.0, 1.0]
    ibl = ibl_to_device(ibl_np, ctx.device)
    bg = (0.02, 0.03, 0.05, 0.0)
    want_rad, want_spec = api.pbr_shade(ctx, g, f["camera"], sa, ibl, background=bg)
    r = B.PBRRendererShaderParameters()
    r.IBLScale[:] = list(sa.IBLScale)
    r.OcclusionStrength, r.EmissionScale, r.PrefilteredCubeLastMip, r.LightCount = sa.OcclusionStrength, sa.EmissionScale, sa.PrefilteredCubeLastMip, sa.LightCount
    r.AverageLogLum, r.MiddleGray, r.WhitePoint, r.Time = 0.3, 0.18, 3.0, 12.5  # (fields of the block that this path does not read)
    block = B.pbr_frame_attribs(f["camera"], f["prev_camera"], r, [sa.Lights[i] for i in range(sa.LightCount)], 16)
    imgs = {k: B.image(v) for k, v in g.items()}
    gb = B.GBuffer(ctypes.pointer(imgs["base_color"]), ctypes.pointer(imgs["normal"]), ctypes.pointer(imgs["material"]), ctypes.pointer(imgs["depth"]), None, None)
    rad, spec = torch.zeros_like(want_rad), torch.zeros_like(want_spec)
    o0, o1 = B.image(rad), B.image(spec)
    ctx.sync_stream()
    B.check(mifx_lib.mifx_pbr_shade_execute_frame_attribs(ctx.handle, ctypes.byref(gb), block, ctypes.c_uint64(len(block)), ctypes.c_uint32(16), ctypes.c_uint32(0), None,
                                                          ctypes.byref(ibl.struct), None, ctypes.c_uint32(3), (ctypes.c_float * 4)(*bg), ctypes.byref(o0), ctypes.byref(o1)))
    assert torch.equal(rad, want_rad) and torch.equal(spec, want_spec) and float(rad[..., :3].max()) > 0.5
    ctx.close()


def test_pbr_shade_specular_glossiness(mifx_lib, ibl_np):
    """PBR_WORKFLOW_SPECULAR_GLOSSINESS (PBR_Shading.fxh:390-403, SolveMetallic :99-117): the shade on a PhysicalDesc plane, and the Material target the
    reference writes for such a surface (USD_Renderer.cpp:98), against the checker."""
    import ctypes

    import chain_util
    from diligentfx_amd import api, binding as B, synth
    from test_oracle_vs_ref import specgloss_material

    lib, pfx = checker("specgloss_material")
    w, h = 160, 96
    ctx = api.PostFXContext(0)
    f = synth.make_frame(synth.Scene(), 4, w, h, ctx.device)
    desc_np = specgloss_material(to_np(f["material"]), to_np(f["base_color"]))
    g = {"base_color": f["base_color"], "normal": f["normal"], "material": torch.from_numpy(desc_np).to(ctx.device), "depth": f["depth"]}
    sa = chain_util.shade_attribs(len(ibl_np["prefiltered"]) - 1)
    sa.Workflow = 1
    bg = (0.02, 0.03, 0.05, 0.0)
    rad, spec = api.pbr_shade(ctx, g, f["camera"], sa, ibl_to_device(ibl_np, ctx.device), background=bg)
    wr, ws = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    lib.call(pfx + "pbr_shade", [to_np(f["base_color"]), to_np(f["normal"]), desc_np, to_np(f["depth"]), None, None, ibl_np["lut"], ibl_np["irradiance"],
                                 ibl_np["prefiltered"]], [wr, ws], cam0=bytes(f["camera"]), attribs=bytes(sa), fval=list(bg))
    assert_close(to_np(rad), wr, max_outlier_frac=0.0, what="specular-glossiness radiance")
    assert_close(to_np(spec), ws, max_outlier_frac=0.0, what="specular-glossiness specular IBL")
    mat = torch.empty(h, w, 4, device=ctx.device)
    i = [B.image(t) for t in (f["base_color"], g["material"], mat)]
    B.check(ctx.lib.mifx_pbr_specgloss_to_material(ctx.handle, *[ctypes.byref(x) for x in i]))
    wm = np.zeros((h, w, 4), np.float32)
    lib.call(pfx + "specgloss_material", [to_np(f["base_color"]), desc_np], [wm])
    assert_close(to_np(mat), wm, what="Material target of a specular-glossiness surface")
    assert wm[..., 1].max() > 0.5 and (wm[: h // 8, :, 1] == 0).all()
    sa.Workflow = 7
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        api.pbr_shade(ctx, g, f["camera"], sa, ibl_to_device(ibl_np, ctx.device), background=bg)
    ctx.close()


def test_pbr_shade_argument_errors(mifx_lib, ibl_np):
    import chain_util
    from diligentfx_amd import api, binding as B, synth

    ctx = api.PostFXContext(0)
    f = synth.make_frame(synth.Scene(), 0, 64, 48, ctx.device)
    g = {k: f[k] for k in ("base_color", "normal", "material", "depth")}
    ibl = ibl_to_device(ibl_np, ctx.device)
    sa = chain_util.shade_attribs(4)
    sa.Lights[0].ShadowMapIndex = 0
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        api.pbr_shade(ctx, g, f["camera"], sa, ibl)  # a shadow-mapped light needs mifx_pbr_shade_execute_with_shadows
    slices, infos = chain_util.make_shadow_inputs(16)
    sm = torch.from_numpy(np.stack(slices)).to(ctx.device)
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        api.pbr_shade(ctx, g, f["camera"], sa, ibl, shadows=(sm, infos, 4))  # no such PCF filter size
    sa.Lights[0].ShadowMapIndex = 5
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        api.pbr_shade(ctx, g, f["camera"], sa, ibl, shadows=(sm, infos, 3))  # index beyond the shadow-map infos
    sa = chain_util.shade_attribs(4)
    sa.LightCount = 17
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        api.pbr_shade(ctx, g, f["camera"], sa, ibl)
    g["depth"] = g["depth"][:10]
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        api.pbr_shade(ctx, g, f["camera"], chain_util.shade_attribs(4), ibl, out_radiance=torch.empty(48, 64, 4, device=ctx.device))
    ctx.close()


@pytest.mark.parametrize("tm_mode", [0, 4, 8])
def test_composite(mifx_lib, ibl_np, tm_mode):
    from diligentfx_amd import api, binding as B, synth

    lib, pfx = checker("composite")
    w, h = 150, 90
    ctx = api.PostFXContext(0)
    f = synth.make_frame(synth.Scene(), 2, w, h, ctx.device)
    gen = torch.Generator(device="cpu").manual_seed(11)
    rnd = lambda *s: torch.rand(*s, generator=gen).to(ctx.device)  # noqa: E731
    color = torch.cat([rnd(h, w, 3) * 3.0, f["base_color"][..., 3:4]], -1).contiguous()  # alpha = opacity (0 on the background)
    spec, ssr, ssao = rnd(h, w, 4), rnd(h, w, 4), rnd(h, w)
    lut = torch.from_numpy(ibl_np["lut"]).to(ctx.device)
    tm = B.ToneMappingAttribs.default(tm_mode) if tm_mode else None
    got = to_np(api.composite(ctx, color, spec, ssr, ssao, f["normal"], f["base_color"], f["material"], lut, f["camera"], 0.9, 0.8, tone_mapping=tm, ave_log_lum=0.3))
    want = np.zeros((h, w, 4), np.float32)
    lib.call(pfx + "composite", [to_np(color), to_np(spec), to_np(ssr), to_np(ssao), to_np(f["normal"]), to_np(f["base_color"]), to_np(f["material"]), ibl_np["lut"]],
             [want], cam0=bytes(f["camera"]), fval=[0.9, 0.8])
    if tm_mode:
        tmd = np.zeros_like(want)
        lib.call(pfx + "tonemap", [want], [tmd], attribs=bytes(tm), fval=[0.3], ival=[0])
        want = tmd
    assert_close(got, want, what=f"composite tm={tm_mode}")
    ctx.close()


def test_pbr_shade_full_size_parity(mifx_lib, ibl_np):
    """BASELINE configs[2]: PBR GGX + IBL shade of a 3840x2160 G-buffer (base colour / normal / material / depth) against the checker."""
    import chain_util
    from diligentfx_amd import api, synth

    lib, pfx = checker("pbr_shade")
    w, h = 3840, 2160
    ctx = api.PostFXContext(0)
    f = synth.make_frame(synth.Scene(), 17, w, h, ctx.device)
    g = {k: f[k] for k in ("base_color", "normal", "material", "depth")}
    sa = chain_util.shade_attribs(len(ibl_np["prefiltered"]) - 1)
    bg = (0.02, 0.03, 0.05, 0.0)
    rad, spec = api.pbr_shade(ctx, g, f["camera"], sa, ibl_to_device(ibl_np, ctx.device), background=bg)
    wr, ws = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    gn = {k: to_np(v) for k, v in g.items()}
    lib.call(pfx + "pbr_shade", [gn["base_color"], gn["normal"], gn["material"], gn["depth"], None, None, ibl_np["lut"], ibl_np["irradiance"], ibl_np["prefiltered"]],
             [wr, ws], cam0=bytes(f["camera"]), attribs=bytes(sa), fval=list(bg))
    assert_close(to_np(rad), wr, max_outlier_frac=0.0, what="radiance 3840x2160")
    assert_close(to_np(spec), ws, max_outlier_frac=0.0, what="specular IBL 3840x2160")
    ctx.close()


@pytest.mark.parametrize("mode,gamma,mip", [(0, 0, 1.0), (0, 0, 2.4), (4, 1, 1.0), (8, 0, 0.0)])
def test_envmap_background_parity(mifx_lib, mode, gamma, mip):
    """mifx_envmap_render == EnvMapRenderer (EnvMap.psh): colour and motion vectors of the background pixels, everything else untouched."""
    from diligentfx_amd import api, binding as B, synth
    from test_oracle_vs_ref import run_envmap

    import pyref

    ref, oracle = pyref.ref_lib(), pyref.oracle_lib()
    w, h = 208, 120
    ctx = api.PostFXContext(0)
    f = synth.make_frame(synth.Scene(), 9, w, h, ctx.device)
    env_mips = api.cube_box_mips(synth.make_sky_cube(32, ctx.device).clamp(max=500.0))
    color = torch.full((h, w, 4), -7.0, device=ctx.device)
    motion = torch.full((h, w, 2), -7.0, device=ctx.device)
    scale = (1.5, 1.0, 0.75)
    api.render_env_map(ctx, env_mips, f["depth"], color, motion, f["camera"], f["prev_camera"], B.ToneMappingAttribs.default(mode), 0.3, mip, 0.25, scale,
                       (api.ENVMAP_OPTION_FLAG_CONVERT_OUTPUT_TO_SRGB if gamma else 0) | api.ENVMAP_OPTION_FLAG_COMPUTE_MOTION_VECTORS)
    torch.cuda.synchronize()
    inp = {"env": [to_np(m) for m in env_mips], "depth": to_np(f["depth"]), "cam": bytes(f["camera"]), "prev": bytes(f["prev_camera"])}
    # the reference is compiled for tone mapping NONE (Hydrogent's call) and Uncharted2 + gamma; other operators against the hand-written oracle
    lib, pfx = (ref, "ref_") if (ref is not None and (mode, gamma) in ((0, 0), (4, 1))) else (oracle, "oracle_")
    want_c, want_m = run_envmap(lib, pfx, inp, mode, gamma, mip, 0.25, scale)
    assert_close(to_np(color), want_c, what=f"env map colour mode {mode}")
    assert_close(to_np(motion), want_m, atol=1e-6, what="env map motion")
    bg = inp["depth"] >= 1.0
    assert (to_np(color)[~bg] == -7.0).all() and (to_np(motion)[~bg] == -7.0).all() and 0.05 < bg.mean() < 0.95
    # OPTION_FLAG_USE_REVERSE_DEPTH on a frame rendered with the reversed projection: the same pixels, against the oracle (no reference build)
    fr = synth.make_frame(synth.Scene(), 9, w, h, ctx.device, reversed_depth=True)
    c2, m2 = torch.full((h, w, 4), -7.0, device=ctx.device), torch.full((h, w, 2), -7.0, device=ctx.device)
    api.render_env_map(ctx, env_mips, fr["depth"], c2, m2, fr["camera"], fr["prev_camera"], B.ToneMappingAttribs.default(mode), 0.3, mip, 0.25, scale,
                       (api.ENVMAP_OPTION_FLAG_CONVERT_OUTPUT_TO_SRGB if gamma else 0) | api.ENVMAP_OPTION_FLAG_COMPUTE_MOTION_VECTORS | 4)
    wc, wm = np.full((h, w, 4), -7.0, np.float32), np.full((h, w, 2), -7.0, np.float32)
    from util import tone_mapping_attribs_bytes

    oracle.call("oracle_envmap", [inp["env"], to_np(fr["depth"])], [wc, wm], cam0=bytes(fr["camera"]), cam1=bytes(fr["prev_camera"]), attribs=tone_mapping_attribs_bytes(mode),
                fval=[0.3, mip, 0.25, *scale], ival=[gamma, 1, 0, 0, 0, 0, 0, 1])
    assert_close(to_np(c2), wc, what="env map colour, reversed depth")
    assert_close(to_np(m2), wm, atol=1e-6, what="env map motion, reversed depth")
    assert np.array_equal(to_np(c2)[..., 3] == 0.25, to_np(fr["depth"]) == 0.0) and (to_np(fr["depth"]) == 0.0).mean() > 0.05
    api.render_env_map(ctx, env_mips, f["depth"], color, None, f["camera"], f["prev_camera"])  # without a motion target
    torch.cuda.synchronize()
    ctx.close()


def test_sphere_map_environment(mifx_lib):
    """Equirectangular environment maps (ENV_MAP_TYPE_SPHERE): the two IBL precompute passes (= equirect -> cube at roughness 0) and the background pass."""
    from diligentfx_amd import api, synth
    from test_oracle_vs_ref import run_envmap, sphere_map_mips  # noqa: F401

    import pyref

    ref, oracle = pyref.ref_lib(), pyref.oracle_lib()
    ctx = api.PostFXContext(0)
    env_np = sphere_map_mips()
    env = [torch.from_numpy(m).to(ctx.device) for m in env_np]
    irr, pre = api.ibl_from_sphere_map(ctx, env, irradiance_size=8, prefiltered_size=16, diffuse_samples=256, specular_samples=48)
    lib, pfx = (ref, "ref_") if ref is not None else (oracle, "oracle_")

    def call(name, outs, **kw):
        if pfx == "ref_":
            lib.call("ref_" + name + "_sphere", [env_np], outs, **kw)
        else:
            kw["ival"] = list(kw["ival"]) + [1]
            lib.call("oracle_" + name, [env_np], outs, **kw)

    want = np.zeros((6 * 8, 8, 4), np.float32)
    call("ibl_irradiance_map", [want], ival=[256])
    assert_close(to_np(irr), want, max_outlier_frac=0.0, what="irradiance from a sphere map")
    levels = len(pre)
    for m, p in enumerate(pre):
        s = 16 >> m
        want = np.zeros((6 * s, s, 4), np.float32)
        call("ibl_prefilter_env_map", [want], ival=[48], fval=[m / (levels - 1)])
        # (the mip level of a tap is continuous in the solid angle, so there are no selection flips; acos / atan2 / asin differ by an ulp or two from libm)
        assert_close(to_np(p), want, max_outlier_frac=0.0, what=f"prefiltered mip {m} from a sphere map")
    assert float(pre[0][..., :3].max()) > 50.0  # the sun made it onto the cube
    # background pass from the sphere map
    w, h = 176, 104
    f = synth.make_frame(synth.Scene(), 9, w, h, ctx.device)
    color, motion = torch.full((h, w, 4), -7.0, device=ctx.device), torch.full((h, w, 2), -7.0, device=ctx.device)
    api.render_env_map(ctx, env, f["depth"], color, motion, f["camera"], f["prev_camera"], None, 0.3, 1.5, 0.0, (1.0, 1.0, 1.0))
    torch.cuda.synchronize()
    wc, wm = np.full((h, w, 4), -7.0, np.float32), np.full((h, w, 2), -7.0, np.float32)
    from util import tone_mapping_attribs_bytes

    args = dict(cam0=bytes(f["camera"]), cam1=bytes(f["prev_camera"]), attribs=tone_mapping_attribs_bytes(0), fval=[0.3, 1.5, 0.0, 1.0, 1.0, 1.0])
    if pfx == "ref_":
        lib.call("ref_envmap_sphere", [env_np, to_np(f["depth"])], [wc, wm], **args)
    else:
        lib.call("oracle_envmap", [env_np, to_np(f["depth"])], [wc, wm], ival=[0, 1, 1], **args)
    assert_close(to_np(color), wc, what="sphere env map colour")
    assert_close(to_np(motion), wm, atol=1e-6, what="sphere env map motion")
    ctx.close()


@pytest.mark.parametrize("pcf", [2, 3, 5, 7])
def test_pbr_shade_with_shadows(mifx_lib, ibl_np, pcf):
    """mifx_pbr_shade_execute_with_shadows == RenderPBR.psh with ENABLE_SHADOWS: lights with a shadow map are attenuated by FilterShadowMapFixedPCF (PCF.fxh)."""
    import chain_util
    from diligentfx_amd import api, synth

    import pyref

    ref, oracle = pyref.ref_lib(), pyref.oracle_lib()
    w, h = 224, 128
    ctx = api.PostFXContext(0)
    f = synth.make_frame(synth.Scene(), 4, w, h, ctx.device)
    g = {k: f[k] for k in ("base_color", "normal", "material", "depth")}
    sa = chain_util.shadowed_shade_attribs(len(ibl_np["prefiltered"]) - 1)
    slices, infos = chain_util.make_shadow_inputs()
    sm = torch.from_numpy(np.stack(slices)).to(ctx.device)
    bg = (0.02, 0.03, 0.05, 0.0)
    ibl = ibl_to_device(ibl_np, ctx.device)
    rad, spec = api.pbr_shade(ctx, g, f["camera"], sa, ibl, background=bg, shadows=(sm, infos, pcf))
    torch.cuda.synchronize()
    ins = [to_np(g["base_color"]), to_np(g["normal"]), to_np(g["material"]), to_np(g["depth"]), None, None, ibl_np["lut"], ibl_np["irradiance"], ibl_np["prefiltered"], slices,
           infos.reshape(1, -1)]
    want, wspec = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    if ref is not None:
        ref.call(f"ref_pbr_shade_shadows{pcf}", ins, [want, wspec], cam0=bytes(f["camera"]), attribs=bytes(sa), fval=list(bg))
    else:
        oracle.call("oracle_pbr_shade", ins, [want, wspec], cam0=bytes(f["camera"]), attribs=bytes(sa), fval=list(bg), ival=[pcf])
    # "reference < texel" on computed light-space depths: a tap exactly on the threshold may flip
    assert_close(to_np(rad), want, max_outlier_frac=0.0, what=f"shadowed shade PCF {pcf}")
    assert_close(to_np(spec), wspec, what="specular IBL (no shadows on IBL)")
    plain, _ = api.pbr_shade(ctx, g, f["camera"], chain_util.shade_attribs(len(ibl_np["prefiltered"]) - 1), ibl, background=bg)
    unshadowed_spot = chain_util.shadowed_shade_attribs(len(ibl_np["prefiltered"]) - 1)
    for i in range(unshadowed_spot.LightCount):
        unshadowed_spot.Lights[i].ShadowMapIndex = -1
    lit, _ = api.pbr_shade(ctx, g, f["camera"], unshadowed_spot, ibl, background=bg)
    darker = ((rad[..., :3] < lit[..., :3] - 1e-4).any(-1)).float().mean()
    assert 0.05 < float(darker) < 0.95 and bool((rad[..., :3] <= lit[..., :3] + 1e-5).all()) and not torch.equal(plain, lit)
    ctx.close()


def test_pbr_shade_on_native_gbuffer(mifx_lib, ibl_np):
    """mifx_pbr_shade_execute_native on the Hydrogent G-buffer formats (HnBeginFrameTask.cpp:63-69) == import of every plane, the fp32 shade, export of the
    targets: the kernel body is the same code, so the stored RGBA16_FLOAT texels agree bit for bit."""
    import chain_util
    from diligentfx_amd import api, synth

    w, h = 224, 128
    ctx = api.PostFXContext(0)
    f = synth.make_frame(synth.Scene(), 4, w, h, ctx.device)
    emissive = (torch.rand(h, w, 4, device=ctx.device) * 0.3).contiguous()
    occlusion = (0.5 + 0.5 * torch.rand(h, w, device=ctx.device)).contiguous()
    planes = {"base_color": f["base_color"], "normal": f["normal"], "material": f["material"], "depth": f["depth"], "emissive": emissive, "occlusion": occlusion}
    sa = chain_util.shade_attribs(len(ibl_np["prefiltered"]) - 1)
    ibl = ibl_to_device(ibl_np, ctx.device)
    bg = (0.02, 0.03, 0.05, 0.0)
    for names in (("base_color", "normal", "material", "depth"), tuple(planes)):
        native = {k: (api.image_export(ctx, planes[k], api.HYDROGENT_GBUFFER_FORMATS[k]), api.HYDROGENT_GBUFFER_FORMATS[k]) for k in names}
        # the three-pass route through fp32 planes
        chans = {"base_color": 4, "normal": 4, "material": 4, "depth": 1, "emissive": 4, "occlusion": 1}
        g32 = {k: api.image_import(ctx, native[k][0], w, native[k][1], channels=chans[k]) for k in names}
        rad, spec = api.pbr_shade(ctx, g32, f["camera"], sa, ibl, background=bg)
        want = [api.image_export(ctx, t, "RGBA16_FLOAT") for t in (rad, spec)]
        got = api.pbr_shade_native(ctx, native, w, f["camera"], sa, ibl, background=bg)
        torch.cuda.synchronize()
        if len(names) == 4:
            # these four planes in Hydrogent's formats can take the per-format instance of the kernel (format switches folded at compile time): the same texels as
            # the generic instance with its run-time switches
            os.environ["MIFX_NATIVE_SHADE_PER_FORMAT"] = "1"
            try:
                fixed = api.pbr_shade_native(ctx, native, w, f["camera"], sa, ibl, background=bg)
                torch.cuda.synchronize()
            finally:
                del os.environ["MIFX_NATIVE_SHADE_PER_FORMAT"]
            assert all(torch.equal(a, b) for a, b in zip(got, fixed))
        for a, b, what in zip(got, want, ("radiance", "specular IBL")):
            ha, hb = a.view(torch.int16).int(), b.view(torch.int16).int()
            diff = (ha - hb).abs()
            print(f"native shade {len(names)} planes, {what}: {int((diff != 0).sum())} of {diff.numel()} fp16 codes differ, max {int(diff.max())}")
            assert torch.equal(a, b), what
        # and it is the same picture as the fp32-contract shade of the unquantised planes (format quantisation only)
        ref32, _ = api.pbr_shade(ctx, {k: planes[k] for k in names}, f["camera"], sa, ibl, background=bg)
        half = got[0].view(torch.float16).reshape(h, w, 4).float()
        assert float((half[..., :3] - ref32[..., :3]).abs().mean()) < 0.05 * float(ref32[..., :3].abs().mean())
    # errors: a depth plane that is not R32_FLOAT, a size mismatch
    bad = dict(native)
    bad["depth"] = (api.image_export(ctx, planes["depth"], "R16_UNORM"), "R16_UNORM")
    with pytest.raises(RuntimeError):
        api.pbr_shade_native(ctx, bad, w, f["camera"], sa, ibl)
    ctx.close()
