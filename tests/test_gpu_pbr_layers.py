"""GPU parity of the shade with material layers (mifx_pbr_shade_execute_layers): clear coat, sheen, anisotropy, iridescence, transmission -- the ENABLE_* blocks of
Shaders/PBR/public/PBR_Shading.fxh:40-62, each a pipeline permutation of the reference.  The checker is the reference's own PBR_Shading.fxh / PBR_Common.fxh /
Iridescence.fxh compiled with the permutation's macros (oracle/ref/ref_pl_*.cpp); there is no hand port of these blocks, so the tests need oracle/_ref."""
import numpy as np
import pytest
import torch

from layers_util import BACKGROUND, CASES, IOR, PERMUTATIONS, ROTATION, SHADOW_CASES, checker_result, make_case, make_layers, ref_checker
from util import assert_close, to_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ibl_np():
    import chain_util

    return chain_util.make_ibl(ref_checker(), "ref_")


def ibl_to_device(ibl_np, device):
    from diligentfx_amd import api

    return api.IBLResources(torch.from_numpy(ibl_np["lut"]).to(device), [torch.from_numpy(m).to(device) for m in ibl_np["irradiance"]],
                            [torch.from_numpy(m).to(device) for m in ibl_np["prefiltered"]])


@pytest.mark.parametrize("perm,size,optional", CASES)
def test_pbr_shade_layers(mifx_lib, ibl_np, perm, size, optional):
    """optional: the clear-coat normal and the tangent planes are bound (USE_CLEAR_COAT_NORMAL_MAP / USE_VERTEX_TANGENTS); otherwise the layer takes the G-buffer normal
    and the tangent (1, 0, 0) -- RenderPBR.psh:201-215,274-286."""
    from diligentfx_amd import api

    lib = ref_checker()
    ctx = api.PostFXContext(0)
    f, gn, sa, planes, albedo, charlie = make_case(perm, size, ibl_np, ctx.device)
    g = {k: torch.from_numpy(v).to(ctx.device) for k, v in gn.items()}
    dev = {k: torch.from_numpy(v).to(ctx.device) for k, v in planes.items()}
    dev["transmission"] = dev["transmission"][..., 0].contiguous()  # F32
    if not optional:
        dev.pop("clearcoat_normal")
        dev.pop("tangent")
    dev["sheen_albedo_scaling_lut"] = torch.from_numpy(albedo).to(ctx.device)  # F32
    dev["preintegrated_charlie"] = torch.from_numpy(np.repeat(charlie[..., None], 4, -1).copy()).to(ctx.device)  # F32X4, r used
    rad, spec = api.pbr_shade_layers(ctx, g, dev, PERMUTATIONS[perm], f["camera"], sa, ibl_to_device(ibl_np, ctx.device), background=BACKGROUND, iridescence_ior=IOR,
                                     anisotropy_rotation=ROTATION)
    wr, ws = checker_result(lib, perm, optional, f, gn, sa, planes, albedo, charlie, ibl_np)
    got, gots = to_np(rad), to_np(spec)
    assert np.isfinite(got).all() and np.isfinite(wr).all()
    assert_close(got, wr, max_outlier_frac=0.0, what=f"radiance, layers {perm}")
    assert_close(gots, ws, max_outlier_frac=0.0, what=f"specular IBL, layers {perm}")
    # the layer changes the picture (it is not the default permutation that is being compared)
    base, _ = api.pbr_shade(ctx, g, f["camera"], sa, ibl_to_device(ibl_np, ctx.device), background=BACKGROUND)
    geom = gn["depth"] < 1.0 - 1e-6
    assert (np.abs(got - to_np(base))[geom] > 1e-3).mean() > 0.2
    ctx.close()


@pytest.mark.parametrize("perm,flags,pcf,size,optional", SHADOW_CASES)
def test_pbr_shade_layers_with_shadows(mifx_lib, ibl_np, perm, flags, pcf, size, optional):
    """ENABLE_SHADOWS on top of the layers (the `shadows` argument of mifx_pbr_shade_execute_layers): two of the lights are attenuated by FilterShadowMapFixedPCF."""
    import chain_util
    from diligentfx_amd import api

    lib = ref_checker()
    ctx = api.PostFXContext(0)
    f, gn, sa, planes, albedo, charlie = make_case(perm, size, ibl_np, ctx.device, shadowed=True)
    slices, infos = chain_util.make_shadow_inputs()
    g = {k: torch.from_numpy(v).to(ctx.device) for k, v in gn.items()}
    dev = {k: torch.from_numpy(v).to(ctx.device) for k, v in planes.items()}
    dev["transmission"] = dev["transmission"][..., 0].contiguous()
    if not optional:
        dev.pop("clearcoat_normal")
        dev.pop("tangent")
    dev["sheen_albedo_scaling_lut"] = torch.from_numpy(albedo).to(ctx.device)
    dev["preintegrated_charlie"] = torch.from_numpy(charlie).to(ctx.device)
    sm = torch.from_numpy(np.stack(slices)).to(ctx.device)
    ibl = ibl_to_device(ibl_np, ctx.device)
    kw = dict(background=BACKGROUND, iridescence_ior=IOR, anisotropy_rotation=ROTATION)
    rad, spec = api.pbr_shade_layers(ctx, g, dev, flags, f["camera"], sa, ibl, shadows=(sm, infos, pcf), **kw)
    wr, ws = checker_result(lib, perm, optional, f, gn, sa, planes, albedo, charlie, ibl_np, shadows=(slices, infos))
    # "reference < texel" on computed light-space depths: a tap exactly on the threshold may flip
    assert_close(to_np(rad), wr, max_outlier_frac=0.0, what=f"radiance, layers {perm}")
    assert_close(to_np(spec), ws, max_outlier_frac=0.0, what=f"specular IBL, layers {perm}")
    from diligentfx_amd import binding as B

    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        api.pbr_shade_layers(ctx, g, dev, flags, f["camera"], sa, ibl, **kw)  # a light with a shadow-map index and no shadow map
    # no layer + shadows = the shadowed default kernel
    want, _ = api.pbr_shade(ctx, g, f["camera"], sa, ibl, background=BACKGROUND, shadows=(sm, infos, pcf))
    got, _ = api.pbr_shade_layers(ctx, g, {}, 0, f["camera"], sa, ibl, shadows=(sm, infos, pcf), **kw)
    assert torch.equal(got, want)
    ctx.close()


def test_pbr_shade_layers_protocol(mifx_lib, ibl_np):
    """No layer = the default kernel, bit for bit; unknown flag bits and a missing plane are refused."""
    import chain_util
    from diligentfx_amd import api, binding as B, synth

    w, h = 96, 64
    ctx = api.PostFXContext(0)
    f = synth.make_frame(synth.Scene(), 4, w, h, ctx.device)
    g = {k: f[k] for k in ("base_color", "normal", "material", "depth")}
    sa = chain_util.shade_attribs(len(ibl_np["prefiltered"]) - 1)
    ibl = ibl_to_device(ibl_np, ctx.device)
    want, wants = api.pbr_shade(ctx, g, f["camera"], sa, ibl)
    got, gots = api.pbr_shade_layers(ctx, g, {}, 0, f["camera"], sa, ibl)
    assert torch.equal(got, want) and torch.equal(gots, wants)
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        api.pbr_shade_layers(ctx, g, {}, 32, f["camera"], sa, ibl)
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        api.pbr_shade_layers(ctx, g, {}, B.PBR_LAYER_CLEAR_COAT, f["camera"], sa, ibl)  # no clear-coat plane
    planes, albedo, charlie = make_layers(to_np(f["normal"]), seed=1)
    sheen = {"sheen": torch.from_numpy(planes["sheen"]).to(ctx.device)}
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        api.pbr_shade_layers(ctx, g, sheen, B.PBR_LAYER_SHEEN, f["camera"], sa, ibl)  # sheen without its two tables
    wrong = {"transmission": torch.zeros(h, w + 1, device=ctx.device)}
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        api.pbr_shade_layers(ctx, g, wrong, B.PBR_LAYER_TRANSMISSION, f["camera"], sa, ibl)
    ctx.close()
