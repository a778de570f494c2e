"""Non-default attribute blocks through the C ABI: the SSAO / SSR / Bloom effects on the GPU against the CPU chain with the attribute sets of
tests/test_oracle_attribs_vs_ref.py (where the checker itself is pinned to the reference build on them).

Every case decides the suite (the round-1 "first hardware run pending" markers are gone: all of them have run on an MI355X)."""
import numpy as np
import pytest
import torch

import cpu_chain
from util import assert_close, blue_noise_tables, centre_tap_slack, to_np

pytestmark = pytest.mark.gpu

W, H, FRAMES = 176, 104, 3


def checker():
    import pyref

    r = pyref.ref_lib()
    return (r, "ref_") if r is not None else (pyref.oracle_lib(), "oracle_")


def scene_color(f):
    c = f["base_color"][..., :3] * (0.6 + 1.5 * f["normal"][..., 1:2].clamp(0, 1)) + 0.05
    return torch.cat([c * f["base_color"][..., 3:4], f["base_color"][..., 3:4]], -1).contiguous()


def drive(make_effect, algorithm="gtao"):
    """Yields (frame index, frame tensors, effect, CPU chain, CPU postfx outputs) with the PostFX context prepared and executed for the frame."""
    from diligentfx_amd import api, synth

    lib, pfx = checker()
    sobol, tile = blue_noise_tables()
    ctx = api.PostFXContext(0, sobol, tile)
    fx = make_effect(api, ctx)
    chain = cpu_chain.CpuChain(lib, pfx, algorithm=algorithm)
    scene = synth.Scene()
    for frame in range(FRAMES):
        f = synth.make_frame(scene, frame, W, H, ctx.device)
        ctx.prepare_resources(frame, W, H)
        fx.prepare_resources()
        ctx.execute(f["depth"], f["prev_depth"], f["motion"], f["camera"], f["prev_camera"])
        pf = chain.postfx(frame, to_np(f["depth"]), to_np(f["prev_depth"]), to_np(f["motion"]), bytes(f["camera"]), bytes(f["prev_camera"]), (sobol, tile))
        yield frame, f, fx, chain, pf
    fx.close()
    ctx.close()


@pytest.mark.parametrize("algo,radius,falloff,mult,mipoff,temporal,spatial,thick", [
    ("gtao", 0.4, 0.3, 1.0, 2.0, 0.5, 2.0, 0.5), ("gtao", 2.5, 0.9, 2.0, 4.5, 0.97, 6.0, 0.5), ("hbao", 1.7, 0.615, 1.2, 3.3, 0.8, 4.0, 0.5), ("vbao", 1.3, 0.615, 1.457, 2.5, 0.9, 3.0, 0.15)])
def test_ssao_attribute_sweep_gpu(mifx_lib, algo, radius, falloff, mult, mipoff, temporal, spatial, thick):
    from diligentfx_amd import binding as B

    a = B.SSAOAttribs.default()
    a.EffectRadius, a.EffectFalloffRange, a.RadiusMultiplier, a.DepthMIPSamplingOffset = radius, falloff, mult, mipoff
    a.TemporalStabilityFactor, a.SpatialReconstructionRadius, a.BitmaskThickness = temporal, spatial, thick
    a.Algorithm = {"gtao": 0, "hbao": 1, "vbao": 2}[algo]
    a.AlphaInterpolation = 0.7
    for frame, f, ssao, chain, pf in drive(lambda api, ctx: api.ScreenSpaceAmbientOcclusion(ctx), algorithm=algo):
        ssao.execute(f["depth"], f["normal"], a)
        want = chain.ssao(pf, to_np(f["depth"]), to_np(f["normal"]), a)
        assert_close(to_np(ssao.get_ambient_occlusion()), want, max_outlier_frac=0.0, what=f"SSAO {algo} frame {frame}")  # (measured 0 on the no-contraction build, round 4)


@pytest.mark.parametrize("thick,thresh,mdm,perceptual,channel,trav,bias,radius,trad,tvar,sigma", [
    (0.05, 0.35, 0, 1, 0, 64, 0.0, 2.0, 0.7, 0.5, 0.5), (0.01, 0.5, 1, 1, 1, 128, 0.6, 6.0, 0.95, 0.9, 1.4), (0.025, 0.15, 2, 0, 2, 24, 0.3, 4.0, 1.0, 0.9, 0.9)])
def test_ssr_attribute_sweep_gpu(mifx_lib, thick, thresh, mdm, perceptual, channel, trav, bias, radius, trad, tvar, sigma):
    from diligentfx_amd import binding as B

    a = B.SSRAttribs.default()
    a.DepthBufferThickness, a.RoughnessThreshold, a.MostDetailedMip, a.IsRoughnessPerceptual, a.RoughnessChannel = thick, thresh, mdm, perceptual, channel
    a.MaxTraversalIntersections, a.GGXImportanceSampleBias, a.SpatialReconstructionRadius = trav, bias, radius
    a.TemporalRadianceStabilityFactor, a.TemporalVarianceStabilityFactor, a.BilateralCleanupSpatialSigmaFactor = trad, tvar, sigma
    a.AlphaInterpolation = 0.6
    for frame, f, ssr, chain, pf in drive(lambda api, ctx: api.ScreenSpaceReflection(ctx)):
        material = f["material"]
        if channel:  # roughness also in the other channels (squared in blue), as in the CPU sweep
            material = torch.stack([material[..., 1], material[..., 1], material[..., 0] ** 2, material[..., 3]], -1).contiguous()
        color = scene_color(f)
        ssr.execute(color, f["depth"], f["normal"], material, f["motion"], a)
        want = chain.ssr(pf, to_np(color), to_np(f["depth"]), to_np(f["normal"]), to_np(material), to_np(f["motion"]), a)
        got = to_np(ssr.get_ssr_radiance())
        assert np.isfinite(got).all()
        # (budget = 2 x the worst case of the sweep measured on an MI355X, 5.8e-3: profiles/r03_parity_outliers_strict_vs_shipped.txt)
        assert_close(got, want, max_outlier_frac=5e-3, what=f"SSR frame {frame}")  # (end to end over several frames: flipped rays travel through the temporal filter; measured 2.43e-3)


@pytest.mark.parametrize("intensity,threshold,soft,radius,alpha", [(0.6, 0.2, 0.5, 0.4, 1.0), (0.05, 2.0, 0.0, 1.0, 0.4)])
def test_bloom_attribute_sweep_gpu(mifx_lib, intensity, threshold, soft, radius, alpha):
    from diligentfx_amd import binding as B

    a = B.BloomAttribs.default()
    a.Intensity, a.Threshold, a.SoftTreshold, a.Radius, a.AlphaInterpolation = intensity, threshold, soft, radius, alpha
    for frame, f, bloom, chain, pf in drive(lambda api, ctx: api.Bloom(ctx)):
        color = (scene_color(f) * 3.0).contiguous()
        bloom.execute(color, a)
        # zero outliers; the only allowance is the derived bound of the centre tap of the source colour (util.centre_tap_slack): with Radius = 1
        # (all 7 levels, down to 2x1 and 1x1), Intensity 0.05 and Alpha 0.4 the output is 98 % source colour, and round 1's one failing texel
        # (frame 2, (30, 110): 0.0194 beside a texel of 3.0) was exactly that weight noise of the checker -- every pyramid level agreed to 1e-6.
        keep = {}
        want = chain.bloom(to_np(color), a, keep)
        assert_close(to_np(bloom.get_bloom_texture()), want, what=f"Bloom frame {frame}", abs_slack=centre_tap_slack(to_np(color)))
        for i, d in enumerate(keep["bloom_down"]):
            assert_close(to_np(bloom.get_intermediate(f"down{i}")), d, what=f"Bloom frame {frame} down{i}")
        for i, u in enumerate(keep["bloom_up"]):
            assert_close(to_np(bloom.get_intermediate(f"up{i}")), u, what=f"Bloom frame {frame} up{i}", abs_slack=centre_tap_slack(keep["bloom_down"][i]))
