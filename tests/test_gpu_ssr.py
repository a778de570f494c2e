"""GPU parity of the SSR passes (R1..R7): per-pass isolation (each HIP pass against the checker fed with the HIP pass' own inputs)
over several frames, plus the end-to-end effect against the independently running CPU chain."""
import numpy as np
import pytest
import torch

import cpu_chain
from util import assert_close, blue_noise_tables, to_np

pytestmark = pytest.mark.gpu


def checker():
    import pyref

    r = pyref.ref_lib()
    if r is not None:
        return r, "ref_"
    o = pyref.oracle_lib()
    if not o.has("oracle_ssr_intersection"):
        pytest.skip("no checker available for SSR")
    return o, "oracle_"


def scene_color(f):
    """A plausible un-composited scene radiance (bright spheres, dim plane) so that reflections carry signal."""
    c = f["base_color"][..., :3] * (0.6 + 1.5 * f["normal"][..., 1:2].clamp(0, 1)) + 0.05
    return torch.cat([c * f["base_color"][..., 3:4], f["base_color"][..., 3:4]], -1).contiguous()


# flags 1: FEATURE_FLAG_PREVIOUS_FRAME; rev: PostFXContext::FEATURE_FLAG_REVERSED_DEPTH (SSR_OPTION_INVERTED_DEPTH)
@pytest.mark.parametrize("size,mdm,flags,rev", [((192, 112), 0, 0, False), ((150, 85), 1, 0, False), ((192, 112), 0, 1, False), ((176, 100), 0, 0, True)])
def test_ssr_per_pass_parity(mifx_lib, size, mdm, flags, rev):
    from diligentfx_amd import api, binding as B, synth

    lib, pfx = checker()
    cc = cpu_chain.CpuChain(lib, pfx, reversed_depth=rev)
    w, h = size
    sobol, tile = blue_noise_tables()
    ctx = api.PostFXContext(0, sobol, tile)
    ssr = api.ScreenSpaceReflection(ctx)
    scene = synth.Scene()
    attribs = B.SSRAttribs.default()
    attribs.MostDetailedMip = mdm
    ab = bytes(attribs)
    prev_rad, prev_var = np.zeros((h, w, 4), np.float32), np.zeros((h, w), np.float32)
    # R5's targets and the history slots are written under the reflection mask only: elsewhere they keep their previous content (ssr.hip), which the checker's passes are
    # handed as the initial content of their outputs -- zero like the product's planes when created
    res_before = [np.zeros((h, w, 4), np.float32), np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)]
    slot_before = {0: (np.zeros((h, w, 4), np.float32), np.zeros((h, w), np.float32)), 1: (np.zeros((h, w, 4), np.float32), np.zeros((h, w), np.float32))}
    worst = {}
    for frame in range(3):
        f = synth.make_frame(scene, frame, w, h, ctx.device, reversed_depth=rev)
        color = scene_color(f)
        ctx.prepare_resources(frame, w, h, feature_flags=1 if rev else 0)
        ssr.prepare_resources(feature_flags=flags)
        ctx.execute(f["depth"], f["prev_depth"], f["motion"], f["camera"], f["prev_camera"])
        ssr.execute(color, f["depth"], f["normal"], f["material"], f["motion"], attribs)
        cam, prev = bytes(f["camera"]), bytes(f["prev_camera"])
        depth, normal, material, motion = (to_np(f[k]) for k in ("depth", "normal", "material", "motion"))
        g = lambda n: to_np(ssr.get_intermediate(n))  # noqa: E731

        def cmp(name, got, want, frac=0.0, **kw):
            e, fr = assert_close(got, want, max_outlier_frac=frac, what=f"frame {frame} {name}", **kw)
            worst[name] = max(worst.get(name, 0.0), fr)

        # R1: bit exact (min)
        hiz = [depth] + [g(f"hiz{k}") for k in range(1, 7)]
        for k in range(1, 7):
            want = np.zeros_like(hiz[k])
            cc.call("ssr_hiz_mip", [hiz[k - 1]], [want], ival=[k - 1])
            assert np.array_equal(hiz[k], want), f"hiz{k}"
        # R2
        wr, wm = np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)
        cc.call("ssr_mask_roughness", [material, depth], [wr, wm], attribs=ab)
        rough, mask = g("roughness"), g("mask")
        assert np.array_equal(rough, wr) and np.array_equal(mask, wm)
        assert 0.05 < mask.mean() < 0.95
        # R4: a data-dependent ray march -- single-ulp differences can change a tile-crossing decision and the ray then lands on another
        # texel; such rays are rare and show up as outliers of the per-pixel comparison
        ws, wd = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
        r4_in = [to_np(color), normal, rough, to_np(ctx.get_2d_blue_noise(0)), hiz, mask, motion]
        if pfx == "ref_":
            cc.call("ssr_intersection_prev" if flags & 1 else "ssr_intersection", r4_in, [ws, wd], cam0=cam, attribs=ab)
        else:
            cc.call("ssr_intersection", r4_in, [ws, wd], cam0=cam, attribs=ab, ival=[flags & 1])
        if flags & 1:  # the variant really reads another texel for moving hits
            w0s, w0d = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
            lib.call(pfx + "ssr_intersection", r4_in, [w0s, w0d], cam0=cam, attribs=ab, **({} if pfx == "ref_" else {"ival": [0]}))
            assert np.array_equal(w0d, wd) and (frame == 0 or not np.array_equal(w0s, ws))
        cmp("R4 specular", g("ray_radiance"), ws)
        cmp("R4 dir/pdf", g("ray_dir_pdf"), wd)
        assert (g("ray_radiance")[..., 3] > 0).mean() > 0.01  # some rays hit
        # R5
        w0, w1, w2 = (a.copy() for a in res_before)
        cc.call("ssr_spatial_reconstruction", [rough, normal, depth, g("ray_dir_pdf"), g("ray_radiance"), mask], [w0, w1, w2], cam0=cam, attribs=ab)
        cmp("R5 radiance", g("res_radiance"), w0)
        cmp("R5 variance", g("res_variance"), w1, atol=1e-6)
        cmp("R5 depth", g("res_depth"), w2)
        res_before = [g("res_radiance").copy(), g("res_variance").copy(), g("res_depth").copy()]
        # R6
        w0, w1 = (a.copy() for a in slot_before[frame & 1])
        cc.call("ssr_temporal_accumulation", [motion, g("res_depth"), to_np(ctx.get_reprojected_depth()), g("res_radiance"), g("res_variance"),
                                                     to_np(f["prev_depth"]), prev_rad, prev_var, mask], [w0, w1], cam0=cam, cam1=prev, attribs=ab)
        cmp("R6 radiance", g("hist_radiance"), w0, frac=5e-5)  # (history rejection thresholds: measured 1.16e-5 = one value of 86 016)
        cmp("R6 variance", g("hist_variance"), w1, frac=1e-4, atol=1e-6)  # (measured 0; the same thresholds as the radiance)
        # R7
        want = np.zeros((h, w, 4), np.float32)
        cc.call("ssr_bilateral_cleanup", [depth, normal, rough, g("hist_radiance"), g("hist_variance"), mask], [want], cam0=cam, attribs=ab)
        out = to_np(ssr.get_ssr_radiance())
        cmp("R7", out, want)
        assert (out[mask == 0] == 0).all()
        prev_rad, prev_var = g("hist_radiance").copy(), g("hist_variance").copy()
        slot_before[frame & 1] = (prev_rad, prev_var)
    print("worst outlier fractions:", {k: round(v, 5) for k, v in worst.items() if v > 0})
    ssr.close()
    ctx.close()


def test_ssr_end_to_end_vs_cpu_chain(mifx_lib):
    from diligentfx_amd import api, binding as B, synth

    lib, pfx = checker()
    w, h = 208, 120
    sobol, tile = blue_noise_tables()
    ctx = api.PostFXContext(0, sobol, tile)
    ssr = api.ScreenSpaceReflection(ctx)
    chain = cpu_chain.CpuChain(lib, pfx)
    scene = synth.Scene()
    attribs = B.SSRAttribs.default()
    for frame in range(4):
        f = synth.make_frame(scene, frame, w, h, ctx.device)
        color = scene_color(f)
        ctx.prepare_resources(frame, w, h)
        ssr.prepare_resources()
        ctx.execute(f["depth"], f["prev_depth"], f["motion"], f["camera"], f["prev_camera"])
        ssr.execute(color, f["depth"], f["normal"], f["material"], f["motion"], attribs)
        pf = chain.postfx(frame, to_np(f["depth"]), to_np(f["prev_depth"]), to_np(f["motion"]), bytes(f["camera"]), bytes(f["prev_camera"]), (sobol, tile))
        want = chain.ssr(pf, to_np(color), to_np(f["depth"]), to_np(f["normal"]), to_np(f["material"]), to_np(f["motion"]), attribs)
        got = to_np(ssr.get_ssr_radiance())
        # flipped rays propagate through the 8-tap reconstruction and the history.  Budget = 2 x the fraction measured on an MI355X (worst frame 3.45e-3; the strict
        # build -- exact divisions, no contraction -- has 2.25e-3: profiles/r03_parity_outliers_strict_vs_shipped.txt)
        assert_close(got, want, max_outlier_frac=5e-3, what=f"SSR output frame {frame}")  # (the effect end to end; measured 2.34e-3)
        assert np.isfinite(got).all()
    ssr.close()
    ctx.close()


def test_ssr_protocol_errors(mifx_lib):
    from diligentfx_amd import api, binding as B

    sobol, tile = blue_noise_tables()
    ctx = api.PostFXContext(0, sobol, tile)
    ssr = api.ScreenSpaceReflection(ctx)
    ctx.prepare_resources(0, 64, 48)
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        ssr.prepare_resources(feature_flags=8)
    ssr.prepare_resources()
    z4, z1, z2 = torch.zeros(48, 64, 4, device=ctx.device), torch.ones(48, 64, device=ctx.device), torch.zeros(48, 64, 2, device=ctx.device)
    with pytest.raises(B.MifxError, match="INVALID_OP"):
        ssr.execute(z4, z1, z4, z4, z2, B.SSRAttribs.default())
    ssr.close()
    ctx.close()


@pytest.mark.parametrize("size", [(192, 112), (151, 89)])
def test_ssr_half_resolution(mifx_lib, size):
    """FEATURE_FLAG_HALF_RESOLUTION: R3 half-size mask (bit-exact), R4 at half size (one pixel of every 2x2 block, ComputeHalfResolutionOffset), R5 on the half-size ray
    textures; every changed pass against the checker on the HIP path's own inputs, the effect against the checker's own run."""
    from diligentfx_amd import api, binding as B, synth

    lib, pfx = checker()
    cc, e2e = cpu_chain.CpuChain(lib, pfx), cpu_chain.CpuChain(lib, pfx)
    w, h = size
    hw, hh = w // 2, h // 2
    sobol, tile = blue_noise_tables()
    ctx = api.PostFXContext(0, sobol, tile)
    ssr = api.ScreenSpaceReflection(ctx)
    scene = synth.Scene()
    attribs = B.SSRAttribs.default()
    ab = bytes(attribs)
    res_before = [np.zeros((h, w, 4), np.float32), np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)]  # (R5's targets keep their content outside the mask: ssr.hip)
    for frame in range(3):
        f = synth.make_frame(scene, frame, w, h, ctx.device)
        color = scene_color(f)
        ctx.prepare_resources(frame, w, h)
        ssr.prepare_resources(feature_flags=2)
        ctx.execute(f["depth"], f["prev_depth"], f["motion"], f["camera"], f["prev_camera"])
        ssr.execute(color, f["depth"], f["normal"], f["material"], f["motion"], attribs)
        cam, prev = bytes(f["camera"]), bytes(f["prev_camera"])
        depth, normal, material, motion = (to_np(f[k]) for k in ("depth", "normal", "material", "motion"))
        g = lambda n: to_np(ssr.get_intermediate(n))  # noqa: E731
        rough, mask = g("roughness"), g("mask")
        hiz = [depth] + [g(f"hiz{k}") for k in range(1, 7)]
        # R3
        want = np.zeros((hh, hw), np.float32)
        cc.call("ssr_downsampled_mask", [rough, depth], [want], attribs=ab)
        half_mask = g("mask_half")
        assert np.array_equal(half_mask, want) and 0.05 < half_mask.mean() < 0.95
        # R4 at half size
        ws, wd = np.zeros((hh, hw, 4), np.float32), np.zeros((hh, hw, 4), np.float32)
        r4_in = [to_np(color), normal, rough, to_np(ctx.get_2d_blue_noise(0)), hiz, half_mask, motion]
        if pfx == "ref_":
            cc.call("ssr_intersection_half", r4_in, [ws, wd], cam0=cam, attribs=ab)
        else:
            cc.call("ssr_intersection", r4_in, [ws, wd], cam0=cam, attribs=ab, ival=[0, 0, 0, 0, 0, 0, 1])
        assert g("ray_radiance").shape == (hh, hw, 4)
        assert_close(g("ray_radiance"), ws, max_outlier_frac=0.0, what=f"half-res R4 specular frame {frame}")
        assert_close(g("ray_dir_pdf"), wd, max_outlier_frac=0.0, what=f"half-res R4 dir/pdf frame {frame}")
        assert (g("ray_radiance")[..., 3] > 0).mean() > 0.01
        # R5 on the half-size ray textures
        w0, w1, w2 = (a.copy() for a in res_before)
        r5_in = [rough, normal, depth, g("ray_dir_pdf"), g("ray_radiance"), mask]
        if pfx == "ref_":
            cc.call("ssr_spatial_reconstruction_half", r5_in, [w0, w1, w2], cam0=cam, attribs=ab)
        else:
            cc.call("ssr_spatial_reconstruction", r5_in, [w0, w1, w2], cam0=cam, attribs=ab, ival=[0, 0, 0, 0, 0, 0, 1])
        assert_close(g("res_radiance"), w0, max_outlier_frac=0.0, what=f"half-res R5 radiance frame {frame}")
        assert_close(g("res_variance"), w1, max_outlier_frac=0.0, atol=1e-6, what=f"half-res R5 variance frame {frame}")
        assert_close(g("res_depth"), w2, max_outlier_frac=0.0, what=f"half-res R5 depth frame {frame}")
        res_before = [g("res_radiance").copy(), g("res_variance").copy(), g("res_depth").copy()]
        # end to end (stochastic + temporal stages run independently on both sides)
        pf = e2e.postfx(frame, depth, to_np(f["prev_depth"]), motion, cam, prev, (sobol, tile))
        want = e2e.ssr(pf, to_np(color), depth, normal, material, motion, attribs, half_resolution=True)
        out = to_np(ssr.get_ssr_radiance())
        assert out.shape == (h, w, 4) and np.isfinite(out).all()
        # budget = 2.4 x the measured worst frame (2.49e-3; strict build 1.69e-3: profiles/r03_parity_outliers_strict_vs_shipped.txt)
        assert_close(out, want, max_outlier_frac=3.7e-3, what=f"half-res SSR end to end frame {frame}")  # (measured 1.84e-3)
    ssr.close()
    ctx.close()
