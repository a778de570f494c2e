"""GPU parity of the depth-of-field effect (PostProcess/DepthOfField): every pass through the C ABI against the checker fed with the
HIP path's own inputs (per-pass isolation), the whole effect against the checker's independent run, properties at 3840x2160, the chain hook."""
import numpy as np
import pytest
import torch

import cpu_chain
from util import assert_close, blue_noise_tables, to_np

pytestmark = pytest.mark.gpu

LENS = (12.0, 1.2, 135.0)  # focus distance (m), f-stop, focal length (mm): blurs both fields of the synthetic scene


def checker():
    import pyref

    r = pyref.ref_lib()
    if r is not None:
        return r, "ref_"
    return pyref.oracle_lib(), "oracle_"


def hdr_colour(f, device):
    """Scene colour with highlights (Karis weights, flood fill) and a non-trivial alpha, derived from the G-buffer so that it moves with the camera."""
    base, n = f["base_color"], f["normal"]
    rgb = base[..., :3] * 1.5 + 0.1 * n[..., :3].abs()
    rgb = rgb * torch.exp2(6.0 * (base[..., 0:1] - 0.5).clamp(min=0.0))
    return torch.cat([rgb, base[..., 3:4] * 0.5 + 0.25], -1).contiguous().to(device)


def lens_camera(cam, lens=LENS):
    cam.fFocusDistance, cam.fFStop, cam.fFocalLength = lens
    return cam


def f32(*shape):
    return np.zeros(shape, np.float32)


class Passes:
    """The checker's passes, one call each (the reference compiled for the CPU has separate entry points per shader permutation)."""

    def __init__(self, lib, pfx, cam, attribs, flags):
        self.lib, self.p, self.cam, self.ab, self.flags = lib, pfx, cam, bytes(attribs), flags

    def call(self, name, ins, outs, **kw):
        self.lib.call(self.p + name, ins, outs, **kw)
        return outs[0] if len(outs) == 1 else outs

    def coc(self, depth):
        return self.call("dof_coc", [depth], [f32(*depth.shape)], cam0=self.cam, attribs=self.ab)

    def temporal(self, coc, prev, motion):
        return self.call("dof_temporal_coc", [coc, prev, motion], [f32(*coc.shape)], cam0=self.cam, attribs=self.ab)

    def separated(self, coc):
        return self.call("dof_separated_coc", [coc], [f32(*coc.shape)])

    def dilation(self, last):
        return self.call("dof_dilation_coc", [last], [f32(last.shape[0] // 2, last.shape[1] // 2)])

    def blur(self, coc, gauss):
        if self.p == "ref_":
            x = self.call("dof_blur_x", [coc, gauss], [f32(*coc.shape)])
            return self.call("dof_blur_y", [x, gauss], [f32(*coc.shape)])
        x = self.call("dof_blur", [coc, gauss], [f32(*coc.shape)], ival=[0])
        return self.call("dof_blur", [x, gauss], [f32(*coc.shape)], ival=[1])

    def prefilter(self, color, coc, blurred):
        h, w = coc.shape
        return self.call("dof_prefilter", [color, coc, blurred], [f32(h // 2, w // 2, 4), f32(h // 2, w // 2, 4)], attribs=self.ab)

    def bokeh_first(self, near, far, kernel, color):
        karis = bool(self.flags & 2)
        outs = [f32(*near.shape), f32(*near.shape)]
        if self.p == "ref_":
            return self.call("dof_bokeh_first_karis" if karis else "dof_bokeh_first", [near, far, kernel, color], outs, cam0=self.cam, attribs=self.ab)
        return self.call("dof_bokeh_first", [near, far, kernel, color], outs, cam0=self.cam, attribs=self.ab, ival=[int(karis)])

    def bokeh_second(self, near, far, kernel):
        return self.call("dof_bokeh_second", [near, far, kernel], [f32(*near.shape), f32(*near.shape)], cam0=self.cam, attribs=self.ab)

    def postfilter(self, near, far):
        return self.call("dof_postfilter", [near, far], [f32(*near.shape), f32(*near.shape)])

    def combine(self, color, coc, near, far):
        return self.call("dof_combine", [color, coc, near, far], [f32(*color.shape)], cam0=self.cam, attribs=self.ab)


def read(dof, names):
    return {n: to_np(dof.get_intermediate(n)).copy() for n in names}


# 256x144: all three dilation levels in one launch; 100x60: two fused + one from an odd source; 202x118: odd first level (three-texel footprints)
@pytest.mark.parametrize("size,flags,rings", [((256, 144), 0, (5, 7)), ((256, 144), 3, (5, 7)), ((202, 118), 1, (3, 4)), ((100, 60), 2, (2, 2)), ((202, 118), 0, (4, 6))])
def test_dof_per_pass_and_output(mifx_lib, size, flags, rings):
    from diligentfx_amd import api, binding as B, synth

    lib, pfx = checker()
    w, h = size
    sobol, tile = blue_noise_tables()
    ctx = api.PostFXContext(0, sobol, tile)
    dof = api.DepthOfField(ctx)
    scene = synth.Scene()
    attribs = B.DOFAttribs.default()
    attribs.MaxCircleOfConfusion, attribs.AlphaInterpolation = 0.02, 0.9
    attribs.BokehKernelRingCount, attribs.BokehKernelRingDensity = rings
    temporal = bool(flags & 1)
    e2e_chain = cpu_chain.CpuChain(lib, pfx)
    prev_temporal = np.zeros((h, w), np.float32)
    worst = {}
    for frame in (7, 8, 9):
        f = synth.make_frame(scene, frame, w, h, ctx.device)
        cam = lens_camera(f["camera"])
        color = hdr_colour(f, ctx.device)
        ctx.prepare_resources(frame, w, h)
        dof.prepare_resources(flags)
        ctx.execute(f["depth"], f["prev_depth"], f["motion"], cam, f["prev_camera"])
        motion = to_np(ctx.get_closest_motion_vectors())
        dof.debug_set_last_pass(7)
        dof.execute(color, f["depth"], attribs)
        first = read(dof, ["coc", "dilation1", "dilation2", "dilation3", "dilation_blurred", "prefiltered0", "prefiltered1", "bokeh0", "bokeh1"]
                     + (["coc_temporal"] if temporal else []))
        dof.debug_set_last_pass(0)
        dof.execute(color, f["depth"], attribs)  # the same frame again, all passes: the history slot is rewritten with the same values
        torch.cuda.synchronize()
        second = read(dof, ["coc", "prefiltered0", "prefiltered1", "bokeh0", "bokeh1"] + (["coc_temporal"] if temporal else []))
        got = to_np(dof.get_depth_of_field_texture()).copy()
        assert np.array_equal(first["coc"], second["coc"]) and (not temporal or np.array_equal(first["coc_temporal"], second["coc_temporal"]))
        cnp, dnp = to_np(color), to_np(f["depth"])
        P = Passes(lib, pfx, bytes(cam), attribs, flags)
        large, small, gauss = e2e_chain.dof_tables(*rings)
        assert np.array_equal(api.DepthOfField.generate_kernel_points(*rings), large[0, :1 + rings[1] * (rings[0] - 1) * rings[0] // 2])

        def cmp(name, a, b, frac=0.0):
            worst[name] = max(worst.get(name, 0.0), assert_close(a, b, max_outlier_frac=frac, what=f"{name} frame {frame} {size} flags {flags}")[0])

        cmp("coc", first["coc"], P.coc(dnp))
        used = first["coc"]
        if temporal:
            # the inside-screen test on the reprojected position is a threshold => a pixel on the frame border may flip
            cmp("coc_temporal", first["coc_temporal"], P.temporal(first["coc"], prev_temporal, motion))
            used = prev_temporal = first["coc_temporal"]
        lvl = P.separated(used)
        for k in (1, 2, 3):
            want = P.dilation(lvl)
            assert np.array_equal(first[f"dilation{k}"], want), f"dilation{k} is a max over copies of the inputs: must be bit-exact"
            lvl = first[f"dilation{k}"]
        cmp("dilation_blurred", first["dilation_blurred"], P.blur(first["dilation3"], gauss))
        n6, f6 = P.prefilter(cnp, used, first["dilation_blurred"])
        cmp("prefiltered near", first["prefiltered0"], n6)
        cmp("prefiltered far", first["prefiltered1"], f6)
        # "a >= CoCFar" compares interpolated alphas: a tap whose alpha equals the centre's up to rounding may flip (1-ulp differences of the
        # texture coordinates); a flipped tap changes the pixel's average by 1 / taps
        n7, f7 = P.bokeh_first(first["prefiltered0"], first["prefiltered1"], large, cnp)
        cmp("bokeh gather near", first["bokeh0"], n7)
        cmp("bokeh gather far", first["bokeh1"], f7)
        n8, f8 = P.bokeh_second(first["bokeh0"], first["bokeh1"], small)
        cmp("bokeh fill near", second["prefiltered0"], n8)
        cmp("bokeh fill far", second["prefiltered1"], f8)
        n9, f9 = P.postfilter(second["prefiltered0"], second["prefiltered1"])
        cmp("postfilter near", second["bokeh0"], n9)
        cmp("postfilter far", second["bokeh1"], f9)
        cmp("combined", got, P.combine(cnp, used, second["bokeh0"], second["bokeh1"]))
        assert np.array_equal(got[..., 3], cnp[..., 3])
        # end to end: the checker's own run of the whole effect (its own history); flips of the far-field test propagate through fill + tent
        pf = {"frame": frame, "cam": bytes(cam), "closest_motion": motion}
        want = e2e_chain.dof(pf, cnp, dnp, attribs, flags)
        # (measured on an MI355X: not one value beyond rtol, profiles/r03_parity_outliers_strict_vs_shipped.txt)
        assert_close(got, want, max_outlier_frac=0.0, what=f"DOF end to end frame {frame}")
        assert np.isfinite(got).all() and np.abs(got[..., :3] - cnp[..., :3]).max() > 0.05
    print("max rel err per pass:", {k: f"{v:.1e}" for k, v in worst.items()})
    dof.close()
    ctx.close()


def test_dof_argument_checks(mifx_lib):
    from diligentfx_amd import api, binding as B, synth

    w, h = 64, 48
    ctx = api.PostFXContext(0)
    dof = api.DepthOfField(ctx)
    attribs = B.DOFAttribs.default()
    with pytest.raises(B.MifxError, match="INVALID_OP"):
        dof.prepare_resources(0)  # before the PostFX context
    ctx.prepare_resources(0, w, h)
    f = synth.make_frame(synth.Scene(), 0, w, h, ctx.device)
    color = hdr_colour(f, ctx.device)
    with pytest.raises(B.MifxError, match="INVALID_OP"):
        dof.execute(color, f["depth"], attribs)  # before prepare
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        dof.prepare_resources(8)  # unknown feature flag
    dof.prepare_resources(0)
    with pytest.raises(B.MifxError, match="INVALID_OP"):
        dof.execute(color, f["depth"], attribs)  # the camera comes from PostFXContext::Execute
    ctx.execute(f["depth"], f["prev_depth"], f["motion"], f["camera"], f["prev_camera"])
    attribs.BokehKernelRingCount, attribs.BokehKernelRingDensity = 6, 9  # 136 points > the 128 the reference's kernel texture holds
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        dof.execute(color, f["depth"], attribs)
    attribs = B.DOFAttribs.default()
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        dof.execute(color[:, : w // 2].contiguous(), f["depth"], attribs)  # size mismatch
    dof.execute(color, f["depth"], attribs)
    torch.cuda.synchronize()
    dof.close()
    ctx.close()


def test_dof_full_size_properties(mifx_lib):
    """3840x2160 (BASELINE.json configs[1]): size-independent properties of the effect."""
    from diligentfx_amd import api, binding as B, synth

    w, h = 3840, 2160
    ctx = api.PostFXContext(0)
    dof = api.DepthOfField(ctx)
    f = synth.make_frame(synth.Scene(), 4, w, h, ctx.device)
    cam = lens_camera(f["camera"])
    color = hdr_colour(f, ctx.device)
    ctx.prepare_resources(4, w, h)
    dof.prepare_resources(0)
    ctx.execute(f["depth"], f["prev_depth"], f["motion"], cam, f["prev_camera"])
    attribs = B.DOFAttribs.default()
    dof.execute(color, f["depth"], attribs)
    out = dof.get_depth_of_field_texture().clone()
    coc = dof.get_intermediate("coc").clone()
    # the CoC is the thin-lens formula: sign = side of the focus plane, range [-1, 1]
    P = synth.cam_mat(cam, "mProj", ctx.device)
    z = (P[3, 2] - f["depth"] * P[3, 3]) / (f["depth"] * P[2, 3] - P[2, 2])
    assert float(coc.min()) >= -1.0 and float(coc.max()) <= 1.0
    far, near = z > cam.fFocusDistance * 1.001, z < cam.fFocusDistance * 0.999
    assert bool((coc[far] > 0).all()) and bool((coc[near] < 0).all()) and bool(far.any()) and bool(near.any())
    # three dilation levels = the max of the near-field CoC over 8x8 blocks, bit for bit
    want = torch.nn.functional.max_pool2d((-coc).clamp(min=0.0)[None, None], 8)[0, 0]
    assert torch.equal(dof.get_intermediate("dilation3"), want)
    # every stage is an average or a max of colours and of the black of the empty half-resolution field (a texel outside the near / far
    # field carries rgb = 0 into the tent filter): the result never exceeds the colour range of the input
    hi = color[..., :3].amax(dim=(0, 1))
    assert bool((out[..., :3] >= 0.0).all()) and bool((out[..., :3] <= hi * (1 + 1e-5)).all())
    assert torch.equal(out[..., 3], color[..., 3])
    changed = (out[..., :3] - color[..., :3]).abs().amax(-1) > 1e-3
    assert 0.05 < float(changed.float().mean()) < 1.0  # blurred fields exist, and so does an in-focus band
    # a constant colour goes through unchanged wherever the blend reads a field that is not next to its own edge (all weights sum to one);
    # at the edges of the far field the reference blends towards the black of the empty field (dark fringes around foreground silhouettes)
    const = torch.empty_like(color)
    const[..., 0], const[..., 1], const[..., 2], const[..., 3] = 0.7, 2.5, 0.04, 1.0
    dof.execute(const, f["depth"], attribs)
    rel = ((dof.get_depth_of_field_texture() - const) / const)[..., :3]
    assert float(rel.max()) < 2e-5 and float(rel.min()) > -1.0
    assert float((rel.abs().amax(-1) > 2e-5).float().mean()) < 0.1
    # AlphaInterpolation = 0 returns the input exactly; a scene entirely in the focus plane is untouched as well
    attribs.AlphaInterpolation = 0.0
    dof.execute(color, f["depth"], attribs)
    assert torch.equal(dof.get_depth_of_field_texture(), color)
    attribs.AlphaInterpolation = 1.0
    zf = torch.full_like(f["depth"], float((P[2, 2] * cam.fFocusDistance + P[3, 2]) / (P[2, 3] * cam.fFocusDistance + P[3, 3])))
    dof.execute(color, zf, attribs)
    assert float(dof.get_intermediate("coc").abs().max()) < 0.05  # (depth <-> z round trip in fp32)
    assert torch.equal(dof.get_depth_of_field_texture(), color)  # smoothstep(0.1, 1, a) = 0 below 0.1
    dof.close()
    ctx.close()


def test_chain_with_depth_of_field(mifx_lib):
    """mifx_chain_set_depth_of_field: DOF runs on the TAA output and Bloom reads its result (HnPostProcessTask.cpp:899-918)."""
    import chain_util
    from diligentfx_amd import api, binding as B, synth

    W, H = 320, 192
    sobol, tile = blue_noise_tables()
    dev = torch.device("cuda", 0)
    scene = synth.Scene()
    a, b = api.Chain(0, sobol, tile), api.Chain(0, sobol, tile)
    ibl = api.precompute_ibl(a.postfx, synth.make_sky_cube(32, dev).clamp(max=200.0), lut_size=32, irradiance_size=8, prefiltered_size=32, lut_samples=32,
                             diffuse_samples=32, specular_samples=16)
    shade = chain_util.shade_attribs(len(ibl.pre) - 1)
    attribs = B.DOFAttribs.default()
    attribs.MaxCircleOfConfusion = 0.02
    flags = api.DepthOfField.FEATURE_FLAG_ENABLE_TEMPORAL_SMOOTHING
    a.set_depth_of_field(attribs, flags)
    with pytest.raises(ValueError):
        b.effect_output("dof")
    out_a, out_b = torch.zeros(H, W, 4, device=dev), torch.zeros(H, W, 4, device=dev)
    standalone = api.DepthOfField(b.postfx)
    for fi in range(3, 6):
        g = synth.make_frame(scene, fi, W, H, dev)
        lens_camera(g["camera"])
        a.execute(a.bind_frame(fi, g, ibl, shade, out_a))
        b.execute(b.bind_frame(fi, g, ibl, shade, out_b))
        torch.cuda.synchronize()
        taa_b = b.effect_output("taa")
        assert torch.equal(a.effect_output("taa"), taa_b)  # identical up to the TAA output
        # chain b's TAA output through a stand-alone effect object = what chain a's own effect produced, and what its Bloom consumed
        standalone.prepare_resources(flags)
        standalone.execute(taa_b, g["depth"], attribs)
        assert torch.equal(standalone.get_depth_of_field_texture(), a.effect_output("dof"))
        bloom = api.Bloom(b.postfx)
        bloom.prepare_resources()
        bloom.execute(standalone.get_depth_of_field_texture(), b.bloom_attribs)
        assert torch.equal(bloom.get_bloom_texture(), a.effect_output("bloom"))
        bloom.close()
        assert not torch.equal(out_a, out_b)
    a.set_row_band(0, H // 2, 8)  # (round 3: depth of field runs inside the row-band phases, tests/test_gpu_sharded.py)
    a.set_row_band(0, 0, 0)
    standalone.close()
    a.close()
    b.close()
