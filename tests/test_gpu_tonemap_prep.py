"""GPU parity: HIP kernels (through the C ABI) vs the oracle, same seeded inputs."""
import numpy as np
import pytest
import torch

from util import assert_close, blue_noise_tables, tone_mapping_attribs_bytes, to_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(mifx_lib):
    from diligentfx_amd import api

    sobol, tile = blue_noise_tables()
    c = api.PostFXContext(0, sobol, tile)
    yield c
    c.close()


def checkers(oracle):
    """(name prefix, lib) pairs: the hand-written oracle always, the compiled reference (oracle/_ref: the reference's own shader source) whenever it travelled --
    it does on the GPU box, so M2 / C1 / C2 / C3 are held to `ref_` directly like every other suite, not through the port."""
    import pyref

    libs = [("oracle_", oracle)]
    r = pyref.ref_lib()
    if r is not None:
        libs.append(("ref_", r))
    return libs


@pytest.mark.parametrize("mode", range(0, 12))
@pytest.mark.parametrize("srgb", [0, 1])
def test_tonemap_all_modes(ctx, oracle, mode, srgb):
    from diligentfx_amd import binding as B, synth

    hdr = synth.make_hdr_buffer(200, 120, ctx.device)  # ragged vs the 64x4 block
    hdr[1, 0, :3] = torch.tensor([-1.0, 0.5, 2.0])
    attr = B.ToneMappingAttribs.default(mode)
    attr.AgXSaturation, attr.AgXSlope, attr.AgXPower, attr.AgXOffset = 1.1, 0.95, 1.05, 0.01
    got = to_np(ctx.tone_map(hdr, attr, 0.3, flags=srgb))
    for prefix, lib in checkers(oracle):
        want = np.zeros_like(got)
        lib.call(prefix + "tonemap", [to_np(hdr)], [want], attribs=bytes(attr), fval=[0.3], ival=[srgb])
        assert_close(got, want, what=f"tonemap mode {mode} vs {prefix}")


def test_tonemap_pitched_and_errors(ctx):
    from diligentfx_amd import binding as B

    big = torch.rand(33, 80, 4, device=ctx.device) * 4
    view = big[:, 7:71, :]  # pitched sub-image (row pitch > width)
    attr = B.ToneMappingAttribs.default(4)
    out = torch.zeros(33, 64, 4, device=ctx.device)
    ctx.tone_map(view, attr, 0.3, out=out)
    ref = ctx.tone_map(view.contiguous(), attr, 0.3)
    assert torch.equal(out, ref)
    bad = B.ToneMappingAttribs.default(12)
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        ctx.tone_map(big, bad, 0.3)
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        ctx.tone_map(big, attr, 0.3, out=torch.zeros(5, 5, 4, device=ctx.device))


def test_tonemap_full_size_properties(ctx):
    """BASELINE config 1 size (1920x1080): pointwise => permutation equivariance; alpha passes through."""
    from diligentfx_amd import binding as B, synth

    hdr = synth.make_hdr_buffer(1920, 1080, ctx.device)
    attr = B.ToneMappingAttribs.default(4)
    out = ctx.tone_map(hdr, attr, 0.3)
    assert torch.equal(out[..., 3], hdr[..., 3])
    flipped = ctx.tone_map(hdr.flip(0).flip(1).contiguous(), attr, 0.3)
    assert torch.equal(flipped.flip(0).flip(1), out)
    assert torch.isfinite(out).all() and (out[..., :3] >= 0).all()


@pytest.mark.parametrize("frame", [0, 1, 17, 300, 255, 256, 1023, 1024, 65535, 65536])
def test_blue_noise_bit_exact(ctx, oracle, frame):
    sobol, tile = blue_noise_tables()
    ctx.prepare_resources(frame, 64, 48)
    z = torch.ones(48, 64, device=ctx.device)
    from diligentfx_amd import synth

    cam = synth.make_camera(frame, 64, 48)
    ctx.execute(z, z, torch.zeros(48, 64, 2, device=ctx.device), cam, cam)
    xy, zw = to_np(ctx.get_2d_blue_noise(0)), to_np(ctx.get_2d_blue_noise(1))
    for prefix, lib in checkers(oracle):  # (the reference's own ComputeBlueNoiseTexture.fx compiled for the CPU when oracle/_ref travelled, and the hand port)
        wxy, wzw = np.zeros((128, 128, 2), np.float32), np.zeros((128, 128, 2), np.float32)
        lib.call(prefix + "blue_noise", [sobol.astype(np.float32).reshape(1, 256), tile.astype(np.float32).reshape(256, 512)], [wxy, wzw], ival=[frame])
        assert np.array_equal(xy, wxy) and np.array_equal(zw, wzw), prefix


@pytest.mark.parametrize("size", [(96, 64), (130, 70)])
def test_prep_passes(ctx, oracle, size):
    from diligentfx_amd import synth

    w, h = size
    f = synth.make_frame(synth.Scene(), 3, w, h, ctx.device)
    ctx.prepare_resources(3, w, h)
    ctx.execute(f["depth"], f["prev_depth"], f["motion"], f["camera"], f["prev_camera"])
    got_rd, got_cm = to_np(ctx.get_reprojected_depth()), to_np(ctx.get_closest_motion_vectors())
    assert ctx.get_previous_depth().data_ptr() == f["prev_depth"].data_ptr()  # C4 is an alias, not a copy
    depth, motion = to_np(f["depth"]), to_np(f["motion"])
    for prefix, lib in checkers(oracle):
        want = np.zeros_like(depth)
        lib.call(prefix + "reprojected_depth", [depth], [want], cam0=bytes(f["camera"]), cam1=bytes(f["prev_camera"]))
        assert_close(got_rd, want, what="reprojected depth " + prefix)
        want = np.zeros_like(motion)
        lib.call(prefix + "closest_motion", [depth, motion], [want])
        assert np.array_equal(got_cm, want)


def test_execute_before_prepare_is_an_error(mifx_lib):
    from diligentfx_amd import api, binding as B, synth

    c = api.PostFXContext(0)
    z = torch.ones(8, 8, device=c.device)
    cam = synth.make_camera(0, 8, 8)
    with pytest.raises(B.MifxError, match="INVALID_OP"):
        c.execute(z, z, torch.zeros(8, 8, 2, device=c.device), cam, cam)
    c.prepare_resources(0, 8, 8)
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        c.execute(z[:4], z, torch.zeros(8, 8, 2, device=c.device), cam, cam)
    with pytest.raises(B.MifxError, match="INVALID_OP"):
        c.get_2d_blue_noise(0)  # no tables supplied
    c.close()


def test_postfx_texture_helpers(ctx):
    """PostFXContext::ClearRenderTarget / CopyTextureDepth / CopyTextureColor / GetSupportedFeatures (PostFXContext.hpp:114, 168-172) on pitched planes."""
    from diligentfx_amd import binding as B

    assert ctx.get_supported_features() == {"TransitionSubresources": True, "TextureSubresourceViews": True, "CopyDepthToColor": True, "ShaderBaseVertexOffset": True}
    big = torch.zeros(40, 100, 4, device=ctx.device)
    view = big[:, 10:74, :]  # a 64-texel-wide window of wider rows: only the window may change
    ctx.clear_render_target(view, (0.25, -1.0, 3.0, 0.5))
    assert torch.equal(view, torch.tensor([0.25, -1.0, 3.0, 0.5], device=ctx.device).expand(40, 64, 4)) and float(big[:, :10].abs().sum()) == 0.0 and float(big[:, 74:].abs().sum()) == 0.0
    one = torch.zeros(33, 70, device=ctx.device)
    ctx.clear_render_target(one, (1.0, 0.0, 0.0, 0.0))  # (an R8 / R16 history target cleared to 1.0: ScreenSpaceAmbientOcclusion.cpp:304-321)
    assert torch.equal(one, torch.ones_like(one))
    src = torch.rand(33, 70, device=ctx.device)
    dst = torch.zeros(33, 90, device=ctx.device)[:, 5:75]
    ctx.copy_texture_depth(src, dst)
    assert torch.equal(dst, src)
    csrc, cdst = torch.rand(20, 31, 4, device=ctx.device), torch.zeros(20, 31, 4, device=ctx.device)
    ctx.copy_texture_color(csrc, cdst)
    assert torch.equal(cdst, csrc)
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        ctx.copy_texture_color(csrc, torch.zeros(20, 30, 4, device=ctx.device))  # another size: the reference's draw would resample; every caller copies 1:1
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        ctx.copy_texture_depth(csrc, torch.zeros(20, 31, device=ctx.device))  # formats differ
