"""GPU parity of the auto-exposure pass (low-resolution weighted log-luminance, box mip chain on wave shuffles, UpdateAverageLuminance)
and of the tone-map entry that reads its result on the device."""
import numpy as np
import pytest
import torch

from util import assert_close, to_np

pytestmark = pytest.mark.gpu


def checker():
    import pyref

    r = pyref.ref_lib()
    if r is not None and r.has("ref_autoexposure"):
        return r, "ref_"
    o = pyref.oracle_lib()
    if not o.has("oracle_autoexposure"):
        pytest.skip("no checker available for the auto-exposure pass")
    return o, "oracle_"


@pytest.mark.parametrize("size", [(230, 150), (1920, 1080), (64, 64), (31, 17)])
def test_autoexposure_parity(mifx_lib, size):
    from diligentfx_amd import api, synth

    lib, pfx = checker()
    w, h = size
    ctx = api.PostFXContext(0)
    hdr = synth.make_hdr_buffer(w, h, ctx.device)
    ae = api.AutoExposure(ctx)
    low_w = np.zeros((64, 64, 2), np.float32)
    avg_w = np.full((1, 1), 0.1, np.float32)
    img = to_np(hdr)
    for dt, adapt in [(0.016, True), (0.4, True), (0.016, False), (1.5, True)]:
        ae.execute(hdr, dt, adapt)
        lib.call(pfx + "autoexposure", [img], [low_w, avg_w], fval=[dt], ival=[1 if adapt else 0])
        assert_close(to_np(ae.plane("low_res_luminance")), low_w, what=f"low-resolution luminance {size}")
        got = float(to_np(ae.plane("average_luminance"))[0, 0])
        assert got == pytest.approx(float(avg_w[0, 0]), rel=1e-3), (dt, adapt)
        assert ae.average() == pytest.approx(max(0.05, float(avg_w[0, 0])), rel=1e-3)
    ae.close()
    ctx.close()


def test_tonemap_with_device_luminance(mifx_lib):
    """mifx_tonemap_execute_auto == mifx_tonemap_execute with the host value of GetAverageSceneLuminance()."""
    from diligentfx_amd import api, binding as B, synth

    ctx = api.PostFXContext(0)
    hdr = synth.make_hdr_buffer(320, 200, ctx.device)
    ae = api.AutoExposure(ctx)
    ae.execute(hdr, 0.0, False)
    tm = B.ToneMappingAttribs.default(4)
    a = ae.tone_map(hdr, tm, flags=1)
    b = ctx.tone_map(hdr, tm, ae.average(), flags=1)
    assert torch.equal(a, b)
    # a frame darker than MinLuminance leaves the average alone; reset restores the initial 0.1
    dark = torch.full_like(hdr, 1e-3)
    before = ae.average()
    ae.execute(dark, 1.0, False)
    assert ae.average() == pytest.approx(before, rel=1e-6)
    ae.reset()
    assert ae.average() == pytest.approx(0.1, rel=1e-6)
    ae.close()
    ctx.close()


def test_chain_with_auto_exposure(mifx_lib):
    """With auto exposure on, the chain's output equals its output with the constant fAveLogLum replaced by what the auto-exposure pass
    computes from the same Bloom output (GetAverageSceneLuminance of the adapted average)."""
    import chain_util
    from diligentfx_amd import api, binding as B, synth
    from util import blue_noise_tables

    sobol, tile = blue_noise_tables()
    w, h = 256, 160
    a, b = api.Chain(0, sobol, tile), api.Chain(0, sobol, tile)
    ibl = api.precompute_ibl(a.postfx, synth.make_sky_cube(32, a.device).clamp(max=200.0), lut_size=32, irradiance_size=8, prefiltered_size=16, lut_samples=32,
                             diffuse_samples=32, specular_samples=16)
    sa = chain_util.shade_attribs(len(ibl.pre) - 1)
    a.set_auto_exposure(True, elapsed_time_s=0.25, light_adaptation=True)
    out_a, out_b = torch.zeros(h, w, 4, device=a.device), torch.zeros(h, w, 4, device=a.device)
    scene = synth.Scene()
    ae = api.AutoExposure(b.postfx)  # the same sequence by hand next to chain b
    for frame in range(3):
        f = synth.make_frame(scene, frame, w, h, a.device)
        a.execute(a.bind_frame(frame, f, ibl, sa, out_a))
        # chain b: run once to get this frame's Bloom output, feed it to a stand-alone auto exposure, re-run the final tone map with its value
        b.execute(b.bind_frame(frame, f, ibl, sa, out_b))
        d = B.Image2D()
        B.check(b.lib.mifx_bloom_get_output(_bloom_handle(b), __import__("ctypes").byref(d)))
        bloom_out = api._view(d, b.device).contiguous()
        ae.execute(bloom_out, 0.25, True)
        want = b.postfx.tone_map(bloom_out, b.tone_mapping, ae.average(), flags=b.tonemap_flags)
        assert a.auto_exposure_average() == pytest.approx(ae.average(), rel=1e-6)
        assert torch.equal(out_a, want)
    ae.close()
    a.close()
    b.close()


def _bloom_handle(chain):
    import ctypes

    # the chain owns its effects; the Bloom object is reachable through the intermediate-plane accessor of the chain's Bloom
    h = ctypes.c_void_p()
    from diligentfx_amd import binding as B

    B.check(chain.lib.mifx_chain_get_effect(chain.handle, b"bloom", ctypes.byref(h)))
    return h
