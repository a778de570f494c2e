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
