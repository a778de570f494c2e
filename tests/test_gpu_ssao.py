"""GPU parity of the SSAO passes (A2..A8): each HIP pass is compared with the checker fed with the HIP pass' own inputs
(per-pass isolation), over several frames so that the temporal path (A5/A7/A8 history logic) is exercised."""
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
    if not o.has("oracle_ssao_compute_ao_gtao"):
        pytest.skip("no checker available for SSAO")
    return o, "oracle_"


# rev: PostFXContext::FEATURE_FLAG_REVERSED_DEPTH (SSAO_OPTION_INVERTED_DEPTH; the reference build has the GTAO permutation)
# the last case: FEATURE_FLAG_HALF_PRECISION_DEPTH (self-occlusion offset 5e-3; reference permutation of GTAO)
# fused: A7 + A8 as one resolve over work lists (the default) / as two full-frame passes (the last case)
@pytest.mark.parametrize("size,algo,rev,halfprec,fused", [((160, 96), "gtao", False, False, True), ((135, 70), "gtao", False, False, True), ((160, 96), "hbao", False, False, True),
                                                          ((160, 96), "vbao", False, False, True), ((152, 90), "gtao", True, False, True), ((144, 88), "gtao", False, True, True),
                                                          ((160, 96), "gtao", False, False, False), ((135, 70), "gtao", False, False, False)])
def test_ssao_per_pass_parity(mifx_lib, size, algo, rev, halfprec, fused):
    from diligentfx_amd import api, binding as B, synth

    lib, pfx = checker()
    import cpu_chain

    cc = cpu_chain.CpuChain(lib, pfx, reversed_depth=rev)
    w, h = size
    sobol, tile = blue_noise_tables()
    ctx = api.PostFXContext(0, sobol, tile)
    ssao = api.ScreenSpaceAmbientOcclusion(ctx)
    ssao.set_fused_resolve(fused)
    scene = synth.Scene()
    attribs = B.SSAOAttribs.default()
    attribs.Algorithm = {"gtao": 0, "hbao": 1, "vbao": 2}[algo]
    chain = cpu_chain.CpuChain(lib, pfx, algorithm=algo)
    worst = {}
    walked = 0
    for frame in range(4):
        f = synth.make_frame(scene, frame, w, h, ctx.device, reversed_depth=rev)
        ctx.prepare_resources(frame, w, h, feature_flags=(1 if rev else 0) | (2 if halfprec else 0))
        ssao.prepare_resources(feature_flags=1 if halfprec else 0)
        ctx.execute(f["depth"], f["prev_depth"], f["motion"], f["camera"], f["prev_camera"])
        # snapshot the HIP history that A5 is about to read (previous slot)
        st = ssao.execute(f["depth"], f["normal"], attribs)
        assert st == (1 if frame == 0 else 0)
        cam, prev = bytes(f["camera"]), bytes(f["prev_camera"])
        depth, normal = to_np(f["depth"]), to_np(f["normal"])
        a = B.SSAOAttribs.from_buffer_copy(bytes(attribs))
        a.ResetAccumulation = 1 if frame == 0 else 0
        ab = bytes(a)
        g = lambda n: to_np(ssao.get_intermediate(n))  # noqa: E731

        def cmp(name, got, want, frac=0.0):
            e, fr = assert_close(got, want, max_outlier_frac=frac, what=f"{algo} frame {frame} {name}")
            worst[name] = max(worst.get(name, 0.0), fr)

        # A2
        pyr = [depth] + [g(f"prefiltered_depth{k}") for k in range(1, 5)]
        for k in range(1, 5):
            want = np.zeros_like(pyr[k])
            cc.call("ssao_prefiltered_depth_mip", [pyr[k - 1]], [want], cam0=cam, attribs=ab, ival=[k - 1])
            cmp(f"A2 mip{k}", pyr[k], want)
        # A3: mip selection floor(lod+0.5) and the point-sample texel choice are discontinuous => allow a few flipped taps
        want = np.ones((h, w), np.float32)
        if halfprec and pfx == "ref_":
            cc.call("ssao_compute_ao_gtao_halfprec", [pyr, normal, to_np(ctx.get_2d_blue_noise(1))], [want], cam0=cam, attribs=ab)
        else:
            cc.call("ssao_compute_ao_" + algo, [pyr, normal, to_np(ctx.get_2d_blue_noise(1))], [want], cam0=cam, attribs=ab, ival=[0, 0, 0, 0, 0, int(halfprec)])
        cmp("A3", g("occlusion"), want)
        # A5 (inputs: HIP A3 output + the checker-side copy of the HIP history of the previous slot)
        if frame == 0:
            prev_ao, prev_len = np.ones((h, w), np.float32), np.ones((h, w), np.float32)
        w_ao, w_len = np.ones((h, w), np.float32), np.ones((h, w), np.float32)
        cc.call("ssao_temporal_accumulation", [g("occlusion"), prev_ao, prev_len, to_np(ctx.get_reprojected_depth()), to_np(f["prev_depth"]),
                                                      to_np(ctx.get_closest_motion_vectors())], [w_ao, w_len], cam0=cam, cam1=prev, attribs=ab)
        cmp("A5 ao", g("accum_ao"), w_ao)
        cmp("A5 len", g("history_len"), w_len)
        # A6
        apyr = [g("accum_ao")] + [g(f"conv_ao{k}") for k in range(1, 5)]
        dpyr = [depth] + [g(f"conv_depth{k}") for k in range(1, 5)]
        for k in range(1, 5):
            w0, w1 = np.zeros_like(apyr[k]), np.zeros_like(dpyr[k])
            cc.call("ssao_convoluted_history_mip", [apyr[k - 1], dpyr[k - 1]], [w0, w1], ival=[k - 1])
            cmp(f"A6 ao mip{k}", apyr[k], w0)
            cmp(f"A6 depth mip{k}", dpyr[k], w1)
        # A7
        want = np.zeros((h, w), np.float32)
        cc.call("ssao_resampled_history", [apyr, dpyr, g("history_len"), normal], [want], cam0=cam)
        resampled = g("resampled")  # (fused resolve: A7's copy by the temporal pass, the walk pass on top of it -- the same plane)
        walked += int((~(depth < 1e-6 if rev else depth >= np.float32(1.0 - 1e-6)) & ((g("history_len") - np.float32(1.0)) / np.float32(4.0) < 1.0)).sum())
        cmp("A7", resampled, want)
        # A8
        want = np.zeros((h, w), np.float32)
        cc.call("ssao_spatial_reconstruction", [resampled, g("history_len"), depth, normal], [want], cam0=cam, attribs=ab)
        out = to_np(ssao.get_ambient_occlusion())
        cmp("A8", out, want)
        assert np.array_equal(g("history_ao"), out)  # the history write-back of the resolve
        prev_ao, prev_len = out.copy(), g("history_len").copy()
    assert walked > 0  # the walk path was exercised
    print("worst outlier fractions:", {k: v for k, v in worst.items() if v > 0})
    ssao.close()
    ctx.close()


def test_ssao_end_to_end_vs_cpu_chain(mifx_lib):
    """Whole SSAO effect over 5 frames against the CPU chain running independently (errors may accumulate through the history)."""
    from diligentfx_amd import api, binding as B, synth

    lib, pfx = checker()
    w, h = 192, 112
    sobol, tile = blue_noise_tables()
    ctx = api.PostFXContext(0, sobol, tile)
    ssao = api.ScreenSpaceAmbientOcclusion(ctx)
    chain = cpu_chain.CpuChain(lib, pfx)
    scene = synth.Scene()
    attribs = B.SSAOAttribs.default()
    for frame in range(5):
        f = synth.make_frame(scene, frame, w, h, ctx.device)
        ctx.prepare_resources(frame, w, h)
        ssao.prepare_resources()
        ctx.execute(f["depth"], f["prev_depth"], f["motion"], f["camera"], f["prev_camera"])
        ssao.execute(f["depth"], f["normal"], attribs)
        pf = chain.postfx(frame, to_np(f["depth"]), to_np(f["prev_depth"]), to_np(f["motion"]), bytes(f["camera"]), bytes(f["prev_camera"]), (sobol, tile))
        want = chain.ssao(pf, to_np(f["depth"]), to_np(f["normal"]), attribs)
        got = to_np(ssao.get_ambient_occlusion())
        assert_close(got, want, max_outlier_frac=1e-4, what=f"SSAO output frame {frame}")  # (the effect end to end over several frames; measured 4.65e-5 = one value)
    # frame-index gap => history reset is reported
    ctx.prepare_resources(20, w, h)
    ssao.prepare_resources()
    ctx.execute(f["depth"], f["prev_depth"], f["motion"], f["camera"], f["prev_camera"])
    assert ssao.execute(f["depth"], f["normal"], attribs) == 1


@pytest.mark.parametrize("size,rev", [((192, 112), False), ((150, 92), False), ((176, 100), True)])
def test_ssao_fused_resolve_is_bit_identical(mifx_lib, size, rev):
    """A7 + A8 as one resolve (inside the temporal pass + walk list + spatial list: ssao.hip) against the two full-frame passes: two effect objects fed with the same thirteen frames
    (reset, growing history up to saturation, a frame-index gap, AlphaInterpolation != 1) must agree on every texel of the output and of the history, bit for bit; the cases cover
    centred and general A7 taps and reversed depth."""
    from diligentfx_amd import api, binding as B, synth

    w, h = size
    sobol, tile = blue_noise_tables()
    ctx = api.PostFXContext(0, sobol, tile)
    fused, plain = api.ScreenSpaceAmbientOcclusion(ctx), api.ScreenSpaceAmbientOcclusion(ctx)
    plain.set_fused_resolve(False)
    scene = synth.Scene()
    attribs = B.SSAOAttribs.default()
    attribs.AlphaInterpolation = 0.85
    outputs = set()
    for frame in (*range(11), 15, 16):  # history lengths 1 .. 11: walk (< 5), filter without walk (5 .. 8), saturated (>= 9); then a gap
        f = synth.make_frame(scene, frame, w, h, ctx.device, reversed_depth=rev)
        ctx.prepare_resources(frame, w, h, feature_flags=1 if rev else 0)
        ctx.execute(f["depth"], f["prev_depth"], f["motion"], f["camera"], f["prev_camera"])
        res = []
        for fx in (fused, plain):
            fx.prepare_resources()
            st = fx.execute(f["depth"], f["normal"], attribs)
            assert st == (1 if frame in (0, 15) else 0)
            res.append((to_np(fx.get_ambient_occlusion()).copy(), to_np(fx.get_intermediate("history_ao")).copy(), to_np(fx.get_intermediate("history_len")).copy(),
                        to_np(fx.get_intermediate("resampled")).copy(), to_np(fx.get_intermediate("accum_ao")).copy()))
            outputs.add((id(fx), fx.get_ambient_occlusion().data_ptr()))
        for a, b, what in zip(res[0], res[1], ("output", "history_ao", "history_len", "resampled", "accum_ao")):
            assert np.array_equal(a, b), f"frame {frame} {what}: {(a != b).sum()} texels differ"
        assert np.array_equal(res[0][0], res[0][1])  # output == history of the frame
        assert res[0][0].min() < 0.95 and np.isfinite(res[0][0]).all()
    assert len(outputs) == 2  # GetAmbientOcclusionSRV: one plane per object for all frames (a descriptor fetched once stays valid)
    fused.close()
    plain.close()
    ctx.close()


def test_ssao_protocol_errors(mifx_lib):
    from diligentfx_amd import api, binding as B, synth

    sobol, tile = blue_noise_tables()
    ctx = api.PostFXContext(0, sobol, tile)
    ssao = api.ScreenSpaceAmbientOcclusion(ctx)
    with pytest.raises(B.MifxError, match="INVALID_OP"):
        ssao.prepare_resources()  # PostFX not prepared
    ctx.prepare_resources(0, 64, 48)
    ssao.prepare_resources()
    d, n = torch.ones(48, 64, device=ctx.device), torch.zeros(48, 64, 4, device=ctx.device)
    with pytest.raises(B.MifxError, match="INVALID_OP"):
        ssao.execute(d, n, B.SSAOAttribs.default())  # PostFX execute missing
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        ssao.prepare_resources(feature_flags=4)  # (ScreenSpaceAmbientOcclusion::FEATURE_FLAGS has bits 0 and 1)


def test_ssao_full_size_parity(mifx_lib):
    """BASELINE configs[1]: SSAO on a 1920x1080 depth + normal G-buffer (GTAO, full resolution), three frames against the CPU chain.
    The checker finishes a 1080p frame in seconds on the host cores; the temporal history runs on both sides."""
    from diligentfx_amd import api, binding as B, synth

    lib, pfx = checker()
    w, h = 1920, 1080
    sobol, tile = blue_noise_tables()
    ctx = api.PostFXContext(0, sobol, tile)
    ssao = api.ScreenSpaceAmbientOcclusion(ctx)
    chain = cpu_chain.CpuChain(lib, pfx)
    scene = synth.Scene()
    attribs = B.SSAOAttribs.default()
    for frame in range(16, 19):
        f = synth.make_frame(scene, frame, w, h, ctx.device)
        ctx.prepare_resources(frame, w, h)
        ssao.prepare_resources()
        ctx.execute(f["depth"], f["prev_depth"], f["motion"], f["camera"], f["prev_camera"])
        ssao.execute(f["depth"], f["normal"], attribs)
        pf = chain.postfx(frame, to_np(f["depth"]), to_np(f["prev_depth"]), to_np(f["motion"]), bytes(f["camera"]), bytes(f["prev_camera"]), (sobol, tile))
        want = chain.ssao(pf, to_np(f["depth"]), to_np(f["normal"]), attribs)
        got = to_np(ssao.get_ambient_occlusion())
        assert got.shape == (h, w) and np.isfinite(got).all()  # (the GTAO arc integral is not clamped: values slightly above 1 occur in the reference too)
        assert_close(got, want, max_outlier_frac=2.5e-4, what=f"SSAO 1920x1080 frame {frame}")  # (measured 1.18e-4)
    ssao.close()
    ctx.close()


@pytest.mark.parametrize("size,algorithm,first", [((160, 96), "gtao", 0), ((150, 92), "gtao", 0), ((160, 96), "hbao", 0), ((150, 92), "vbao", 0), ((70, 36), "gtao", 47)])
def test_ssao_half_resolution(mifx_lib, size, algorithm, first):
    """FEATURE_FLAG_HALF_RESOLUTION: A1 checkerboard depth (bit-exact), pyramid + GTAO at half size, A4 bilateral upsampling, the full-size tail;
    every new pass against the checker on the HIP path's own inputs, the result against the checker's own run of the effect.
    The case that starts at camera position 47 (found by the random sequences of test_gpu_host_sequence.py) has a floor at a glancing angle at
    position 48: the nine depth weights of A4 are exp(-103.8) on whole rows, 0.6 of the smallest denormal number (ssao.hip, the comment in A4)."""
    import cpu_chain
    from diligentfx_amd import api, binding as B, synth

    lib, pfx = checker()
    cc, e2e = cpu_chain.CpuChain(lib, pfx, algorithm=algorithm), cpu_chain.CpuChain(lib, pfx, algorithm=algorithm)
    w, h = size
    sobol, tile = blue_noise_tables()
    ctx = api.PostFXContext(0, sobol, tile)
    ssao = api.ScreenSpaceAmbientOcclusion(ctx)
    scene = synth.Scene()
    attribs = B.SSAOAttribs.default()
    attribs.Algorithm = {"gtao": 0, "hbao": 1, "vbao": 2}[algorithm]
    for frame in range(first, first + 3):
        f = synth.make_frame(scene, frame, w, h, ctx.device)
        ctx.prepare_resources(frame, w, h)
        ssao.prepare_resources(feature_flags=2)
        ctx.execute(f["depth"], f["prev_depth"], f["motion"], f["camera"], f["prev_camera"])
        ssao.execute(f["depth"], f["normal"], attribs)
        cam, prev = bytes(f["camera"]), bytes(f["prev_camera"])
        depth, normal = to_np(f["depth"]), to_np(f["normal"])
        a = B.SSAOAttribs.from_buffer_copy(bytes(attribs))
        a.ResetAccumulation = 1 if frame == first else 0
        ab = bytes(a)
        g = lambda n: to_np(ssao.get_intermediate(n))  # noqa: E731
        hw, hh = w // 2, h // 2
        # A1
        want = np.zeros((hh, hw), np.float32)
        cc.call("ssao_downsampled_depth", [depth], [want])
        checker_depth = g("checkerboard_depth")
        assert np.array_equal(checker_depth, want) and checker_depth.shape == (hh, hw)
        # A2 on the half-size pyramid
        pyr = [checker_depth] + [g(f"prefiltered_depth{k}") for k in range(1, 5)]
        for k in range(1, 5):
            assert pyr[k].shape == (max(hh >> k, 1), max(hw >> k, 1))
            want = np.zeros_like(pyr[k])
            cc.call("ssao_prefiltered_depth_mip", [pyr[k - 1]], [want], cam0=cam, attribs=ab, ival=[k - 1])
            assert_close(pyr[k], want, what=f"half-res A2 mip{k} frame {frame}")
        # A3 at half size (mip / texel selection is discontinuous: a few flipped taps)
        want = np.ones((hh, hw), np.float32)
        if pfx == "ref_":
            cc.call(f"ssao_compute_ao_{algorithm}_half", [pyr, normal, to_np(ctx.get_2d_blue_noise(1))], [want], cam0=cam, attribs=ab)
        else:
            cc.call(f"ssao_compute_ao_{algorithm}", [pyr, normal, to_np(ctx.get_2d_blue_noise(1))], [want], cam0=cam, attribs=ab, ival=[0, 0, 0, 0, 0, 0, 1])
        assert_close(g("occlusion"), want, max_outlier_frac=0.0, what=f"half-res A3 frame {frame}")
        # A4
        want = np.zeros((h, w), np.float32)
        cc.call("ssao_bilateral_upsampling", [depth, g("occlusion")], [want], cam0=cam, attribs=ab)
        # (a pixel whose nine depth weights all underflow takes the fallback branch: "WeightSum > 0" is a threshold on denormal numbers)
        assert_close(g("occlusion_upsampled"), want, max_outlier_frac=0.0, what=f"A4 frame {frame}")
        assert np.isfinite(g("occlusion_upsampled")).all() and np.isfinite(g("occlusion")).all()
        # end to end
        pf = e2e.postfx(frame, depth, to_np(f["prev_depth"]), to_np(f["motion"]), cam, prev, (sobol, tile))
        want = e2e.ssao(pf, depth, normal, attribs, half_resolution=True)
        out = to_np(ssao.get_ambient_occlusion())
        assert out.shape == (h, w)
        # (measured 1.95e-4 on an MI355X in both the shipped and the strict build: profiles/r03_parity_outliers_strict_vs_shipped.txt)
        assert_close(out, want, max_outlier_frac=4e-4, what=f"half-res SSAO end to end frame {frame}")  # (measured 1.95e-4)
        assert out.min() < 0.9 and np.isfinite(out).all()
    ssao.close()
    ctx.close()
