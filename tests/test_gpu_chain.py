"""GPU: IBL precompute (I1-I3) parity and the whole chain (mifx_chain_execute) against the CPU chain, frame by frame."""
import numpy as np
import pytest
import torch

import cpu_chain
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


def test_ibl_precompute_parity(mifx_lib):
    import chain_util
    from diligentfx_amd import api, synth

    lib, pfx = checker("ibl_brdf_lut")
    ctx = api.PostFXContext(0)
    env = synth.make_sky_cube(32, ctx.device).clamp(max=200.0)
    ibl = api.precompute_ibl(ctx, env, lut_size=32, irradiance_size=8, prefiltered_size=16, lut_samples=64, diffuse_samples=128, specular_samples=32)
    want = chain_util.make_ibl(lib, pfx)  # same sizes / sample counts
    assert_close(to_np(ibl.lut), want["lut"], what="BRDF LUT")
    # the per-sample mip level (log2 of a pdf ratio) and cube-face selection are discontinuous: a few samples may land on another texel
    assert_close(to_np(ibl.irr[0]), want["irradiance"][0], max_outlier_frac=0.0, what="irradiance")
    for m, (g, w) in enumerate(zip(ibl.pre, want["prefiltered"])):
        assert_close(to_np(g), w, max_outlier_frac=0.0, what=f"prefiltered mip {m}")
    # known answers of the split-sum LUT: A + B -> 1 for a smooth surface seen head-on, energy is bounded
    lut = to_np(ibl.lut)
    assert 0.9 < lut[0, -1].sum() <= 1.01 and (lut >= 0).all() and lut.sum(-1).max() <= 1.05
    ctx.close()


def test_chain_vs_cpu_chain(mifx_lib):
    """6 frames of the full chain.  Every stochastic / temporal stage runs independently on both sides, so flipped SSR rays and
    thresholded decisions accumulate: the final LDR image must agree on all but a small fraction of texel-channels."""
    import chain_util
    from diligentfx_amd import api, synth

    lib, pfx = checker("pbr_shade")
    w, h = 224, 128
    sobol, tile = blue_noise_tables()
    chain = api.Chain(0, sobol, tile)
    ibl_np = chain_util.make_ibl(lib, pfx)
    ibl = api.IBLResources(torch.from_numpy(ibl_np["lut"]).to(chain.device), [torch.from_numpy(m).to(chain.device) for m in ibl_np["irradiance"]],
                           [torch.from_numpy(m).to(chain.device) for m in ibl_np["prefiltered"]])
    cpu = cpu_chain.CpuChain(lib, pfx)
    scene = synth.Scene()
    sa = chain_util.shade_attribs(len(ibl_np["prefiltered"]) - 1)
    out = torch.zeros(h, w, 4, device=chain.device)
    fracs = []
    for frame in range(6):
        f = synth.make_frame(scene, frame, w, h, chain.device)
        chain.execute(chain.bind_frame(frame, f, ibl, sa, out))
        want = chain_util.run_frame(cpu, scene, frame, w, h, ibl_np)
        got = to_np(out)
        assert np.isfinite(got).all()
        # the contract's 1e-3; budget = 2.5 x the fraction measured on the steady-state run at this frame count (tests/test_gpu_steady_state.py: 1.9e-3 by frame 17),
        # second tier: at most 2e-4 of the values off by more than 5e-2
        _, frac = assert_close(got, want, max_outlier_frac=5e-3, outlier_cap=(5e-2, 2e-4), what=f"final image frame {frame}")
        fracs.append(frac)
        assert np.abs(got[..., :3] - want[..., :3]).mean() < 1e-3  # and the images are the same picture
    print("outlier fractions per frame:", [round(x, 5) for x in fracs])
    # history reset: replaying frame 0 after reset_history reproduces the first output exactly
    chain.reset_history()
    f0 = synth.make_frame(scene, 0, w, h, chain.device)
    first = torch.zeros_like(out)
    chain.execute(chain.bind_frame(0, f0, ibl, sa, first))
    a = first.clone()
    chain.reset_history()
    chain.execute(chain.bind_frame(0, f0, ibl, sa, first))
    assert torch.equal(a, first)
    chain.close()


# Outlier budgets of the 3840x2160 history = ~2 x the fractions measured on an MI355X (profiles/r06_full_size_parity.txt); they grow with the frame number as flipped SSR rays and
# history decisions travel through the temporal filters (the same growth tests/test_gpu_steady_state.py shows at 384x216, smaller here: a 4K pixel is a small solid angle)
FULL_SIZE_CHECK = (1, 2, 8, 17)  # SURVEY 8d / 4 iv (1-based)
# measured (round 6, both sides on the same input arrays): final 1.4e-5 / 1.3e-4 / 1.2e-4 / 1.7e-4 at frames 1 / 2 / 8 / 17; frame 17: SSAO 1.7e-4, SSR 1.1e-3, TAA 6.0e-4
FULL_SIZE_BUDGET = {"final": {1: 5e-5, 2: 3e-4, 8: 3e-4, 17: 4e-4}, "ssao": 4e-4, "ssr": 2.2e-3, "taa": 1.2e-3}


def test_chain_full_size_parity(mifx_lib):
    """BASELINE configs[3]: frames 1, 2, 8 and 17 of ONE history of the full chain at 3840x2160 against the CPU chain (the frames at which the temporal passes
    TAA_ComputeTemporalAccumulation.fx:229 / SSR_ComputeTemporalAccumulation.fx:224 / SSAO's history have reset, warmed up and saturated) -- the reference shader source on
    the host cores takes a few seconds per 4K frame, the seventeen frames about a minute.  Both sides get the same input arrays (rendered once, on the device).  Same
    acceptance as the small-size chain test: SSR rays and thresholded decisions that flip in one implementation change isolated texels, the images must agree everywhere
    else; at frame 17 the SSAO, SSR and TAA outputs are compared as well and the saturated paths must have run (history length 16)."""
    import os

    import chain_util
    from diligentfx_amd import api, synth

    lib, pfx = checker("pbr_shade")
    w, h = 3840, 2160
    sobol, tile = blue_noise_tables()
    chain = api.Chain(0, sobol, tile)
    ibl_np = chain_util.make_ibl(lib, pfx)
    ibl = api.IBLResources(torch.from_numpy(ibl_np["lut"]).to(chain.device), [torch.from_numpy(m).to(chain.device) for m in ibl_np["irradiance"]],
                           [torch.from_numpy(m).to(chain.device) for m in ibl_np["prefiltered"]])
    cpu = cpu_chain.CpuChain(lib, pfx)
    scene = synth.Scene()
    sa = chain_util.shade_attribs(len(ibl_np["prefiltered"]) - 1)
    out = torch.zeros(h, w, 4, device=chain.device)
    measure = bool(os.environ.get("MIFX_PARITY_MEASURE"))
    report = []
    for n in range(1, max(FULL_SIZE_CHECK) + 1):
        frame = 15 + n
        f = synth.make_frame(scene, frame, w, h, chain.device)
        chain.execute(chain.bind_frame(frame, f, ibl, sa, out))
        g = {k: to_np(v) for k, v in f.items() if isinstance(v, torch.Tensor)}
        keep = {} if n == max(FULL_SIZE_CHECK) else None
        want = chain_util.run_frame_inputs(cpu, g, bytes(f["camera"]), bytes(f["prev_camera"]), frame, ibl_np, sa, keep)
        if n not in FULL_SIZE_CHECK:
            continue
        got = to_np(out)
        assert np.isfinite(got).all()
        fr = {}
        _, fr["final"] = assert_close(got, want, max_outlier_frac=1.0 if measure else FULL_SIZE_BUDGET["final"][n], outlier_cap=None if measure else (5e-2, 2e-4),
                                      what=f"3840x2160 final image, frame {n} of one history")
        assert np.abs(got[..., :3] - want[..., :3]).mean() < 1e-3
        if keep is not None:
            for name, key, cap in (("ssao", "ssao_out", 0.5), ("ssr", "ssr_out", None), ("taa", "taa_out", None)):
                _, fr[name] = assert_close(to_np(chain.effect_output(name)), keep[key], max_outlier_frac=1.0 if measure else FULL_SIZE_BUDGET[name], outlier_cap=None if measure else cap,
                                           what=f"3840x2160 {name.upper()} output, frame {n} of one history")
            hist_len = to_np(chain.effect("ssao").get_intermediate("history_len"))
            geom = g["depth"] < 1.0 - 1e-6
            fr["len16"] = float((hist_len[geom] >= 16.0).mean())
            assert hist_len.max() == 16.0 and fr["len16"] > 0.2, fr  # the saturated paths ran (SSAO_MAX_HISTORY_LENGTH; measured: 28 % of the surface pixels at 16 by frame 17)
        report.append((n, fr))
        print(f"3840x2160 frame {n:2d} of one history: " + " ".join(f"{k} {v:.2e}" for k, v in fr.items()), flush=True)
    chain.close()


def test_chain_8k_frame_pair_parity(mifx_lib):
    """BASELINE configs[4]'s frame size: two consecutive 7680x4320 frames of the unsharded chain against the CPU chain (the sharded frame is held to the unsharded one bit
    for bit in tests/test_gpu_sharded.py and by bench.py's shard_verified; this is the missing leg -- the unsharded frame at that size against the reference).  ~10 s of
    reference CPU time per frame on the box's cores; the same arrays go to both sides."""
    import os

    import chain_util
    from diligentfx_amd import api, synth

    lib, pfx = checker("pbr_shade")
    w, h = 7680, 4320
    sobol, tile = blue_noise_tables()
    chain = api.Chain(0, sobol, tile)
    ibl_np = chain_util.make_ibl(lib, pfx)
    ibl = api.IBLResources(torch.from_numpy(ibl_np["lut"]).to(chain.device), [torch.from_numpy(m).to(chain.device) for m in ibl_np["irradiance"]],
                           [torch.from_numpy(m).to(chain.device) for m in ibl_np["prefiltered"]])
    cpu = cpu_chain.CpuChain(lib, pfx)
    scene = synth.Scene()
    sa = chain_util.shade_attribs(len(ibl_np["prefiltered"]) - 1)
    out = torch.zeros(h, w, 4, device=chain.device)
    measure = bool(os.environ.get("MIFX_PARITY_MEASURE"))
    for frame in range(16, 18):
        f = synth.make_frame(scene, frame, w, h, chain.device)
        chain.execute(chain.bind_frame(frame, f, ibl, sa, out))
        g = {k: to_np(v) for k, v in f.items() if isinstance(v, torch.Tensor)}
        want = chain_util.run_frame_inputs(cpu, g, bytes(f["camera"]), bytes(f["prev_camera"]), frame, ibl_np, sa)
        got = to_np(out)
        assert np.isfinite(got).all()
        # measured: 1.4e-4 at the second frame (the first, a reset frame, less)
        _, frac = assert_close(got, want, max_outlier_frac=1.0 if measure else 3e-4, outlier_cap=None if measure else (5e-2, 2e-4), what=f"7680x4320 final image frame {frame}")
        assert np.abs(got[..., :3] - want[..., :3]).mean() < 1e-3
        print(f"7680x4320 frame {frame}: outlier fraction {frac:.5f}", flush=True)
        del g, want, got
    chain.close()


def test_chain_full_size_properties(mifx_lib):
    """BASELINE config 4 size (3840x2160): size-independent properties of the chain output."""
    from diligentfx_amd import api, synth

    sobol, tile = blue_noise_tables()
    chain = api.Chain(0, sobol, tile)
    w, h = 3840, 2160
    env = synth.make_sky_cube(64, chain.device)
    ibl = api.precompute_ibl(chain.postfx, env, lut_size=128, irradiance_size=16, prefiltered_size=64, lut_samples=128, diffuse_samples=512, specular_samples=64)
    sa = synth.make_lights()
    sa.PrefilteredCubeLastMip = float(len(ibl.pre) - 1)
    scene = synth.Scene()
    out = torch.zeros(h, w, 4, device=chain.device)
    outs = []
    for frame in range(3):
        f = synth.make_frame(scene, frame, w, h, chain.device)
        chain.execute(chain.bind_frame(frame, f, ibl, sa, out))
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        assert float(out[..., :3].min()) >= 0.0 and float(out[..., :3].max()) < 4.0  # tone-mapped + sRGB-encoded (Uncharted2 may exceed 1 on highlights)
        outs.append(out.clone())
    # determinism: replaying the same three frames from a reset gives bit-identical images
    chain.reset_history()
    for frame in range(3):
        f = synth.make_frame(scene, frame, w, h, chain.device)
        chain.execute(chain.bind_frame(frame, f, ibl, sa, out))
        assert torch.equal(out, outs[frame]), f"frame {frame} is not reproducible"
    chain.close()


def test_chain_reversed_depth(mifx_lib):
    """PostFXContext::FEATURE_FLAG_REVERSED_DEPTH (HnPostProcessTask.cpp:666-670): frames rendered with a reversed projection (near = 1, background = 0)
    through the chain against the reversed permutation of the checker, and against the same scene in the normal convention."""
    import ctypes

    import chain_util
    from diligentfx_amd import api, binding as B, synth

    lib, pfx = checker("pbr_shade")
    w, h = 224, 128
    sobol, tile = blue_noise_tables()
    chain, plain = api.Chain(0, sobol, tile), api.Chain(0, sobol, tile)
    chain.set_postfx_feature_flags(1)
    ibl_np = chain_util.make_ibl(lib, pfx)
    ibl = api.IBLResources(torch.from_numpy(ibl_np["lut"]).to(chain.device), [torch.from_numpy(m).to(chain.device) for m in ibl_np["irradiance"]],
                           [torch.from_numpy(m).to(chain.device) for m in ibl_np["prefiltered"]])
    cpu = cpu_chain.CpuChain(lib, pfx, reversed_depth=True)
    scene = synth.Scene()
    sa = chain_util.shade_attribs(len(ibl_np["prefiltered"]) - 1)
    out, out_plain = torch.zeros(h, w, 4, device=chain.device), torch.zeros(h, w, 4, device=chain.device)

    def plane(effect, name):
        hnd, d = ctypes.c_void_p(), B.Image2D()
        B.check(chain.lib.mifx_chain_get_effect(chain.handle, effect.encode(), ctypes.byref(hnd)))
        B.check(getattr(chain.lib, f"mifx_{effect}_get_intermediate")(hnd, name.encode(), ctypes.byref(d)))
        return to_np(api._view(d, chain.device))

    for frame in range(4):
        f = synth.make_frame(scene, frame, w, h, chain.device, reversed_depth=True)
        assert float(f["depth"].min()) == 0.0 and 0.0 < float(f["depth"].max()) < 0.1 and f["camera"].fFarPlaneDepth == 0.0
        chain.execute(chain.bind_frame(frame, f, ibl, sa, out))
        keep = {}
        want = chain_util.run_frame(cpu, scene, frame, w, h, ibl_np, keep)
        got = to_np(out)
        assert np.isfinite(got).all()
        # integer-like work of the convention, on the GPU's own inputs: the depth hierarchy (max instead of min) and the reflection mask are bit-exact
        level = to_np(f["depth"])
        for k in range(1, 7):
            nxt = np.zeros((max(h >> k, 1), max(w >> k, 1)), np.float32)
            cpu.call("ssr_hiz_mip", [level], [nxt], ival=[k - 1])
            assert np.array_equal(plane("ssr", f"hiz{k}"), nxt), f"hiz{k}"
            level = nxt
        wr, wm = np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)
        cpu.call("ssr_mask_roughness", [to_np(f["material"]), to_np(f["depth"])], [wr, wm], attribs=bytes(B.SSRAttribs.default()))
        assert np.array_equal(plane("ssr", "mask"), wm) and 0.05 < wm.mean() < 0.95
        _, frac = assert_close(got, want, max_outlier_frac=5e-3, outlier_cap=(5e-2, 2e-4), what=f"reversed depth, final image frame {frame}")
        assert np.abs(got[..., :3] - want[..., :3]).mean() < 1e-3
        # the same scene in the normal convention gives the same picture (the two depth encodings round differently, nothing else differs)
        g = synth.make_frame(scene, frame, w, h, plain.device)
        plain.execute(plain.bind_frame(frame, g, ibl, sa, out_plain))
        assert float((out - out_plain)[..., :3].abs().mean()) < 4e-3
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        chain.set_postfx_feature_flags(8)  # unknown flag (4 = FEATURE_FLAG_TEMPORAL_UPSCALING is accepted since round 2)
    chain.close()
    plain.close()


def test_chain_effect_feature_flags(mifx_lib):
    """mifx_chain_set_effect_feature_flags: half-resolution SSAO and previous-frame SSR inside the chain (same picture, different pixels)."""
    import chain_util
    from diligentfx_amd import api, binding as B, synth

    w, h = 256, 144
    sobol, tile = blue_noise_tables()
    dev = torch.device("cuda", 0)
    chains = [api.Chain(0, sobol, tile) for _ in range(4)]
    chains[1].set_effect_feature_flags(ssao_feature_flags=2)
    chains[2].set_effect_feature_flags(ssr_feature_flags=1)
    chains[3].set_effect_feature_flags(ssr_feature_flags=2)
    ibl = api.precompute_ibl(chains[0].postfx, synth.make_sky_cube(32, dev).clamp(max=200.0), lut_size=32, irradiance_size=8, prefiltered_size=32, lut_samples=32,
                             diffuse_samples=32, specular_samples=16)
    sa = chain_util.shade_attribs(len(ibl.pre) - 1)
    outs = [torch.zeros(h, w, 4, device=dev) for _ in chains]
    scene = synth.Scene()
    for frame in range(4):
        f = synth.make_frame(scene, frame, w, h, dev)
        for c, o in zip(chains, outs):
            c.execute(c.bind_frame(frame, f, ibl, sa, o))
        torch.cuda.synchronize()
        assert all(bool(torch.isfinite(o).all()) for o in outs)
        for k in (1, 2, 3):
            assert float((outs[0] - outs[k])[..., :3].abs().mean()) < 3e-2
            if k != 2 or frame > 0:  # (frame 0 of the synthetic sequence has no camera motion: previous-frame SSR reads the same texels)
                assert not torch.equal(outs[0], outs[k])
    assert chains[1].effect_output("ssao").shape == (h, w)
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        chains[0].set_effect_feature_flags(ssao_feature_flags=8)  # unknown flag
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        chains[0].set_effect_feature_flags(ssr_feature_flags=4)   # unknown flag
    chains[0].set_row_band(0, h // 2, 8)
    chains[0].set_effect_feature_flags(ssao_feature_flags=2)  # (round 3: the half-resolution variants run inside the row-band phases, tests/test_gpu_sharded.py)
    for c in chains:
        c.close()


def test_chain_all_options_together(mifx_lib):
    """Every optional piece at once: reversed depth, half-resolution SSAO and SSR, previous-frame SSR, depth of field (temporal + Karis), auto exposure, stage
    profiling -- the options only interact through the planes they hand each other, so the chain must run, stay finite and be deterministic."""
    import chain_util
    from diligentfx_amd import api, binding as B, synth

    w, h = 256, 144
    sobol, tile = blue_noise_tables()
    dev = torch.device("cuda", 0)
    scene = synth.Scene()

    def run():
        c = api.Chain(0, sobol, tile)
        c.set_postfx_feature_flags(1)
        c.set_effect_feature_flags(ssao_feature_flags=2, ssr_feature_flags=3)
        attribs = B.DOFAttribs.default()
        attribs.MaxCircleOfConfusion = 0.02
        c.set_depth_of_field(attribs, 3)
        c.set_auto_exposure(True, 1.0 / 60.0, True)
        ibl = api.precompute_ibl(c.postfx, synth.make_sky_cube(32, dev).clamp(max=200.0), lut_size=32, irradiance_size=8, prefiltered_size=32, lut_samples=32,
                                 diffuse_samples=32, specular_samples=16)
        sa = chain_util.shade_attribs(len(ibl.pre) - 1)
        out = torch.zeros(h, w, 4, device=dev)
        frames = []
        for frame in range(4):
            f = synth.make_frame(scene, frame, w, h, dev, reversed_depth=True)
            f["camera"].fFocusDistance, f["camera"].fFStop, f["camera"].fFocalLength = 12.0, 1.2, 135.0
            c.execute(c.bind_frame(frame, f, ibl, sa, out))
            torch.cuda.synchronize()
            assert bool(torch.isfinite(out).all()) and float(out[..., :3].min()) >= 0.0 and float(out[..., :3].mean()) > 0.02
            frames.append(out.clone())
        for name in ("ssao", "ssr", "taa", "dof", "bloom"):
            assert bool(torch.isfinite(c.effect_output(name)).all()), name
        assert c.effect_output("ssao").shape == (h, w) and c.auto_exposure_average() > 0
        c.close()
        return frames

    a, b = run(), run()
    assert all(torch.equal(x, y) for x, y in zip(a, b))  # no atomics, no uninitialised reads: two runs give the same bits
    assert not torch.equal(a[0], a[3])


def test_chain_fusion_is_bit_identical(mifx_lib):
    """mifx_chain_set_fusion_mask: the copy-frame ToneMap as the tail of Bloom's final up-sample, SSR's mask / roughness pass as a by-product of the shade, SSR's
    bilateral cleanup inside the composite, SSAO's A7 + A8 as one resolve over work lists, the composite inside the TAA kernel (and the IBL apron copies kept across frames) give the same frame, Bloom /
    TAA / SSAO / SSR outputs and SSR planes, bit for bit, as the separate passes."""
    import chain_util
    from diligentfx_amd import api, synth

    lib, pfx = checker("pbr_shade")
    w, h = 208, 120
    sobol, tile = blue_noise_tables()
    fused, plain = api.Chain(0, sobol, tile), api.Chain(0, sobol, tile)
    plain.set_fusion_mask(0)
    fused.set_fusion_mask(api.Chain.FUSE_EVERY_SWITCH)  # (the default mask + the composite inside the TAA kernel, which is off by default: measured slower)
    fused.postfx.set_static_ibl(True)
    ibl_np = chain_util.make_ibl(lib, pfx)
    ibl = api.IBLResources(torch.from_numpy(ibl_np["lut"]).to(fused.device), [torch.from_numpy(m).to(fused.device) for m in ibl_np["irradiance"]],
                           [torch.from_numpy(m).to(fused.device) for m in ibl_np["prefiltered"]])
    scene = synth.Scene()
    sa = chain_util.shade_attribs(len(ibl_np["prefiltered"]) - 1)
    a, b = torch.zeros(h, w, 4, device=fused.device), torch.zeros(h, w, 4, device=fused.device)
    for mode, srgb, channel, perceptual in ((4, 1, 0, 1), (7, 0, 0, 1), (0, 1, 1, 0), (10, 1, 0, 1)):
        for c in (fused, plain):
            c.tone_mapping = B_tm(mode)
            c.tonemap_flags = srgb
            c.ssr_attribs.RoughnessChannel, c.ssr_attribs.IsRoughnessPerceptual = channel, perceptual
            c.reset_history()
        for frame in range(7 if mode == 4 else 3):  # (the first case runs long enough for SSAO's history to pass the thresholds of A7 and A8)
            f = synth.make_frame(scene, frame, w, h, fused.device)
            fused.execute(fused.bind_frame(frame, f, ibl, sa, a))
            plain.execute(plain.bind_frame(frame, f, ibl, sa, b))
            assert torch.equal(a, b), (mode, srgb, frame, int((a != b).sum()))
            for fx in ("bloom", "taa", "ssao"):
                assert torch.equal(fused.effect_output(fx), plain.effect_output(fx)), (fx, frame)
            for name in ("mask", "roughness", "hist_radiance", "hist_variance"):
                assert torch.equal(fused.effect("ssr").get_intermediate(name), plain.effect("ssr").get_intermediate(name)), name
            for name in ("history_ao", "history_len"):
                assert torch.equal(fused.effect("ssao").get_intermediate(name), plain.effect("ssao").get_intermediate(name)), name
            # (the fused chain has not run R7: mifx_ssr_get_output produces the plane on demand)
            assert torch.equal(fused.effect_output("ssr"), plain.effect_output("ssr"))
    fused.close()
    plain.close()


def test_chain_refuses_specular_glossiness_shade(mifx_lib):
    """The chain's single material plane is the metallic-roughness Material target that SSR and the composite read; a specular-glossiness shade on it is refused."""
    import chain_util
    from diligentfx_amd import api, binding as B, synth

    lib, pfx = checker("pbr_shade")
    sobol, tile = blue_noise_tables()
    chain = api.Chain(0, sobol, tile)
    ibl_np = chain_util.make_ibl(lib, pfx)
    ibl = api.IBLResources(torch.from_numpy(ibl_np["lut"]).to(chain.device), [torch.from_numpy(m).to(chain.device) for m in ibl_np["irradiance"]],
                           [torch.from_numpy(m).to(chain.device) for m in ibl_np["prefiltered"]])
    sa = chain_util.shade_attribs(len(ibl_np["prefiltered"]) - 1)
    sa.Workflow = 1  # MIFX_PBR_WORKFLOW_SPECULAR_GLOSSINESS
    out = torch.zeros(64, 96, 4, device=chain.device)
    with pytest.raises(B.MifxError, match="Workflow"):
        chain.execute(chain.bind_frame(0, synth.make_frame(synth.Scene(), 0, 96, 64, chain.device), ibl, sa, out))
    chain.close()


def B_tm(mode):
    from diligentfx_amd import binding as B

    return B.ToneMappingAttribs.default(mode)


def test_chain_native_target(mifx_lib):
    """mifx_chain_execute_native: the frame written in the copy-frame target's format equals the fp32 frame exported afterwards, bit for bit."""
    import chain_util
    from diligentfx_amd import api, synth

    lib, pfx = checker("pbr_shade")
    w, h = 160, 96
    sobol, tile = blue_noise_tables()
    chain = api.Chain(0, sobol, tile)
    ibl_np = chain_util.make_ibl(lib, pfx)
    ibl = api.IBLResources(torch.from_numpy(ibl_np["lut"]).to(chain.device), [torch.from_numpy(m).to(chain.device) for m in ibl_np["irradiance"]],
                           [torch.from_numpy(m).to(chain.device) for m in ibl_np["prefiltered"]])
    scene = synth.Scene()
    sa = chain_util.shade_attribs(len(ibl_np["prefiltered"]) - 1)
    out = torch.zeros(h, w, 4, device=chain.device)
    for fmt in ("RGBA8_UNORM_SRGB", "RGBA16_FLOAT"):
        chain.reset_history()
        for frame in range(2):
            chain.execute(chain.bind_frame(frame, synth.make_frame(scene, frame, w, h, chain.device), ibl, sa, out))
        want = api.image_export(chain.postfx, out, fmt)
        chain.reset_history()
        for frame in range(2):
            raw = chain.execute_native(chain.bind_frame(frame, synth.make_frame(scene, frame, w, h, chain.device), ibl, sa, out), fmt)
        assert torch.equal(raw, want), fmt
    chain.close()


def test_rocTX_markers_do_not_disturb_the_chain(mifx_lib):
    """mifx_set_markers(1): ranges named after the reference's ScopedDebugGroup markers are pushed around the effects and passes (libroctx64 loaded on demand);
    the frame is the same as without them."""
    import chain_util
    from diligentfx_amd import api, synth

    lib, pfx = checker("pbr_shade")
    w, h = 160, 96
    sobol, tile = blue_noise_tables()
    chain = api.Chain(0, sobol, tile)
    ibl_np = chain_util.make_ibl(lib, pfx)
    ibl = api.IBLResources(torch.from_numpy(ibl_np["lut"]).to(chain.device), [torch.from_numpy(m).to(chain.device) for m in ibl_np["irradiance"]],
                           [torch.from_numpy(m).to(chain.device) for m in ibl_np["prefiltered"]])
    sa = chain_util.shade_attribs(len(ibl_np["prefiltered"]) - 1)
    f = synth.make_frame(synth.Scene(), 3, w, h, chain.device)
    a, b = torch.zeros(h, w, 4, device=chain.device), torch.zeros(h, w, 4, device=chain.device)
    chain.execute(chain.bind_frame(0, f, ibl, sa, a))
    mifx_lib.mifx_set_markers(1)
    try:
        chain.reset_history()
        chain.execute(chain.bind_frame(0, f, ibl, sa, b))
        assert any("roctx" in line for line in open("/proc/self/maps")), "libroctx64 was not loaded"
    finally:
        mifx_lib.mifx_set_markers(0)
    assert torch.equal(a, b)
    chain.close()


LANE_EDGES = "ssao_compute_ao_kernel<ssr_intersection_kernel@1,ssao_temporal_kernel<ssr_temporal_kernel@1,bloom_prefilter_kernel<ssr_spatial_kernel@0,taa_kernel<pbr_shade_ssr_mask_kernel@0"


@pytest.mark.parametrize("mode", [1, 2, 3, 4, 5, "4 + edges", "4 at 1920x1080", "5 at 1920x1080", "3 at 3840x2160", "4 at 3840x2160", "5 at 3840x2160"])
def test_chain_stream_overlap_is_bit_identical(mifx_lib, mode):
    """mifx_chain_set_overlap: prep + SSAO on the second stream (1), across frames (2: several frames are queued without a synchronisation in between, so that
    the next frame's prep + SSAO really run beside the previous frame's Bloom), the three lanes of mode 3 (shade + prep + Hi-Z + SSAO | SSR + composite + TAA | Bloom),
    mode 5 (mode 4 with the composite, TAA and depth of field on the Bloom lane: the next frame's ray march beside them),
    and mode 4 -- those lanes with two frames in flight, the planes between lane S and lane X alternating between two sets (also with mifx_chain_set_lane_edges, and at
    a size whose kernels outlast the host's launches so that the frames really overlap): the frames and the histories equal the one-stream chain's bit for bit.
    "3 / 4 at 3840x2160": the modes and the size bench.py's headline number ran in until / runs in since the second session of round 6 (bench.py repeats this check on its own
    orbit after every timed region: overlap_verified)."""
    import chain_util
    from diligentfx_amd import api, synth

    w, h = (1920, 1080) if str(mode).endswith("at 1920x1080") else (3840, 2160) if str(mode).endswith("at 3840x2160") else (640, 360)
    sobol, tile = blue_noise_tables()
    plain, over = api.Chain(0, sobol, tile), api.Chain(0, sobol, tile)
    over.set_overlap(int(str(mode)[0]))
    if mode == "4 + edges":
        over.set_lane_edges(LANE_EDGES)
    ibl = api.precompute_ibl(plain.postfx, synth.make_sky_cube(32, plain.device).clamp(max=200.0), lut_size=32, irradiance_size=8, prefiltered_size=32, lut_samples=32,
                             diffuse_samples=32, specular_samples=16)
    sa = chain_util.shade_attribs(len(ibl.pre) - 1)
    scene = synth.Scene()
    frames = [synth.make_frame(scene, i, w, h, plain.device) for i in range(8)]
    want = [torch.zeros(h, w, 4, device=plain.device) for _ in frames]
    got = [torch.zeros(h, w, 4, device=plain.device) for _ in frames]
    torch.cuda.synchronize()  # (mode 2: the inputs of every frame are complete before the first execute)
    for i, f in enumerate(frames):
        plain.execute(plain.bind_frame(i, f, ibl, sa, want[i]))
    for i, f in enumerate(frames):
        over.execute(over.bind_frame(i, f, ibl, sa, got[i]))
    torch.cuda.synchronize()
    for i in range(len(frames)):
        assert torch.equal(got[i], want[i]), (mode, i, int((got[i] != want[i]).sum()))
    for name in ("history_ao", "history_len"):
        assert torch.equal(over.effect("ssao").get_intermediate(name), plain.effect("ssao").get_intermediate(name)), name
    for name in ("hist_radiance", "hist_variance"):
        assert torch.equal(over.effect("ssr").get_intermediate(name), plain.effect("ssr").get_intermediate(name)), name
    over.close()
    plain.close()


@pytest.mark.parametrize("deep", [4, 5])
def test_chain_pipelined_frames_with_index_gaps_and_resizes(mifx_lib, deep):
    """Modes 4 and 5 keep two frames in flight only while FrameDesc.Index advances by one (the histories ping-pong by its parity); repeated and skipped indices, a change of the
    frame size and a mode switch in the middle of the run leave the frames equal to the one-stream chain's."""
    import chain_util
    from diligentfx_amd import api, synth

    sobol, tile = blue_noise_tables()
    plain, over = api.Chain(0, sobol, tile), api.Chain(0, sobol, tile)
    over.set_overlap(deep)
    ibl = api.precompute_ibl(plain.postfx, synth.make_sky_cube(32, plain.device).clamp(max=200.0), lut_size=32, irradiance_size=8, prefiltered_size=32, lut_samples=32,
                             diffuse_samples=32, specular_samples=16)
    sa = chain_util.shade_attribs(len(ibl.pre) - 1)
    scene = synth.Scene()
    #        (frame index, width, height, mode of the second chain from this frame on; 4 stands for `deep`)
    plan = [(0, 640, 360, 4), (1, 640, 360, 4), (2, 640, 360, 4), (4, 640, 360, 4), (5, 640, 360, 4), (5, 640, 360, 4), (6, 640, 360, 4), (7, 320, 200, 4), (8, 320, 200, 4),
            (9, 320, 200, 4), (10, 640, 360, 4), (11, 640, 360, 3), (12, 640, 360, 3), (13, 640, 360, 4), (14, 640, 360, 4), (15, 640, 360, 4), (17, 640, 360, 4), (18, 640, 360, 4)]
    frames = [synth.make_frame(scene, i, w, h, plain.device) for i, w, h, _ in plan]
    want = [torch.zeros(h, w, 4, device=plain.device) for _, w, h, _ in plan]
    got = [torch.zeros(h, w, 4, device=plain.device) for _, w, h, _ in plan]
    torch.cuda.synchronize()
    for n, (i, w, h, _) in enumerate(plan):
        plain.execute(plain.bind_frame(i, frames[n], ibl, sa, want[n]))
    for n, (i, w, h, m) in enumerate(plan):
        over.set_overlap(deep if m == 4 else m)
        over.execute(over.bind_frame(i, frames[n], ibl, sa, got[n]))
    torch.cuda.synchronize()
    for n in range(len(plan)):
        assert torch.equal(got[n], want[n]), (plan[n], int((got[n] != want[n]).sum()))
    over.close()
    plain.close()


@pytest.mark.parametrize("mode", [2, 3, 4, 5])
def test_chain_overlap_orders_history_fills(mifx_lib, mode):
    """The cross-frame modes let the next frame's lanes wait for events of the previous frame only.  Work the library itself queues on the context stream between two
    frames -- the history fills of mifx_chain_reset_history, a history import, depth of field switched on (a re-allocating prepare) -- must still be ordered in front of
    them (mifx_postfx::stream_epoch): frames queued back to back around such calls equal the one-stream chain's."""
    import chain_util
    from diligentfx_amd import api, synth

    w, h = 640, 360
    sobol, tile = blue_noise_tables()
    plain, over = api.Chain(0, sobol, tile), api.Chain(0, sobol, tile)
    over.set_overlap(mode)
    ibl = api.precompute_ibl(plain.postfx, synth.make_sky_cube(32, plain.device).clamp(max=200.0), lut_size=32, irradiance_size=8, prefiltered_size=32, lut_samples=32,
                             diffuse_samples=32, specular_samples=16)
    sa = chain_util.shade_attribs(len(ibl.pre) - 1)
    scene = synth.Scene()
    frames = [synth.make_frame(scene, i, w, h, plain.device) for i in range(9)]
    outs = {c: [torch.zeros(h, w, 4, device=plain.device) for _ in frames] for c in (plain, over)}
    torch.cuda.synchronize()
    for c in (plain, over):
        saved = None
        for i, f in enumerate(frames):
            if i == 3:
                c.reset_history()
            if i == 5:  # the SSAO history of frame 4 written back into the object: the copy is queued on the context stream, the next frame's SSAO runs on a lane
                ao, ln, idx = c.effect("ssao").export_history()
                c.effect("ssao").import_history(ao, ln, idx)
            if i == 7:
                c.reset_history()
            c.execute(c.bind_frame(i, f, ibl, sa, outs[c][i]))
    torch.cuda.synchronize()
    for i in range(len(frames)):
        assert torch.equal(outs[over][i], outs[plain][i]), (mode, i, int((outs[over][i] != outs[plain][i]).sum()))
    for name in ("history_ao", "history_len"):
        assert torch.equal(over.effect("ssao").get_intermediate(name), plain.effect("ssao").get_intermediate(name)), name
    over.close()
    plain.close()


@pytest.mark.parametrize("mode", [0, 5, "auto exposure"])
def test_chain_reset_history_with_depth_of_field_equals_a_fresh_chain(mifx_lib, mode):
    """mifx_chain_reset_history: a chain with a past continues like a fresh one -- also with depth of field, whose temporal circle of confusion the reference never resets
    (it is cleared when the targets are created, DepthOfField.cpp:205-223; mifx_dof_reset_history clears it the same way).  Found by `bench.py --dof`, whose
    overlap_verified compares the run's chain after a reset with a fresh one: until the second session of round 6 every frame differed, in every stream mode."""
    import chain_util
    from diligentfx_amd import api, binding as B, synth

    w, h = 640, 360
    sobol, tile = blue_noise_tables()
    old, fresh = api.Chain(0, sobol, tile), api.Chain(0, sobol, tile)
    ibl = api.precompute_ibl(old.postfx, synth.make_sky_cube(32, old.device).clamp(max=200.0), lut_size=32, irradiance_size=8, prefiltered_size=32, lut_samples=32,
                             diffuse_samples=32, specular_samples=16)
    sa = chain_util.shade_attribs(len(ibl.pre) - 1)
    scene = synth.Scene()
    ae = mode == "auto exposure"  # (the adapted average luminance is such a state as well: mifx_chain_reset_history puts it back at the reference's start value)
    mode = 3 if ae else mode
    for c in (old, fresh):
        c.set_depth_of_field(B.DOFAttribs.default(), api.DepthOfField.FEATURE_FLAG_ENABLE_TEMPORAL_SMOOTHING)
        if ae:
            c.set_auto_exposure(True, elapsed_time_s=0.25)
    frames = [synth.make_frame(scene, 16 + i, w, h, old.device) for i in range(10)]
    for f in frames:
        f["camera"].fFocusDistance, f["camera"].fFStop, f["camera"].fFocalLength = 12.0, 1.2, 135.0
    scratch = torch.zeros(h, w, 4, device=old.device)
    for i in range(6):  # the past of the first chain
        old.execute(old.bind_frame(16 + i, frames[i], ibl, sa, scratch))
    torch.cuda.synchronize()
    old.set_overlap(mode)
    old.reset_history()
    fresh.reset_history()
    got = [torch.zeros(h, w, 4, device=old.device) for _ in range(4)]
    want = [torch.zeros(h, w, 4, device=old.device) for _ in range(4)]
    for i in range(4):
        old.execute(old.bind_frame(1000 + i, frames[6 + i], ibl, sa, got[i]))
    for i in range(4):
        fresh.execute(fresh.bind_frame(1000 + i, frames[6 + i], ibl, sa, want[i]))
    torch.cuda.synchronize()
    for i in range(4):
        assert torch.equal(got[i], want[i]), (mode, i, int((got[i] != want[i]).sum()))
    old.close()
    fresh.close()
