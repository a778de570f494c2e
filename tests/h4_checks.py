"""Run in its own process with MIFX_STORAGE=h4 (tests/test_gpu_storage_h4.py does): the native-storage build of the library (libmifx_h4.so: the reference's own target
formats -- RGBA16_FLOAT colour planes, R8_UNORM ambient occlusion and roughness, R16_FLOAT variance / resolved depth / history length, RG16_FLOAT closest motion,
R11G11B10_FLOAT Bloom) against the checker with format emulation -- every image a reference pass writes goes through the rounding of its target format when it is stored
(oracle/pyref.py QuantizingLib), the inputs are the binary16 values the HIP side is given.  Prints what it measured; exits non-zero on the first violated bound."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import chain_util  # noqa: E402
import cpu_chain  # noqa: E402
import pyref  # noqa: E402
from diligentfx_amd import api, binding as B, synth  # noqa: E402
from util import assert_close, blue_noise_tables, to_np  # noqa: E402

MEASURE = bool(os.environ.get("MIFX_PARITY_MEASURE"))
# binary16 has 11 significant bits: one rounding step is 4.9e-4 relative, two values that agree to 1e-3 before the store can land two steps apart after it
RTOL = 2.5e-3
# R8_UNORM: values that agree to 1e-3 before the store can land one code apart (1 / 255)
AO_STEP = 1.02 / 255.0
# R11G11B10_FLOAT (Bloom levels and output, hence the final image): 6 / 6 / 5 mantissa bits -- one rounding step is 1.6 % (red, green) or 3.1 % (blue) of the value
RTOL_BLOOM = 3.3e-2


def q16(a):
    with np.errstate(over="ignore"):
        return a.astype(np.float16).astype(np.float32)


def f32(t):
    return to_np(t.float())


def bloom_rgba(t):
    """Bloom's output plane of the native-storage build: packed R11G11B10_FLOAT texels (torch.int32 view) -> float32 (H, W, 4); the format has no alpha: reads as 1."""
    assert t.dtype == torch.int32, t.dtype
    rgb = to_np(api.widen(t))
    return np.concatenate([rgb, np.ones(rgb.shape[:2] + (1,), np.float32)], axis=-1)


def setup():
    lib = B.load()
    assert lib.mifx_storage_mode() == 1 and B.storage_dtype() == torch.float16, "not the RGBA16_FLOAT storage build"
    plain = pyref.ref_lib() or pyref.oracle_lib()
    pfx = "ref_" if pyref.ref_lib() is not None else "oracle_"
    quant = pyref.QuantizingLib(plain)
    w, h = 224, 128
    sobol, tile = blue_noise_tables()
    chain = api.Chain(0, sobol, tile)
    dev = chain.device
    ibl_np = chain_util.make_ibl(plain, pfx)  # cube maps and the LUT stay fp32 in both builds
    ibl = api.IBLResources(torch.from_numpy(ibl_np["lut"]).to(dev), [torch.from_numpy(m).to(dev) for m in ibl_np["irradiance"]], [torch.from_numpy(m).to(dev) for m in ibl_np["prefiltered"]])
    cpu = cpu_chain.CpuChain(quant, pfx)
    sa = chain_util.shade_attribs(len(ibl_np["prefiltered"]) - 1)
    scene = synth.Scene()
    out = torch.zeros(h, w, 4, device=dev, dtype=torch.float16)
    return dict(lib=lib, plain=plain, pfx=pfx, quant=quant, w=w, h=h, sobol=sobol, tile=tile, chain=chain, dev=dev, ibl_np=ibl_np, ibl=ibl, cpu=cpu, sa=sa, scene=scene, out=out)


def section_chain(S):
    lib, pfx, quant, w, h, chain, dev, ibl_np, ibl, cpu, sa, scene, out = (S[k] for k in ("lib", "pfx", "quant", "w", "h", "chain", "dev", "ibl_np", "ibl", "cpu", "sa", "scene", "out"))
    # 1. a 4-channel float32 image is refused, loudly
    f = synth.make_frame(scene, 0, w, h, dev)
    i32, o16 = B.image(f["base_color"]), B.image(out)
    import ctypes

    tm = B.ToneMappingAttribs.default(4)
    st = lib.mifx_tonemap_execute(chain.postfx.handle, ctypes.byref(i32), ctypes.byref(o16), ctypes.byref(tm), ctypes.c_float(0.3), ctypes.c_uint32(1))
    assert st == -1 and b"F16X4" in lib.mifx_last_error(), (st, lib.mifx_last_error())

    # 2. the chain, frame by frame, against the checker with RGBA16_FLOAT stores
    # outlier budgets: round 2 measured radiance 6e-5, SSR 5.8e-3, TAA 3.2e-4, 2.3e-3 of the Bloom / final values not on the same code (profiles/r02_h4_parity.txt) with the
    # contracting build; the build without contraction (round 4) has 0 everywhere over these six frames except <= 1.8e-5 of the values not on the same R11G11B10 code
    # (profiles/r04_h4_parity.txt) -- the budgets below are a small multiple of that; MIFX_PARITY_MEASURE=1 reports without deciding
    budget = dict.fromkeys(("radiance", "ssr", "ssao", "taa", "bloom", "final"), 1.0) if MEASURE else {"radiance": 5e-5, "ssr": 1e-3, "ssao": 2e-4, "taa": 1e-4, "bloom": 1e-4, "final": 1e-4}
    budget_exact = dict.fromkeys(("bloom", "final"), 1.0) if MEASURE else {"bloom": 5e-4, "final": 5e-4}  # values that did not land on the same R11G11B10 code
    for frame in range(6):
        f = synth.make_frame(scene, frame, w, h, dev)
        chain.execute(chain.bind_frame(frame, f, ibl, sa, out))
        g = {k: to_np(v) for k, v in f.items() if isinstance(v, torch.Tensor)}
        for k in ("base_color", "normal", "material"):
            g[k] = q16(g[k])  # the checker reads the binary16 values the HIP side was given
        keep = {}
        want = chain_util.run_frame_inputs(cpu, g, bytes(f["camera"]), bytes(f["prev_camera"]), frame, ibl_np, sa, keep)
        got = f32(out)
        assert np.isfinite(got).all() and out.dtype == torch.float16
        res = {}
        _, res["radiance"] = assert_close(f32(chain.shard_plane_image("radiance")), keep["radiance"], rtol=RTOL, max_outlier_frac=budget["radiance"], what=f"radiance frame {frame}")
        _, res["ssr"] = assert_close(f32(chain.effect_output("ssr")), keep["ssr_out"], rtol=RTOL, max_outlier_frac=budget["ssr"], what=f"SSR frame {frame}")
        _, res["ssao"] = assert_close(to_np(api.widen(chain.effect_output("ssao"))), keep["ssao_out"], max_outlier_frac=budget["ssao"], abs_slack=AO_STEP, what=f"SSAO frame {frame}")
        _, res["taa"] = assert_close(f32(chain.effect_output("taa")), keep["taa_out"], rtol=RTOL, max_outlier_frac=budget["taa"], what=f"TAA frame {frame}")
        # Bloom's levels and output are R11G11B10_FLOAT: the bound is one rounding step of the format; how many values landed on the very same code is reported beside it
        _, res["bloom"] = assert_close(bloom_rgba(chain.effect_output("bloom")), keep["bloom_out"], rtol=RTOL_BLOOM, max_outlier_frac=budget["bloom"], what=f"Bloom frame {frame}")
        _, res["final"] = assert_close(got, want, rtol=RTOL_BLOOM, max_outlier_frac=budget["final"], what=f"final image frame {frame}")
        _, res["bloom_same_code"] = assert_close(bloom_rgba(chain.effect_output("bloom")), keep["bloom_out"], rtol=RTOL, max_outlier_frac=budget_exact["bloom"], what=f"Bloom frame {frame} (same code)")
        _, res["final_same_code"] = assert_close(got, want, rtol=RTOL, max_outlier_frac=budget_exact["final"], what=f"final image frame {frame} (same code)")
        assert np.abs(got[..., :3] - want[..., :3]).mean() < 2e-3
        print(f"h4 chain frame {frame}: outlier fractions " + " ".join(f"{k} {v:.2e}" for k, v in res.items()), flush=True)
    assert chain.effect_output("ssr").dtype == torch.float16 and chain.effect_output("ssao").dtype == torch.uint8 and chain.effect_output("bloom").dtype == torch.int32
    # the Bloom output is an R11G11B10_FLOAT plane (4 bytes per texel, Bloom.cpp:137); alpha reads as 1
    bo = bloom_rgba(chain.effect_output("bloom"))
    assert np.array_equal(pyref.store_r11g11b10(bo, alpha_reads_as=1.0), bo)
    # ... and the tone map takes it as it is: the stand-alone pass on the plane equals the tone map fused into Bloom's last pass
    import ctypes as C

    unfused = torch.zeros(h, w, 4, device=dev, dtype=torch.float16)
    bdesc, odesc = B.image(chain.effect_output("bloom")), B.image(unfused)
    f5 = synth.make_frame(scene, 5, w, h, dev)
    bound = chain.bind_frame(5, f5, ibl, sa, out)
    B.check(lib.mifx_tonemap_execute(chain.postfx.handle, C.byref(bdesc), C.byref(odesc), bound[0].tone_mapping, C.c_float(bound[0].ave_log_lum), C.c_uint32(bound[0].tonemap_flags)))
    torch.cuda.synchronize()
    assert torch.equal(unfused, out), f"tone map on the packed Bloom output vs the fused pass: {int((unfused != out).sum())} values differ"
    # ... and so does the auto exposure: the packed plane and an RGBA16_FLOAT copy of its values give the same average and the same tone-mapped frame
    packed = chain.effect_output("bloom")
    as_half = torch.from_numpy(bo).to(dev).half()  # (R11G11B10 values are exact in binary16)
    assert np.array_equal(f32(as_half), bo)
    ae_p, ae_h = api.AutoExposure(chain.postfx), api.AutoExposure(chain.postfx)
    tm = B.ToneMappingAttribs.default(4)
    ldr_p, ldr_h = torch.zeros_like(unfused), torch.zeros_like(unfused)
    for ae, img, ldr in ((ae_p, packed, ldr_p), (ae_h, as_half, ldr_h)):
        ae.reset(0.1)
        ae.execute(img, 0.5)
        ae.tone_map(img, tm, 1, out=ldr)
    torch.cuda.synchronize()
    assert ae_p.average() == ae_h.average() and torch.equal(ldr_p, ldr_h), (ae_p.average(), ae_h.average())
    assert torch.equal(ae_p.plane("low_res_luminance"), ae_h.plane("low_res_luminance"))
    ae_p.close()
    ae_h.close()
    ao, hl, idx_ao = chain.effect("ssao").export_history()
    assert ao.dtype == torch.uint8 and hl.dtype == torch.float16
    chain.effect("ssao").import_history(ao, hl, idx_ao)
    # 3. a stored value is exactly representable: storing it again does not change it
    assert torch.equal(out, out.float().half())
    # 4. history export / import carry the binary16 planes
    col, idx = chain.effect("taa").export_history()
    assert col.dtype == torch.float16 and idx == 5
    chain.effect("taa").import_history(col, idx)


def section_fusion(S):
    w, h, sobol, tile, dev, ibl, sa, scene = (S[k] for k in ("w", "h", "sobol", "tile", "dev", "ibl", "sa", "scene"))
    # 4b. every fusion switch of the chain gives the same bits in this build too (the narrow stores round identically in every translation unit)
    fused, plain = api.Chain(0, sobol, tile), api.Chain(0, sobol, tile)
    plain.set_fusion_mask(0)
    oa, ob = torch.zeros(h, w, 4, device=dev, dtype=torch.float16), torch.zeros(h, w, 4, device=dev, dtype=torch.float16)
    for frame in range(4):
        f = synth.make_frame(scene, frame, w, h, dev)
        fused.execute(fused.bind_frame(frame, f, ibl, sa, oa))
        plain.execute(plain.bind_frame(frame, f, ibl, sa, ob))
        assert torch.equal(oa, ob), f"h4 fusion on / off: frame {frame}: {int((oa != ob).sum())} values differ"
        for name in ("roughness", "mask", "hist_radiance"):
            assert torch.equal(fused.effect("ssr").get_intermediate(name), plain.effect("ssr").get_intermediate(name)), name
        assert torch.equal(fused.effect_output("ssao"), plain.effect_output("ssao"))
    fused.close()
    plain.close()
    print("h4 fusion on / off: 4 frames bit-identical", flush=True)


def section_dof(S):
    pfx, quant, sobol, tile, dev, scene = (S[k] for k in ("pfx", "quant", "sobol", "tile", "dev", "scene"))
    # 5. depth of field (its eleven passes store five 4-channel targets) end to end against the format-emulating checker
    import test_gpu_dof as D

    ctx = api.PostFXContext(0, sobol, tile)
    dof = api.DepthOfField(ctx)
    attribs = B.DOFAttribs.default()
    attribs.MaxCircleOfConfusion, attribs.AlphaInterpolation = 0.02, 0.9
    e2e = cpu_chain.CpuChain(quant, pfx)
    w2, h2 = 256, 144
    for frame in (7, 8):
        f = synth.make_frame(scene, frame, w2, h2, dev)
        cam = D.lens_camera(f["camera"])
        color = D.hdr_colour(f, dev)
        ctx.prepare_resources(frame, w2, h2)
        dof.prepare_resources(1)
        ctx.execute(f["depth"], f["prev_depth"], f["motion"], cam, f["prev_camera"])
        dof.execute(B.to_storage(color), f["depth"], attribs)
        pf = {"frame": frame, "cam": bytes(cam), "closest_motion": f32(ctx.get_closest_motion_vectors())}
        want = e2e.dof(pf, q16(to_np(color)), to_np(f["depth"]), attribs, 1)
        got = bloom_rgba(dof.get_depth_of_field_texture())  # (an R11G11B10_FLOAT target, DepthOfField.cpp:281-289: a 4-byte plane since round 5)
        _, frac = assert_close(got, want, rtol=RTOL, max_outlier_frac=1.0 if MEASURE else 2e-4, what=f"depth of field frame {frame}")  # measured <= 2.7e-5 (profiles/r04_h4_parity.txt)
        print(f"h4 depth of field frame {frame}: outlier fraction {frac:.2e}", flush=True)
        assert dof.get_depth_of_field_texture().dtype == torch.int32
    dof.close()
    ctx.close()


def section_dof_chain(S):
    """Depth of field between TAA and Bloom in the native-storage build: its output is a 4-byte R11G11B10_FLOAT plane (round 5) that Bloom's prefilter and final pass read
    as such.  (a) the chain with depth of field against a second chain + stand-alone DOF + stand-alone Bloom on the packed plane (tests/test_gpu_dof.py: the chain's fused
    final pass and the stand-alone passes take the packed source); (b) Bloom on the packed plane == Bloom on an RGBA16_FLOAT plane holding the same values (every R11G11B10
    value is a binary16 value): the packed load is the only difference, the pyramid and the output are equal code for code."""
    sobol, tile, dev, scene, ibl, sa = (S[k] for k in ("sobol", "tile", "dev", "scene", "ibl", "sa"))
    import test_gpu_dof as D

    W, H = 320, 192
    a, b = api.Chain(0, sobol, tile), api.Chain(0, sobol, tile)
    da = B.DOFAttribs.default()
    da.MaxCircleOfConfusion = 0.02
    flags = api.DepthOfField.FEATURE_FLAG_ENABLE_TEMPORAL_SMOOTHING
    a.set_depth_of_field(da, flags)
    out_a, out_b = (torch.zeros(H, W, 4, device=dev, dtype=torch.float16) for _ in range(2))
    standalone = api.DepthOfField(b.postfx)
    for fi in range(3, 6):
        g = synth.make_frame(scene, fi, W, H, dev)
        D.lens_camera(g["camera"])
        a.execute(a.bind_frame(fi, g, ibl, sa, out_a))
        b.execute(b.bind_frame(fi, g, ibl, sa, out_b))
        torch.cuda.synchronize()
        taa_b = b.effect_output("taa")
        assert torch.equal(a.effect_output("taa"), taa_b)
        standalone.prepare_resources(flags)
        standalone.execute(taa_b, g["depth"], da)
        assert standalone.get_depth_of_field_texture().dtype == torch.int32 and torch.equal(standalone.get_depth_of_field_texture(), a.effect_output("dof"))
        bl = api.Bloom(b.postfx)
        bl.prepare_resources()
        bl.execute(standalone.get_depth_of_field_texture(), b.bloom_attribs)  # (the prefilter and the final up-sample on the packed plane)
        assert torch.equal(bl.get_bloom_texture(), a.effect_output("bloom"))  # (the chain's: produced on demand from the packed plane it kept)
        ldr = b.postfx.tone_map(bl.get_bloom_texture(), b.tone_mapping, b.ave_log_lum, b.tonemap_flags, out=torch.zeros_like(out_a))
        assert torch.equal(ldr, out_a), f"frame {fi}: the chain's fused final pass on the packed depth-of-field output differs from Bloom + ToneMap() on it"
        bl.close()
        assert not torch.equal(out_a, out_b)
    standalone.close()
    a.close()
    b.close()
    ctx = api.PostFXContext(0, sobol, tile)
    dof, bloom = api.DepthOfField(ctx), api.Bloom(ctx)
    attribs = B.DOFAttribs.default()
    attribs.MaxCircleOfConfusion = 0.02
    ba = B.BloomAttribs.default()
    w2, h2 = 320, 192
    for frame in (7, 8):
        f = synth.make_frame(scene, frame, w2, h2, dev)
        cam = D.lens_camera(f["camera"])
        ctx.prepare_resources(frame, w2, h2)
        dof.prepare_resources(1)
        bloom.prepare_resources()
        ctx.execute(f["depth"], f["prev_depth"], f["motion"], cam, f["prev_camera"])
        dof.execute(B.to_storage(D.hdr_colour(f, dev)), f["depth"], attribs)
        packed = dof.get_depth_of_field_texture()
        assert packed.dtype == torch.int32
        bloom.execute(packed, ba)
        torch.cuda.synchronize()
        from_packed = {"out": bloom.get_bloom_texture().clone(), "down0": bloom.get_intermediate("down0").clone(), "up0": bloom.get_intermediate("up0").clone()}
        unpacked = B.to_storage(torch.from_numpy(bloom_rgba(packed)).to(dev))
        assert unpacked.dtype == torch.float16 and np.array_equal(to_np(unpacked.float())[..., :3], bloom_rgba(packed)[..., :3]), "an R11G11B10 value is a binary16 value"
        bloom.execute(unpacked, ba)
        torch.cuda.synchronize()
        assert torch.equal(bloom.get_intermediate("down0"), from_packed["down0"]) and torch.equal(bloom.get_intermediate("up0"), from_packed["up0"]), frame
        assert torch.equal(bloom.get_bloom_texture(), from_packed["out"]), f"frame {frame}: Bloom on the packed depth-of-field output differs from Bloom on the same values in RGBA16_FLOAT"
    print("h4 depth of field -> Bloom: the chain equals the stand-alone effects on the packed plane; packed and RGBA16_FLOAT sources give the same pyramid and output", flush=True)
    bloom.close()
    dof.close()
    ctx.close()


def section_dof_passes(S):
    """Depth of field pass by pass in the reference's target formats: the circle of confusion and its history R16_FLOAT, the separated / dilated / blurred one
    R16_UNORM, the colour targets RGBA16_FLOAT, the combined output R11G11B10_FLOAT values with alpha 1.  Every pass of the HIP side runs on the HIP side's own previous
    planes; the checker's pass runs on the same (widened) planes and stores through QuantizingLib."""
    import test_gpu_dof as D

    pfx, quant, plain, sobol, tile, dev, scene = (S[k] for k in ("pfx", "quant", "plain", "sobol", "tile", "dev", "scene"))
    ctx = api.PostFXContext(0, sobol, tile)
    dof = api.DepthOfField(ctx)
    U16 = 1.02 / 65535.0
    tables = cpu_chain.CpuChain(plain, pfx)
    for (w, h), flags, rings in (((256, 144), 3, (5, 7)), ((202, 118), 1, (3, 4)), ((100, 60), 0, (4, 6))):
        attribs = B.DOFAttribs.default()
        attribs.MaxCircleOfConfusion, attribs.AlphaInterpolation = 0.02, 0.9
        attribs.BokehKernelRingCount, attribs.BokehKernelRingDensity = rings
        temporal = bool(flags & 1)
        prev_temporal = np.zeros((h, w), np.float32)
        large, small, gauss = tables.dof_tables(*rings)
        worst = {}
        for frame in (7, 8, 9):
            f = synth.make_frame(scene, frame, w, h, dev)
            cam = D.lens_camera(f["camera"])
            color = B.to_storage(D.hdr_colour(f, dev))
            ctx.prepare_resources(frame, w, h)
            dof.prepare_resources(flags)
            ctx.execute(f["depth"], f["prev_depth"], f["motion"], cam, f["prev_camera"])
            motion = f32(ctx.get_closest_motion_vectors())
            names = ["coc", "dilation1", "dilation2", "dilation3", "dilation_blurred", "prefiltered0", "prefiltered1", "bokeh0", "bokeh1"] + (["coc_temporal"] if temporal else [])

            def read():
                return {n: to_np(api.widen(dof.get_intermediate(n))).copy() for n in names}

            dof.debug_set_last_pass(7)
            dof.execute(color, f["depth"], attribs)
            first = read()
            dof.debug_set_last_pass(0)
            dof.execute(color, f["depth"], attribs)
            torch.cuda.synchronize()
            second = read()
            got = bloom_rgba(dof.get_depth_of_field_texture())
            assert dof.get_depth_of_field_texture().dtype == torch.int32 and dof.get_intermediate("coc").dtype == torch.float16 and dof.get_intermediate("dilation3").dtype == torch.int16 and dof.get_intermediate("bokeh0").dtype == torch.float16
            cnp, dnp = f32(color), to_np(f["depth"])
            P = D.Passes(quant, pfx, bytes(cam), attribs, flags)

            def cmp(name, a, b, rtol=RTOL, slack=None, frac=0.0):
                worst[name] = max(worst.get(name, 0.0), assert_close(a, b, rtol=rtol, abs_slack=slack, max_outlier_frac=1.0 if MEASURE else frac, what=f"h4 DOF {name} frame {frame} {w}x{h}")[1])

            cmp("coc", first["coc"], P.coc(dnp))
            used = first["coc"]
            if temporal:
                cmp("coc_temporal", first["coc_temporal"], P.temporal(first["coc"], prev_temporal, motion))
                used = prev_temporal = first["coc_temporal"]
            lvl = P.separated(used)
            for k in (1, 2, 3):
                assert np.array_equal(first[f"dilation{k}"], P.dilation(lvl)), f"dilation{k}: a maximum of R16_UNORM values is bit-exact"
                lvl = first[f"dilation{k}"]
            cmp("dilation_blurred", first["dilation_blurred"], P.blur(first["dilation3"], gauss), slack=2.0 * U16)  # two stores: the horizontal pass and the vertical one
            n6, f6 = P.prefilter(cnp, used, first["dilation_blurred"])
            cmp("prefiltered near", first["prefiltered0"], n6)
            cmp("prefiltered far", first["prefiltered1"], f6)
            n7, f7 = P.bokeh_first(first["prefiltered0"], first["prefiltered1"], large, cnp)
            cmp("bokeh gather near", first["bokeh0"], n7)
            cmp("bokeh gather far", first["bokeh1"], f7, frac=2e-3)  # "a >= CoCFar" on interpolated binary16 alphas: a tap that ties up to rounding may flip
            n8, f8 = P.bokeh_second(first["bokeh0"], first["bokeh1"], small)
            cmp("bokeh fill near", second["prefiltered0"], n8)
            cmp("bokeh fill far", second["prefiltered1"], f8, frac=2e-3)
            n9, f9 = P.postfilter(second["prefiltered0"], second["prefiltered1"])
            cmp("postfilter near", second["bokeh0"], n9)
            cmp("postfilter far", second["bokeh1"], f9)
            want = P.combine(cnp, used, second["bokeh0"], second["bokeh1"])
            cmp("combined", got, want, rtol=RTOL_BLOOM)
            cmp("combined (same code)", got, want, frac=6e-3)
            assert np.array_equal(pyref.store_r11g11b10(got, alpha_reads_as=1.0), got), "the combined output holds R11G11B10 values, alpha 1"
        print(f"h4 DOF passes {w}x{h} flags {flags}: outlier fractions " + " ".join(f"{k} {v:.1e}" for k, v in worst.items() if v > 0.0) + " (others 0)", flush=True)
    dof.close()
    ctx.close()


def section_half_precision_depth(S):
    """FEATURE_FLAG_HALF_PRECISION_DEPTH of PostFXContext and ScreenSpaceAmbientOcclusion: the reference makes the reprojected / previous depth and SSAO's two
    depth pyramids R16_UNORM targets; the native-storage build gives those planes the values such targets keep.  PostFX decides the format when it creates the planes, i.e. on
    a change of the frame size, not of the flag (PostFXContext.cpp:246-247) -- the second step below checks that too."""
    pfx, plain, sobol, tile, dev, scene = (S[k] for k in ("pfx", "plain", "sobol", "tile", "dev", "scene"))
    quant = pyref.QuantizingLib(plain)
    ctx = api.PostFXContext(0, sobol, tile)
    ssao, taa = api.ScreenSpaceAmbientOcclusion(ctx), api.TemporalAntiAliasing(ctx)
    cpu = cpu_chain.CpuChain(quant, pfx, taa_flags=2)
    U16 = 1.02 / 65535.0
    # (frame, width, height, PostFX flags, planes hold R16_UNORM values?, SSAO flags)
    steps = [(0, 224, 128, 0, False, 0), (1, 224, 128, 2, False, 1), (2, 224, 128, 2, False, 1), (3, 208, 112, 2, True, 1), (4, 208, 112, 2, True, 1), (5, 208, 112, 2, True, 1),
             (6, 208, 112, 0, True, 0)]
    for frame, w, h, pflags, p16, sflags in steps:
        f = synth.make_frame(scene, frame, w, h, dev)
        color = B.to_storage((torch.from_numpy(np.random.default_rng(2000 + frame).random((h, w, 4)).astype(np.float32)) * 2.0).to(dev))
        sa, ta = B.SSAOAttribs.default(), B.TAAAttribs.default()
        ctx.prepare_resources(frame, w, h, feature_flags=pflags)
        ssao.prepare_resources(feature_flags=sflags)
        taa.prepare_resources(2)
        ctx.execute(f["depth"], f["prev_depth"], f["motion"], f["camera"], f["prev_camera"])
        ssao.execute(f["depth"], B.to_storage(f["normal"]), sa)
        taa.execute(color, ta)
        quant.depth16["postfx"], quant.depth16["ssao"] = p16, bool(sflags & 1)
        g = {k: to_np(f[k]) for k in ("depth", "prev_depth", "motion", "normal")}
        g["normal"] = q16(g["normal"])
        keep = {}
        pf = cpu.postfx(frame, g["depth"], g["prev_depth"], g["motion"], bytes(f["camera"]), bytes(f["prev_camera"]), (sobol, tile))
        want_ao = cpu.ssao(pf, g["depth"], g["normal"], sa, keep, half_precision_depth=bool(sflags & 1))
        want_taa = cpu.taa(pf, f32(color), ta, None)
        prevd, reproj = to_np(ctx.get_previous_depth()), to_np(ctx.get_reprojected_depth())
        assert np.array_equal(prevd, pyref.store_unorm16(g["prev_depth"]) if p16 else g["prev_depth"]), f"previous depth frame {frame}"
        assert not p16 or np.array_equal(pyref.store_unorm16(reproj), reproj), "the reprojected depth holds R16_UNORM values"
        res = {}
        _, res["reprojected depth"] = assert_close(reproj, pf["reproj_depth"], abs_slack=U16 if p16 else None, max_outlier_frac=0.0, what=f"reprojected depth frame {frame}")
        if sflags & 1:
            for k in range(1, 5):
                got = to_np(ssao.get_intermediate(f"prefiltered_depth{k}"))
                assert np.array_equal(pyref.store_unorm16(got), got), f"prefiltered depth level {k} holds R16_UNORM values"
                _, res[f"prefiltered {k}"] = assert_close(got, keep["ssao_prefiltered_depth"][k], abs_slack=U16, max_outlier_frac=0.0, what=f"prefiltered depth level {k} frame {frame}")
        _, res["ssao"] = assert_close(to_np(api.widen(ssao.get_ambient_occlusion())), want_ao, abs_slack=AO_STEP, max_outlier_frac=1.0 if MEASURE else 2e-3, what=f"SSAO frame {frame}")
        _, res["taa"] = assert_close(f32(taa.get_accumulated_frame()), want_taa, rtol=RTOL, max_outlier_frac=1.0 if MEASURE else 2e-3, what=f"TAA frame {frame}")
        print(f"h4 half-precision depth frame {frame} ({w}x{h}, PostFX planes R16 {p16}, SSAO flag {sflags}): outlier fractions " + " ".join(f"{k} {v:.1e}" for k, v in res.items()), flush=True)
    # the flag changes results: the same frame with and without it differs in the AO (a check that the path is live, not a bound)
    for fx in (ssao, taa):
        fx.close()
    ctx.close()


def section_sharded(S):
    sobol, tile, dev, ibl, sa, scene = (S[k] for k in ("sobol", "tile", "dev", "ibl", "sa", "scene"))
    # 6. the sharded frame on binary16 planes: two in-process ranks (mifx_comm_create_local_group), band for band bit-identical to the unsharded chain
    import threading

    w3, h3 = 256, 512
    ref = api.Chain(0, sobol, tile)
    chains = [api.Chain(0, sobol, tile) for _ in range(2)]
    comms = api.Comm.local_group(chains[0].postfx, 2)
    cuts = [0, 240, h3]
    frames = [synth.make_frame(scene, i, w3, h3, dev) for i in range(3)]
    mm = int(max(float(fr["motion"][..., 1].abs().max()) for fr in frames) * 0.5 * h3) + 2
    outs = [torch.zeros(h3, w3, 4, device=dev, dtype=torch.float16) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    want = torch.zeros(h3, w3, 4, device=dev, dtype=torch.float16)
    for r in range(2):
        chains[r].set_sharding(comms[r], cuts, mm)
    errors = []
    for i, fr in enumerate(frames):
        ref.execute(ref.bind_frame(i, fr, ibl, sa, want))
        torch.cuda.synchronize()

        def run(r):
            try:
                with torch.cuda.stream(streams[r]):
                    chains[r].execute_sharded(chains[r].bind_frame(i, fr, ibl, sa, outs[r]))
                streams[r].synchronize()
            except Exception as e:  # noqa: BLE001
                errors.append((r, repr(e)))

        ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(120)
        assert not errors, errors
        torch.cuda.synchronize()
        for r in range(2):
            assert torch.equal(outs[r][cuts[r]:cuts[r + 1]], want[cuts[r]:cuts[r + 1]]), f"h4 sharded frame {i}: band of rank {r} differs"
    for r in range(2):
        chains[r].set_sharding(None)
        comms[r].close()
        chains[r].close()
    ref.close()
    print("h4 sharded: 3 frames x 2 ranks bit-identical to the unsharded chain", flush=True)


def section_layers(S):
    """The shade with material layers in the native-storage build: the G-buffer's and the layers' 4-channel planes are RGBA16_FLOAT, the outputs too; the checker (the
    reference's permutation) reads the binary16 values and its output takes the store's rounding."""
    from layers_util import BACKGROUND, IOR, ROTATION, checker_result, make_case

    lib = pyref.ref_lib()
    if lib is None or not lib.has("ref_pbr_shade_layers_all_shadows3"):
        print("h4 section layers: no oracle/_ref with the layered permutations, skipped")
        return
    dev, ibl_np, ibl, chain = S["dev"], S["ibl_np"], S["ibl"], S["chain"]
    f, gn, sa, planes, albedo, charlie = make_case("all_shadows3", (224, 128), ibl_np, dev, shadowed=True)
    slices, infos = chain_util.make_shadow_inputs()
    for k in ("base_color", "normal", "material", "emissive"):
        gn[k] = q16(gn[k])
    planes = {k: (v if k == "transmission" else q16(v)) for k, v in planes.items()}
    g = {k: (B.to_storage(torch.from_numpy(v).to(dev)) if v.ndim == 3 else torch.from_numpy(v).to(dev)) for k, v in gn.items()}
    lp = {k: B.to_storage(torch.from_numpy(v).to(dev)) for k, v in planes.items() if k != "transmission"}
    lp["transmission"] = torch.from_numpy(np.ascontiguousarray(planes["transmission"][..., 0])).to(dev)
    lp["sheen_albedo_scaling_lut"], lp["preintegrated_charlie"] = torch.from_numpy(albedo).to(dev), torch.from_numpy(charlie).to(dev)
    sm = torch.from_numpy(np.stack(slices)).to(dev)
    rad, spec = api.pbr_shade_layers(chain.postfx, g, lp, 31, f["camera"], sa, ibl, background=BACKGROUND, iridescence_ior=IOR, anisotropy_rotation=ROTATION, shadows=(sm, infos, 3))
    assert rad.dtype == torch.float16 and spec.dtype == torch.float16
    wr, ws = checker_result(lib, "all_shadows3", True, f, gn, sa, planes, albedo, charlie, ibl_np, shadows=(slices, infos))
    budget = 1.0 if MEASURE else 5e-5
    _, a = assert_close(f32(rad), q16(wr), rtol=RTOL, max_outlier_frac=budget, what="layered shade, radiance (RGBA16_FLOAT)")
    _, b = assert_close(f32(spec), q16(ws), rtol=RTOL, max_outlier_frac=budget, what="layered shade, specular IBL (RGBA16_FLOAT)")
    print(f"h4 layers: outlier fractions radiance {a:.2e}, specular IBL {b:.2e}; {(f32(rad) == q16(wr)).mean():.4f} of the radiance values on the same binary16 code")


SECTIONS = {"chain": section_chain, "fusion": section_fusion, "dof": section_dof, "dof_chain": section_dof_chain, "dof_passes": section_dof_passes, "half_precision_depth": section_half_precision_depth, "sharded": section_sharded, "layers": section_layers}


def main():
    names = sys.argv[1:] or list(SECTIONS)
    S = setup()
    for n in names:
        SECTIONS[n](S)
        print(f"h4 section {n} OK", flush=True)
    S["chain"].close()
    print("h4 checks OK")


if __name__ == "__main__":
    main()
