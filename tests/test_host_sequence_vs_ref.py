"""The reference's HOST code, executed, against the hand restatement of its sequencing.

oracle/_ref/libmifx_refhost.so = PostFXContext, ScreenSpaceAmbientOcclusion, ScreenSpaceReflection, TemporalAntiAliasing, Bloom and DepthOfField compiled from the sources where they
lie under /root/reference/PostProcess against a recording DiligentCore stand-in (oracle/refhost/dg); oracle/refhost.py replays what they ask the device to do with the
reference's own shaders (oracle/_ref).  That is the reference's frame -- pass order, render-target clears, ping-pong by FrameDesc.Index & 1, reset on the first frame /
an index gap / on request, mip loops, the depth-buffer reflection mask, resource re-creation on a resize or a flag change, TAA's placeholder frame -- and
oracle/cpu_chain.py, the checker every GPU parity test of the product's host objects (csrc/api_*.cpp) runs against, must reproduce it BIT FOR BIT: both sides run the
same compiled shader code on the same inputs, so any difference is a difference of sequencing.  SURVEY 8a rows C0 / A0 / R0 / T0 / B0, 8f N1 (depth of field).

(tests/test_gpu_host_sequence.py runs the same scenarios through the C ABI on the GPU.)"""
import numpy as np
import pytest
import torch

import cpu_chain
import pyref
import refhost
from diligentfx_amd import binding as B, synth
from util import blue_noise_tables

pytestmark = pytest.mark.skipif(not refhost.available() or pyref.ref_lib() is None, reason="oracle/_ref (the compiled reference) is not built here")

ALGOS = ["gtao", "hbao", "vbao"]
# (frame index, width, height, ResetAccumulation requested)
PLAIN = [(0, 96, 64, 0), (1, 96, 64, 0), (2, 96, 64, 0), (3, 96, 64, 1), (4, 96, 64, 0), (7, 96, 64, 0), (8, 96, 64, 0), (9, 80, 48, 0), (10, 80, 48, 0), (11, 96, 64, 0)]
SHORT = [(0, 96, 64, 0), (1, 96, 64, 0), (2, 96, 64, 0)]
SCENARIOS = {
    "frames 0-2, reset, index gap, two resizes": dict(steps=PLAIN),
    "half-resolution SSAO and SSR": dict(steps=SHORT, ssao_flags=2, ssr_flags=2),
    "reversed depth": dict(steps=SHORT, postfx_flags=1),
    "previous-frame SSR, HBAO, TAA flag set 7": dict(steps=SHORT, ssr_flags=1, algo=1, taa_flags=7),
    "VBAO, TAA flag set 0": dict(steps=SHORT, algo=2, taa_flags=0),
    "half-precision depth (GTAO)": dict(steps=SHORT, ssao_flags=1),
    "odd size": dict(steps=[(5, 70, 36, 0), (6, 70, 36, 0)]),
}


def frame_inputs(scene, idx, w, h, reversed_depth):
    f = synth.make_frame(scene, idx, w, h, torch.device("cpu"), reversed_depth=reversed_depth)
    g = {k: v.numpy() for k, v in f.items() if isinstance(v, torch.Tensor)}
    color = np.random.default_rng(1000 + idx).random((h, w, 4)).astype(np.float32) * np.float32(2.0)  # (the scene colour SSR reflects and the frame TAA / Bloom take)
    return g, bytes(f["camera"]), bytes(f["prev_camera"]), color


def attribs(algo, reset, alpha):
    ssao, ssr, taa, bloom = B.SSAOAttribs.default(), B.SSRAttribs.default(), B.TAAAttribs.default(), B.BloomAttribs.default()
    ssao.Algorithm = algo
    ssao.ResetAccumulation = taa.ResetAccumulation = 1 if reset else 0
    ssao.AlphaInterpolation = ssr.AlphaInterpolation = bloom.AlphaInterpolation = alpha  # (what the effects' frame timers make of it: the harness drives the timer)
    return ssao, ssr, taa, bloom


def test_replay_table_matches_the_wrappers():
    """The variable -> input slot table of the replay equals the `ref_bind` lines of the oracle/_ref wrappers it calls."""
    assert refhost.Replayer.check_table() == []


def test_attribute_blocks_have_the_reference_sizes():
    host = refhost.RefHost(0)
    for name, t in (("CameraAttribs", B.CameraAttribs), ("ScreenSpaceAmbientOcclusionAttribs", B.SSAOAttribs), ("ScreenSpaceReflectionAttribs", B.SSRAttribs),
                    ("TemporalAntiAliasingAttribs", B.TAAAttribs), ("BloomAttribs", B.BloomAttribs), ("DepthOfFieldAttribs", B.DOFAttribs)):
        import ctypes

        assert host.sizeof(name) == ctypes.sizeof(t), name
    host.close()


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_cpu_chain_equals_the_executed_reference_host(name):
    sc = dict(ssao_flags=0, ssr_flags=0, taa_flags=2, postfx_flags=0, algo=0)
    sc.update(SCENARIOS[name])
    ref = pyref.ref_lib()
    host, rp = refhost.RefHost(15), refhost.Replayer(ref)
    rev = bool(sc["postfx_flags"] & 1)
    chain = cpu_chain.CpuChain(ref, "ref_", algorithm=ALGOS[sc["algo"]], taa_flags=sc["taa_flags"], reversed_depth=rev)
    scene = synth.Scene()
    for n, (idx, w, h, reset) in enumerate(sc["steps"]):
        g, cam, prev, color = frame_inputs(scene, idx, w, h, rev)
        alpha = 1.0 if n % 2 == 0 else 0.6
        ssao_a, ssr_a, taa_a, bloom_a = attribs(sc["algo"], reset, alpha)
        cmds = host.frame(idx, w, h, cam, prev, ssao=ssao_a, ssr=ssr_a, taa=taa_a, bloom=bloom_a, taa_flags=sc["taa_flags"], ssao_flags=sc["ssao_flags"], ssr_flags=sc["ssr_flags"],
                          postfx_flags=sc["postfx_flags"], timer=alpha)
        out = rp.run(cmds, {"depth": g["depth"], "prev_depth": g["prev_depth"], "motion": g["motion"], "normal": g["normal"], "material": g["material"], "color": color})
        pf = chain.postfx(idx, g["depth"], g["prev_depth"], g["motion"], cam, prev, blue_noise_tables())
        want = {"reprojected_depth": pf["reproj_depth"], "closest_motion": pf["closest_motion"], "previous_depth": g["prev_depth"],
                "ssr": chain.ssr(pf, color, g["depth"], g["normal"], g["material"], g["motion"], ssr_a, None, previous_frame=bool(sc["ssr_flags"] & 1), half_resolution=bool(sc["ssr_flags"] & 2)),
                "ssao": chain.ssao(pf, g["depth"], g["normal"], ssao_a, None, half_resolution=bool(sc["ssao_flags"] & 2), half_precision_depth=bool(sc["ssao_flags"] & 1))}
        want["taa"] = chain.taa(pf, color, taa_a, None)
        want["bloom"] = chain.bloom(want["taa"], bloom_a, None)
        for k, w_ in want.items():
            assert np.array_equal(out[k], w_), (name, idx, k, float(np.abs(out[k] - w_).max()), int((out[k] != w_).sum()))
        # the two blue-noise planes of the frame (what every stochastic pass of the frame read)
        noise = [t for t in rp.tex.values() if t["name"] == "PostFXContext::BlueNoiseTexture"]
        assert len(noise) == 2 and np.array_equal(noise[0]["planes"][0], pf["noise_xy"]) and np.array_equal(noise[1]["planes"][0], pf["noise_zw"])
    host.close()


def test_recorded_pass_list_of_a_steady_frame():
    """What the reference's classes draw in one steady-state frame, by debug group and pass: the list csrc/api_*.cpp were written from, now recorded from the classes
    themselves.  (A change of the reference's sequencing shows up here by name, not only as a numeric difference.)"""
    ref = pyref.ref_lib()
    host, rp = refhost.RefHost(15), refhost.Replayer(ref)
    scene = synth.Scene()
    for idx in (0, 1):
        g, cam, prev, color = frame_inputs(scene, idx, 96, 64, False)
        ssao_a, ssr_a, taa_a, bloom_a = attribs(0, 0, 1.0)
        cmds = host.frame(idx, 96, 64, cam, prev, ssao=ssao_a, ssr=ssr_a, taa=taa_a, bloom=bloom_a, taa_flags=2, timer=1.0)
        rp.run(cmds, {"depth": g["depth"], "prev_depth": g["prev_depth"], "motion": g["motion"], "normal": g["normal"], "material": g["material"], "color": color})
    got = [p for _, p in rp.passes]
    want = (["blue_noise", "reprojected_depth", "closest_motion", "copy:PostFXContext::ComputePreviousDepth"]
            + ["copy:PostFXContext::CopyTextureDepth"] + ["ssr_hiz_mip"] * 6 + ["ssr_mask_roughness", "ssr_intersection", "ssr_spatial_reconstruction", "ssr_temporal_accumulation", "ssr_bilateral_cleanup"]
            + ["copy:PostFXContext::CopyTextureDepth"] + ["ssao_prefiltered_depth_mip"] * 4 + ["ssao_compute_ao_gtao", "ssao_temporal_accumulation", "copy:PostFXContext::CopyTextureDepth"]
            + ["ssao_convoluted_history_mip"] * 4 + ["ssao_resampled_history", "ssao_spatial_reconstruction"]
            + ["taa_flags2"]
            + ["bloom_prefilter"] + ["bloom_downsample"] * 3 + ["bloom_upsample"] * 4)  # (48 x 32 first level: 6 levels exist, Radius 0.75 uses 4)
    assert got == want, got
    groups = {g.split("/")[0] for g, _ in rp.passes}
    assert groups == {"PreparePostFX", "ScreenSpaceReflection", "ScreenSpaceAmbientOcclusion", "TemporalAccumulation", "Bloom"}, groups
    # clears and copies of the frame, in order: what is cleared to what (texture names of the reference)
    ops = []
    for c in cmds:
        if c["op"] == "clear":
            ops.append(("clear", rp.tex[c["view"]["tex"]]["name"].split("::")[1], tuple(c["color"])))
        elif c["op"] == "clear_depth":
            ops.append(("clear_depth", rp.tex[c["view"]["tex"]]["name"].split("::")[1], c["depth"]))
        elif c["op"] == "copy":
            ops.append(("copy", rp.tex[c["src"]]["name"].split("::")[1], rp.tex[c["dst"]]["name"].split("::")[1]))
    assert ops == [("clear_depth", "DepthStencilMask", 0), ("clear", "Radiance", (0, 0, 0, 0)), ("clear", "RayDirectionPDF", (0, 0, 0, 0)), ("clear", "Output", (0, 0, 0, 0)),
                   ("clear", "Occlusion", (1, 0, 0, 0)), ("clear", "OcclusionHistory", (1, 0, 0, 0)), ("clear", "OcclusionHistoryLength", (1, 0, 0, 0)),
                   ("copy", "OcclusionHistory", "OcclusionHistoryConvoluted"), ("copy", "OcclusionHistoyResolved", "OcclusionHistory")], ops
    assert not [c for c in cmds if c["op"] == "error"]
    host.close()


def test_taa_first_frame_of_a_flag_set_is_a_copy():
    """TemporalAntiAliasing evaluates `m_AllPSOsReady` in PrepareResources, before Execute creates the technique (TemporalAntiAliasing.cpp:161-171, 184): the first frame of
    every flag set is ComputePlaceholderTexture -- CopyTextureColor of the frame, alpha included -- and the second accumulates onto it.  Round 4 found this by executing
    the reference; mifx_taa_execute and cpu_chain.taa now do the same."""
    ref = pyref.ref_lib()
    host, rp = refhost.RefHost(refhost.RefHost.TAA), refhost.Replayer(ref)
    scene = synth.Scene()
    seen = []
    for idx, flags in ((0, 2), (1, 2), (2, 5), (3, 5), (4, 2)):
        g, cam, prev, color = frame_inputs(scene, idx, 64, 48, False)
        cmds = host.frame(idx, 64, 48, cam, prev, taa=B.TAAAttribs.default(), taa_flags=flags)
        out = rp.run(cmds, {"depth": g["depth"], "prev_depth": g["prev_depth"], "motion": g["motion"], "color": color})
        seen.append([p for _, p in rp.passes if p.startswith(("taa", "copy:PostFXContext::CopyTextureColor"))])
        if seen[-1] == ["copy:PostFXContext::CopyTextureColor"]:
            assert np.array_equal(out["taa"], color)
    assert seen == [["copy:PostFXContext::CopyTextureColor"], ["taa_flags2"], ["copy:PostFXContext::CopyTextureColor"], ["taa_flags5"], ["taa_flags2"]], seen
    host.close()


# (frame index, width, height, DOF feature flags, (ring count, ring density))
DOF_STEPS = [(0, 96, 64, 1, (5, 7)), (1, 96, 64, 1, (5, 7)), (2, 96, 64, 1, (5, 7)), (3, 96, 64, 3, (5, 7)), (4, 96, 64, 3, (5, 7)), (5, 96, 64, 3, (4, 5)), (6, 80, 48, 3, (4, 5)),
             (7, 80, 48, 3, (5, 7)), (8, 80, 48, 0, (5, 7)), (9, 80, 48, 2, (3, 4))]


def test_depth_of_field_host_sequence():
    """DepthOfField.cpp executed (TAA -> depth of field -> Bloom, HnPostProcessTask.cpp:871-918) -- temporal circle of confusion with its ping-pong and its
    cleared history, the Karis permutation, a change of the feature flags (every target is re-created: DepthOfField.cpp:184-193), a change of the bokeh kernel (UpdateTexture
    of the first n texels, :801-809), a resize -- against cpu_chain.dof, the restatement csrc/api_dof.cpp follows.  Bit for bit, every intermediate target included."""
    ref = pyref.ref_lib()
    host, rp = refhost.RefHost(refhost.RefHost.TAA | refhost.RefHost.BLOOM | refhost.RefHost.DOF), refhost.Replayer(ref)
    chain = cpu_chain.CpuChain(ref, "ref_", taa_flags=2)
    scene = synth.Scene()
    seen = None
    for n, (idx, w, h, flags, (rings, density)) in enumerate(DOF_STEPS):
        g, cam, prev, color = frame_inputs(scene, idx, w, h, False)
        alpha = 1.0 if n % 2 == 0 else 0.7
        _, _, taa_a, bloom_a = attribs(0, 0, alpha)
        dof_a = B.DOFAttribs.default()
        dof_a.BokehKernelRingCount, dof_a.BokehKernelRingDensity, dof_a.AlphaInterpolation = rings, density, alpha
        cmds = host.frame(idx, w, h, cam, prev, taa=taa_a, bloom=bloom_a, dof=dof_a, dof_flags=flags, taa_flags=2, timer=alpha)
        out = rp.run(cmds, {"depth": g["depth"], "prev_depth": g["prev_depth"], "motion": g["motion"], "color": color})
        pf = chain.postfx(idx, g["depth"], g["prev_depth"], g["motion"], cam, prev, blue_noise_tables())
        taa = chain.taa(pf, color, taa_a, None)
        byname = {}  # the reference's targets by their texture names
        for t in rp.tex.values():
            byname.setdefault(t["name"], []).append(t["planes"][0])
        # The kernel tables.  The Gauss kernel and the small Octaweb kernel equal ref_dof_kernel_points / ref_dof_gauss_kernel (GenerateKernelPoints compiled through the shader
        # shim) bit for bit; the large kernel within one ulp: `cos(Theta)` with a float argument is cosf() there and cos(double) in this compilation of DepthOfField.cpp (which
        # overload an unqualified cos(float) finds depends on the platform's headers) -- so the passes below run on the table the executed host code uploaded.
        want_tables = chain.dof_tables(rings, density)
        count = 1 + density * (rings - 1) * rings // 2  # GenerateKernelPoints (DepthOfField.cpp:49-74); UpdateTexture replaces the first `count` texels only (:801-809)
        large = byname["DepthOfField::LargeBokehKernel"][0]
        assert np.abs(large[:, :count] - want_tables[0][:, :count]).max() <= 6e-8, idx
        assert np.array_equal(byname["DepthOfField::SmallBokehKernel"][0], want_tables[1]) and np.array_equal(byname["DepthOfField::GaussKernel"][0], want_tables[2])
        keep = {}
        dof = chain.dof(pf, taa, g["depth"], dof_a, flags, keep, tables=(large.copy(), want_tables[1], want_tables[2]))
        bloom = chain.bloom(dof, bloom_a, None)
        for k, want in (("taa", taa), ("dof", dof), ("bloom", bloom)):
            assert np.array_equal(out[k], want), (idx, k, float(np.abs(out[k] - want).max()), int((out[k] != want).sum()))
        assert np.array_equal(byname["DepthOfField::CircleOfConfusion"][0], keep["dof_coc"])
        for got, want in zip(byname["DepthOfField::DilationCircleOfConfusion"], keep["dof_dilation"][:3] + [keep["dof_blur_y"]]):  # (the last level is blurred in place, via Intermediate)
            assert np.array_equal(got, want), idx
        assert np.array_equal(byname["DepthOfField::DilationCircleOfConfusionIntermediate"][0], keep["dof_blur_x"])
        for got, want in zip(byname["DepthOfField::Prefiltered"], keep["dof_fill"]):     # the second bokeh pass writes the prefiltered targets again (.cpp:1049-1058)
            assert np.array_equal(got, want), idx
        for got, want in zip(byname["DepthOfField::Bokeh"], keep["dof_post"]):           # and the post-filter the bokeh targets (.cpp:1073-1080)
            assert np.array_equal(got, want), idx
        seen = [p for _, p in rp.passes if p.startswith("dof")]
        assert seen == ["dof_coc"] + (["dof_temporal_coc"] if flags & 1 else []) + ["dof_separated_coc"] + ["dof_dilation_coc"] * 3 + ["dof_blur_x", "dof_blur_y", "dof_prefilter",
                        "dof_bokeh_first_karis" if flags & 2 else "dof_bokeh_first", "dof_bokeh_second", "dof_postfilter", "dof_combine"], seen
    assert not [c for c in cmds if c["op"] == "error"]
    host.close()


@pytest.mark.parametrize("size", [(96, 64), (200, 120), (70, 36)])
def test_bloom_mip_count_follows_the_radius(size):
    """Bloom::ComputeMipCount (Bloom.cpp:152-156: int(Radius * ComputeMipLevelsCount(w / 2, h / 2))) decides how many levels the down / up loops walk (:313-396).  The executed
    class, for a sweep of Radius (changing from frame to frame, as a slider would): the number of down-sample / up-sample draws it records, and its output, against cpu_chain.bloom --
    whose refusal of a radius that leaves fewer than two levels is where the reference would sample a level nobody wrote."""
    w, h = size
    ref = pyref.ref_lib()
    host, rp = refhost.RefHost(refhost.RefHost.BLOOM), refhost.Replayer(ref)
    chain = cpu_chain.CpuChain(ref, "ref_")
    scene = synth.Scene()
    levels = cpu_chain.compute_mip_levels_count(w // 2, h // 2)
    seen = []
    for idx, radius in enumerate([0.75, 0.4, 1.0, 0.55, 0.3, 0.9, 0.75]):
        g, cam, prev, color = frame_inputs(scene, idx, w, h, False)
        bloom_a = B.BloomAttribs.default()
        bloom_a.Radius, bloom_a.AlphaInterpolation = radius, 1.0
        mips = int(np.float32(radius) * np.float32(levels))
        if mips < 2:
            continue
        cmds = host.frame(idx, w, h, cam, prev, bloom=bloom_a, timer=1.0)
        out = rp.run(cmds, {"depth": g["depth"], "prev_depth": g["prev_depth"], "motion": g["motion"], "color": color})
        passes = [p for _, p in rp.passes if p.startswith("bloom")]
        assert passes == ["bloom_prefilter"] + ["bloom_downsample"] * (mips - 1) + ["bloom_upsample"] * mips, (radius, mips, passes)
        want = chain.bloom(color, bloom_a, None)  # (no TAA object in this host: Bloom takes the frame the application composed = the colour input)
        assert np.array_equal(out["bloom"], want), (radius, float(np.abs(out["bloom"] - want).max()))
        seen.append(mips)
    assert len(set(seen)) >= 3, seen
    host.close()


def test_ssao_algorithm_changes_without_a_reset():
    """The AO algorithm is an attribute (ScreenSpaceAmbientOcclusionAttribs::Algorithm -> another pipeline of the same effect, ScreenSpaceAmbientOcclusion.cpp:476-479): switching it
    between frames keeps every target and the accumulated history.  Executed: GTAO, GTAO, HBAO, VBAO, GTAO on consecutive frames against cpu_chain with its algorithm switched alike."""
    ref = pyref.ref_lib()
    host, rp = refhost.RefHost(refhost.RefHost.SSAO), refhost.Replayer(ref)
    chain = cpu_chain.CpuChain(ref, "ref_")
    scene = synth.Scene()
    for idx, algo in enumerate([0, 0, 1, 2, 0]):
        g, cam, prev, color = frame_inputs(scene, idx, 96, 64, False)
        ssao_a = B.SSAOAttribs.default()
        ssao_a.Algorithm, ssao_a.AlphaInterpolation = algo, 1.0
        cmds = host.frame(idx, 96, 64, cam, prev, ssao=ssao_a, timer=1.0)
        out = rp.run(cmds, {"depth": g["depth"], "prev_depth": g["prev_depth"], "motion": g["motion"], "normal": g["normal"]})
        assert [p for _, p in rp.passes if p.startswith("ssao_compute_ao")] == ["ssao_compute_ao_" + ALGOS[algo]]
        assert not [c for c in cmds if c["op"] == "create_texture" and idx > 0], "no target is re-created by a change of the algorithm"
        chain.algorithm = ALGOS[algo]
        pf = chain.postfx(idx, g["depth"], g["prev_depth"], g["motion"], cam, prev, blue_noise_tables())
        want = chain.ssao(pf, g["depth"], g["normal"], ssao_a, None)
        assert np.array_equal(out["ssao"], want), (idx, algo, int((out["ssao"] != want).sum()))
    host.close()


def test_ssr_and_ssao_feature_flags_change_between_frames():
    """What a change of the effects' feature flags does to their targets and histories, executed: ScreenSpaceReflection and ScreenSpaceAmbientOcclusion with HALF_RESOLUTION /
    PREVIOUS_FRAME / HALF_PRECISION_DEPTH switched on and off on consecutive frames (ScreenSpaceReflection.cpp:65-90, ScreenSpaceAmbientOcclusion.cpp:65-96)."""
    ref = pyref.ref_lib()
    host, rp = refhost.RefHost(refhost.RefHost.SSAO | refhost.RefHost.SSR), refhost.Replayer(ref)
    chain = cpu_chain.CpuChain(ref, "ref_")
    scene = synth.Scene()
    steps = [(0, 0), (0, 0), (2, 2), (2, 2), (0, 0), (1, 1), (1, 1), (0, 0), (2, 2), (1, 2), (1, 0)]  # (SSR flags, SSAO flags) of frames 0 .. 10 (the reference build has no PREVIOUS_FRAME + HALF_RESOLUTION permutation of R4)
    for idx, (ssr_flags, ssao_flags) in enumerate(steps):
        g, cam, prev, color = frame_inputs(scene, idx, 96, 64, False)
        ssao_a, ssr_a, _, _ = attribs(0, 0, 1.0)
        cmds = host.frame(idx, 96, 64, cam, prev, ssao=ssao_a, ssr=ssr_a, ssr_flags=ssr_flags, ssao_flags=ssao_flags, timer=1.0)
        out = rp.run(cmds, {"depth": g["depth"], "prev_depth": g["prev_depth"], "motion": g["motion"], "normal": g["normal"], "material": g["material"], "color": color})
        pf = chain.postfx(idx, g["depth"], g["prev_depth"], g["motion"], cam, prev, blue_noise_tables())
        want_ssr = chain.ssr(pf, color, g["depth"], g["normal"], g["material"], g["motion"], ssr_a, None, previous_frame=bool(ssr_flags & 1), half_resolution=bool(ssr_flags & 2))
        continue_ssao = True  # (flag sets 0, 1, 2: the reference build has no half-resolution + half-precision permutation of A3)
        want_ssao = chain.ssao(pf, g["depth"], g["normal"], ssao_a, None, half_resolution=bool(ssao_flags & 2), half_precision_depth=bool(ssao_flags & 1))
        assert np.array_equal(out["ssr"], want_ssr), (idx, "ssr", ssr_flags, int((out["ssr"] != want_ssr).sum()))
        if continue_ssao:
            assert np.array_equal(out["ssao"], want_ssao), (idx, "ssao", ssao_flags, int((out["ssao"] != want_ssao).sum()))
    host.close()


def test_frame_indices_that_repeat_go_back_or_skip_an_effect():
    """The reset rule `FrameDesc.Index != last + 1` (ScreenSpaceAmbientOcclusion.cpp:797-800, TemporalAntiAliasing.cpp:125-128) with indices that repeat, go backwards and jump, and
    with an effect that the application leaves out for a frame (its own last index then lags: the next frame is a reset for THAT effect only)."""
    ref = pyref.ref_lib()
    host, rp = refhost.RefHost(15), refhost.Replayer(ref)
    chain = cpu_chain.CpuChain(ref, "ref_", taa_flags=2)
    scene = synth.Scene()
    # (frame index, run SSAO?, run SSR?, run TAA + Bloom?)
    steps = [(5, 1, 1, 1), (5, 1, 1, 1), (6, 1, 1, 1), (4, 1, 1, 1), (5, 1, 1, 1), (6, 0, 1, 1), (7, 1, 1, 0), (8, 1, 0, 1), (9, 1, 1, 1), (10, 1, 1, 1)]
    for idx, do_ssao, do_ssr, do_taa in steps:
        g, cam, prev, color = frame_inputs(scene, idx, 96, 64, False)
        ssao_a, ssr_a, taa_a, bloom_a = attribs(0, 0, 1.0)
        cmds = host.frame(idx, 96, 64, cam, prev, ssao=ssao_a if do_ssao else None, ssr=ssr_a if do_ssr else None, taa=taa_a if do_taa else None, bloom=bloom_a if do_taa else None,
                          taa_flags=2, timer=1.0)
        out = rp.run(cmds, {"depth": g["depth"], "prev_depth": g["prev_depth"], "motion": g["motion"], "normal": g["normal"], "material": g["material"], "color": color})
        pf = chain.postfx(idx, g["depth"], g["prev_depth"], g["motion"], cam, prev, blue_noise_tables())
        want = {}
        if do_ssr:
            want["ssr"] = chain.ssr(pf, color, g["depth"], g["normal"], g["material"], g["motion"], ssr_a, None)
        if do_ssao:
            want["ssao"] = chain.ssao(pf, g["depth"], g["normal"], ssao_a, None)
        if do_taa:
            want["taa"] = chain.taa(pf, color, taa_a, None)
            want["bloom"] = chain.bloom(want["taa"], bloom_a, None)
        for k, w_ in want.items():
            assert np.array_equal(out[k], w_), (idx, k, int((out[k] != w_).sum()))
    host.close()


def test_reversed_depth_switched_between_frames():
    """PostFXContext::FEATURE_FLAG_REVERSED_DEPTH toggled at run time: SSAO and SSR pick it up in PrepareResources (ScreenSpaceAmbientOcclusion.cpp:72-84,
    ScreenSpaceReflection.cpp:72-84: other shader permutations, no target is re-created, the histories continue)."""
    ref = pyref.ref_lib()
    host, rp = refhost.RefHost(15), refhost.Replayer(ref)
    chain = cpu_chain.CpuChain(ref, "ref_", taa_flags=2)
    scene = synth.Scene()
    for idx, rev in enumerate([False, False, True, True, False]):
        g, cam, prev, color = frame_inputs(scene, idx, 96, 64, rev)
        ssao_a, ssr_a, taa_a, bloom_a = attribs(0, 0, 1.0)
        cmds = host.frame(idx, 96, 64, cam, prev, ssao=ssao_a, ssr=ssr_a, taa=taa_a, bloom=bloom_a, taa_flags=2, postfx_flags=1 if rev else 0, timer=1.0)
        assert idx == 0 or not [c for c in cmds if c["op"] == "create_texture"], "no target is re-created by the switch"
        out = rp.run(cmds, {"depth": g["depth"], "prev_depth": g["prev_depth"], "motion": g["motion"], "normal": g["normal"], "material": g["material"], "color": color})
        chain.reversed_depth = rev
        pf = chain.postfx(idx, g["depth"], g["prev_depth"], g["motion"], cam, prev, blue_noise_tables())
        want = {"ssr": chain.ssr(pf, color, g["depth"], g["normal"], g["material"], g["motion"], ssr_a, None), "ssao": chain.ssao(pf, g["depth"], g["normal"], ssao_a, None)}
        want["taa"] = chain.taa(pf, color, taa_a, None)
        want["bloom"] = chain.bloom(want["taa"], bloom_a, None)
        for k, w_ in want.items():
            assert np.array_equal(out[k], w_), (idx, rev, k, int((out[k] != w_).sum()))
    host.close()


def test_blue_noise_and_ping_pong_at_large_frame_indices():
    """PostFXContext hands the frame index to the blue-noise pass through the draw's first vertex (PostFXContext.cpp:300-312) and the effects ping-pong by Index & 1: frame indices
    around the powers of two a modulus could hide behind (127 / 128, 255 / 256, 1023 / 1024, 65535 / 65536) and the bench's range (1000+)."""
    ref = pyref.ref_lib()
    host, rp = refhost.RefHost(refhost.RefHost.SSAO | refhost.RefHost.TAA), refhost.Replayer(ref)
    chain = cpu_chain.CpuChain(ref, "ref_", taa_flags=2)
    scene = synth.Scene()
    for idx in (127, 128, 255, 256, 1000, 1001, 1023, 1024, 65535, 65536, 65537):
        g, cam, prev, color = frame_inputs(scene, idx % 64, 64, 48, False)  # (the camera of a nearby orbit position; the index under test goes into FrameDesc only)
        ssao_a, _, taa_a, _ = attribs(0, 0, 1.0)
        cmds = host.frame(idx, 64, 48, cam, prev, ssao=ssao_a, taa=taa_a, taa_flags=2, timer=1.0)
        out = rp.run(cmds, {"depth": g["depth"], "prev_depth": g["prev_depth"], "motion": g["motion"], "normal": g["normal"], "color": color})
        pf = chain.postfx(idx, g["depth"], g["prev_depth"], g["motion"], cam, prev, blue_noise_tables())
        noise = [t for t in rp.tex.values() if t["name"] == "PostFXContext::BlueNoiseTexture"]
        assert np.array_equal(noise[0]["planes"][0], pf["noise_xy"]) and np.array_equal(noise[1]["planes"][0], pf["noise_zw"]), idx
        assert np.array_equal(out["ssao"], chain.ssao(pf, g["depth"], g["normal"], ssao_a, None)), idx
        assert np.array_equal(out["taa"], chain.taa(pf, color, taa_a, None)), idx
    host.close()


@pytest.mark.parametrize("mode,options,mip,alpha", [(0, 0, 1.0, 0.0), (4, 1, 0.0, 0.25), (0, 4, 2.4, 1.0), (0, 2, 1.0, 0.0)])
def test_envmap_renderer_host(mode, options, mip, alpha):
    """SURVEY 8f N2, host side: Components/src/EnvMapRenderer.cpp executed (Prepare + Render).  What it uploads and how it draws, against what the tests of mifx_envmap_render hand
    the checker (tests/test_oracle_vs_ref.py run_envmap: ToneMappingAttribs + {AverageLogLum, MipLevel, Alpha, Scale}): the constant buffer byte for byte in the shader's cbuffer
    order (EnvMap.psh: g_ToneMappingAttribs, g_AverageLogLum, g_MipLevel, g_Alpha, padding, g_Scale with w = 1), the macros of the permutation, one full-screen triangle with the
    depth test LESS_EQUAL (GREATER_EQUAL with OPTION_FLAG_USE_REVERSE_DEPTH) and no depth writes; then the draw replayed with the reference's shader equals run_envmap."""
    import struct

    from test_oracle_vs_ref import envmap_inputs, run_envmap, tone_mapping_attribs_bytes

    ref = pyref.ref_lib()
    inp = envmap_inputs()
    h, w = inp["depth"].shape
    scale = (1.5, 1.0, 0.75)
    tm = tone_mapping_attribs_bytes(mode)
    cmds = refhost.envmap_render(w, h, options, tm, bytes(inp["cam"]) + bytes(inp["prev"]), 0.3, mip, alpha, scale, cube=True, env_size=inp["env"][0].shape[1], env_mips=len(inp["env"]))
    assert not [c for c in cmds if c["op"] == "error"], cmds
    host = refhost.RefHost(0)
    assert host.sizeof("ToneMappingAttribs") == len(tm)
    host.close()
    import base64

    bufs = {}
    for c in cmds:
        if c["op"] == "create_buffer":
            bufs[c["id"]] = [c["name"], base64.b64decode(c["bytes_b64"])]
        elif c["op"] == "update_buffer":
            bufs[c["buf"]][1] = base64.b64decode(c["bytes_b64"])
    cb = [b for n, b in bufs.values() if n == "EnvMap Render Attribs CB"]
    assert len(cb) == 1 and cb[0] == tm + struct.pack("<4f", 0.3, mip, alpha, 0.0) + struct.pack("<4f", *scale, 1.0), "EnvMapShaderAttribs as the class fills it"
    draws = [c for c in cmds if c["op"] == "draw"]
    assert len(draws) == 1
    d = draws[0]
    assert d["ps"]["file"] == "EnvMap.psh" and d["vs"]["file"] == "EnvMap.vsh" and d["instances"] == 1 and d["start_vertex"] == 0
    m = d["ps"]["macros"]
    assert m["TONE_MAPPING_MODE"] == str(mode) and m["CONVERT_OUTPUT_TO_SRGB"] == str(options & 1) and m["COMPUTE_MOTION_VECTORS"] == str((options >> 1) & 1), m
    assert m["ENV_MAP_TYPE"] == m["ENV_MAP_TYPE_CUBE"] == "0" and m["ENV_MAP_TYPE_SPHERE"] == "1", m
    assert d["depth"]["enable"] and not d["depth"]["write"] and d["depth"]["func"] == ("GREATER_EQUAL" if options & 4 else "LESS_EQUAL"), d["depth"]
    assert len(d["rtvs"]) == 2 and d["dsv"] is not None and set(d["vars"]) >= {"EnvMap", "cbCameraAttribs", "cbEnvMapRenderAttribs"}, (d["rtvs"], list(d["vars"]))
    if options & 4:
        return  # (reversed depth: the pipeline state is what differs; the shader wrapper below is the LESS_EQUAL one)
    # the draw, with what the class bound: tone-mapping block and the four scalars + scale out of ITS constant buffer, the two cameras out of the camera buffer
    cams = bufs[d["vars"]["cbCameraAttribs"]["buf"]][1]
    fv = struct.unpack("<8f", cb[0][len(tm):])
    color, motion = np.full((h, w, 4), -7.0, np.float32), np.full((h, w, 2), -7.0, np.float32)
    ref.call("ref_envmap_ldr" if mode else "ref_envmap", [inp["env"], inp["depth"]], [color, motion], cam0=cams[:576], cam1=cams[576:1152], attribs=cb[0][:len(tm)],
             fval=[fv[0], fv[1], fv[2], fv[4], fv[5], fv[6]])
    want_c, want_m = run_envmap(ref, "ref_", inp, mode, options & 1, mip, alpha, scale)
    assert np.array_equal(color, want_c) and np.array_equal(motion, want_m)


def random_sequence(seed, steps=30):
    """One random run of the reference's executed host classes against cpu_chain: every step may resize the frame, move FrameDesc.Index by 0 / -1 / +1 / +2 / +5, change
    the feature flags of SSR / SSAO / TAA / depth of field, the AO algorithm, the bokeh kernel, REVERSED_DEPTH, Bloom's radius, AlphaInterpolation, request a reset, or leave an
    effect out.  The camera walks consecutive orbit positions whatever the index does.  Returns None, or a description of the first difference.  (Permutations the reference
    build of oracle/_ref does not have -- PREVIOUS_FRAME + HALF_RESOLUTION, half precision beyond GTAO, the flag sets under reversed depth -- are not drawn.)"""
    import random

    rnd = random.Random(seed)
    ref = pyref.ref_lib()
    host, rp = refhost.RefHost(31), refhost.Replayer(ref)
    chain = cpu_chain.CpuChain(ref, "ref_", taa_flags=2)
    scene = synth.Scene()
    idx, size = rnd.randrange(0, 50), (96, 64)
    st = dict(ssao_flags=0, ssr_flags=0, taa_flags=2, dof_flags=0, algo=0, rev=False, rings=(5, 7))
    try:
        for n in range(steps):
            idx = max(idx + (rnd.choice([0, -1, 2, 5]) if rnd.random() < 0.10 else 1), 0)
            if rnd.random() < 0.12:
                size = rnd.choice([(96, 64), (80, 48), (70, 36), (128, 72)])
            for key, p, values in (("ssr_flags", 0.15, [0, 1, 2]), ("ssao_flags", 0.15, [0, 1, 2]), ("taa_flags", 0.15, [0, 2, 5, 7]), ("dof_flags", 0.15, [0, 1, 2, 3]),
                                   ("algo", 0.10, [0, 1, 2]), ("rings", 0.10, [(5, 7), (4, 5), (3, 4), (2, 3)])):
                if rnd.random() < p:
                    st[key] = rnd.choice(values)
            if rnd.random() < 0.08:
                st["rev"] = not st["rev"]
            algo, ssao_flags, ssr_flags, rev = st["algo"], st["ssao_flags"], st["ssr_flags"], st["rev"]
            if ssao_flags & 1 and algo != 0:
                ssao_flags &= ~1
            if rev:
                ssr_flags = ssao_flags = algo = 0
            w, h = size
            do = {k: rnd.random() > 0.08 for k in ("ssao", "ssr", "taa", "dof", "bloom")}
            do["dof"] = do["dof"] and do["taa"]  # (HnPostProcessTask runs depth of field only behind TAA)
            reset, alpha = rnd.random() < 0.07, rnd.choice([1.0, 0.6, 0.3])
            g, cam, prev, color = frame_inputs(scene, 20 + n, w, h, rev)
            ssao_a, ssr_a, taa_a, bloom_a = attribs(algo, reset, alpha)
            bloom_a.Radius = rnd.choice([0.75, 0.5, 1.0])
            dof_a = B.DOFAttribs.default()
            dof_a.AlphaInterpolation = alpha
            dof_a.BokehKernelRingCount, dof_a.BokehKernelRingDensity = st["rings"]
            cmds = host.frame(idx, w, h, cam, prev, ssao=ssao_a if do["ssao"] else None, ssr=ssr_a if do["ssr"] else None, taa=taa_a if do["taa"] else None,
                              bloom=bloom_a if do["bloom"] else None, dof=dof_a if do["dof"] else None, taa_flags=st["taa_flags"], ssao_flags=ssao_flags, ssr_flags=ssr_flags,
                              dof_flags=st["dof_flags"], postfx_flags=1 if rev else 0, timer=alpha)
            out = rp.run(cmds, {"depth": g["depth"], "prev_depth": g["prev_depth"], "motion": g["motion"], "normal": g["normal"], "material": g["material"], "color": color})
            chain.reversed_depth, chain.algorithm, chain.taa_flags = rev, ALGOS[algo], st["taa_flags"]
            pf = chain.postfx(idx, g["depth"], g["prev_depth"], g["motion"], cam, prev, blue_noise_tables())
            chain.prepare(w, h, ssr_flags=ssr_flags, ssao_flags=ssao_flags, dof_flags=st["dof_flags"])  # (every effect is prepared every frame, executed or not)
            want = {}
            if do["ssr"]:
                want["ssr"] = chain.ssr(pf, color, g["depth"], g["normal"], g["material"], g["motion"], ssr_a, None, previous_frame=bool(ssr_flags & 1), half_resolution=bool(ssr_flags & 2))
            if do["ssao"]:
                want["ssao"] = chain.ssao(pf, g["depth"], g["normal"], ssao_a, None, half_resolution=bool(ssao_flags & 2), half_precision_depth=bool(ssao_flags & 1))
            frame = color
            if do["taa"]:
                want["taa"] = frame = chain.taa(pf, color, taa_a, None)
            if do["dof"]:
                large = [t["planes"][0] for t in rp.tex.values() if t["name"] == "DepthOfField::LargeBokehKernel"][0].copy()
                tabs = chain.dof_tables(*st["rings"])
                want["dof"] = frame = chain.dof(pf, frame, g["depth"], dof_a, st["dof_flags"], None, tables=(large, tabs[1], tabs[2]))
            if do["bloom"] and int(np.float32(bloom_a.Radius) * np.float32(cpu_chain.compute_mip_levels_count(w // 2, h // 2))) >= 2:
                want["bloom"] = chain.bloom(frame, bloom_a, None)
            for k, w_ in want.items():
                if not np.array_equal(out[k], w_):
                    return f"seed {seed} step {n} (FrameDesc.Index {idx}, {w}x{h}): {k}: {int((out[k] != w_).sum())} values differ; {st}, executed {do}, reset {reset}"
    finally:
        host.close()
    return None


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_random_sequences_of_flag_size_index_changes(seed):
    """Differential test against the executed reference classes (`python tests/test_host_sequence_vs_ref.py FIRST LAST` runs more seeds): 30 random steps, bit for bit.  Found
    in round 4: SSR's R5 / R6 leave texels outside the reflection mask as an earlier frame wrote them and R6 / R7 read them beside the mask's edge -- the checker (and the
    product) wrote 0 there, which 0.1-0.2 % of the SSR output saw on camera positions the fixed scenarios above happened not to visit."""
    assert random_sequence(seed) is None


if __name__ == "__main__":
    import sys

    for s_ in range(int(sys.argv[1]), int(sys.argv[2])):
        r_ = random_sequence(s_)
        print(s_, "OK" if r_ is None else r_, flush=True)
