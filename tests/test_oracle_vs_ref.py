"""Pins the hand-written oracle (oracle/mifx_oracle.cpp) against the reference itself (oracle/_ref: the reference's
shader source compiled for the CPU) and against known-answer vectors generated from it.  CPU only."""
import numpy as np
import pytest

from util import assert_close, blue_noise_tables, tone_mapping_attribs_bytes

# reference ToneMapping.fxh compiled verbatim, default attribs, fAveLogLum = 0.3, RGB = (2.0, 0.5, 0.1)
TONEMAP_KAT = {
    1: (0.9556412, 0.2389103, 0.0477821), 2: (0.8140653, 0.2035163, 0.0407033), 3: (0.8569469, 0.2142367, 0.0428473),
    4: (0.8852031, 0.3240846, 0.0730275), 5: (0.7242465, 0.3552477, 0.0568655), 6: (0.7085059, 0.1771265, 0.0354253),
    7: (0.8269524, 0.2067381, 0.0413476), 8: (0.6856485, 0.3254194, 0.1326309), 9: (0.6856485, 0.3254194, 0.1326309),
    10: (0.9600000, 0.3211358, 0.1507719), 11: (0.9773243, 0.3990930, 0.2448979),
}


def hdr_test_image(h=48, w=64, seed=7):
    rng = np.random.default_rng(seed)
    lum = np.exp2(rng.uniform(-8, 8, (h, w)))
    rgb = rng.uniform(0.01, 1, (h, w, 3))
    img = np.concatenate([rgb * lum[..., None], np.ones((h, w, 1))], -1).astype(np.float32)
    img[0, 0, :3] = 0.0
    img[0, 1, :3] = 1e-12
    img[0, 2, :3] = 1e4
    img[0, 3, :3] = [-1.0, 0.5, 2.0]  # negative input exercises the max(color, 0) guard
    return np.ascontiguousarray(img)


@pytest.mark.parametrize("mode", range(1, 12))
def test_tonemap_kat(oracle, mode):
    img = np.array([[[2.0, 0.5, 0.1, 1.0]]], np.float32)
    out = np.zeros_like(img)
    oracle.call("oracle_tonemap", [img], [out], attribs=tone_mapping_attribs_bytes(mode), fval=[0.3], ival=[0])
    np.testing.assert_allclose(out[0, 0, :3], TONEMAP_KAT[mode], rtol=2e-6, atol=2e-7)
    assert out[0, 0, 3] == 1.0


@pytest.mark.parametrize("mode", range(0, 12))
@pytest.mark.parametrize("srgb", [0, 1])
def test_tonemap_oracle_vs_ref(oracle, ref, mode, srgb):
    img = hdr_test_image()
    a, b = np.zeros_like(img), np.zeros_like(img)
    attr = tone_mapping_attribs_bytes(mode, agx=(1.1, 0.95, 1.05, 0.01))
    oracle.call("oracle_tonemap", [img], [a], attribs=attr, fval=[0.3], ival=[srgb])
    ref.call("ref_tonemap", [img], [b], attribs=attr, fval=[0.3], ival=[srgb])
    assert_close(a, b, rtol=1e-6, atol=1e-9, what=f"tonemap mode {mode}")


@pytest.mark.parametrize("frame", [0, 1, 17, 255, 1000])
def test_blue_noise_oracle_vs_ref_bit_exact(oracle, ref, frame):
    sobol, tile = blue_noise_tables()
    s = sobol.astype(np.float32).reshape(1, 256)
    t = tile.astype(np.float32).reshape(256, 512)
    outs = []
    for lib, name in ((oracle, "oracle_blue_noise"), (ref, "ref_blue_noise")):
        xy, zw = np.zeros((128, 128, 2), np.float32), np.zeros((128, 128, 2), np.float32)
        lib.call(name, [s, t], [xy, zw], ival=[frame])
        outs.append((xy, zw))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    q = outs[0][0] * 255.0
    assert np.abs(q - np.round(q)).max() < 1e-4  # UNORM8 lattice


def small_frame(frame=3, w=96, h=64):
    import torch
    from diligentfx_amd import synth

    scene = synth.Scene()
    return synth.make_frame(scene, frame, w, h, torch.device("cpu"))


def test_prep_oracle_vs_ref(oracle, ref):
    from diligentfx_amd.binding import as_bytes

    f = small_frame()
    depth, motion = f["depth"].numpy(), f["motion"].numpy()
    c0, c1 = as_bytes(f["camera"]), as_bytes(f["prev_camera"])
    a, b = np.zeros_like(depth), np.zeros_like(depth)
    oracle.call("oracle_reprojected_depth", [depth], [a], cam0=c0, cam1=c1)
    ref.call("ref_reprojected_depth", [depth], [b], cam0=c0, cam1=c1)
    assert_close(a, b, rtol=1e-6, what="reprojected depth")
    assert np.abs(b - depth)[depth < 1].max() < 0.05  # sanity: small camera move => similar depth
    a, b = np.zeros_like(motion), np.zeros_like(motion)
    oracle.call("oracle_closest_motion", [depth, motion], [a])
    ref.call("ref_closest_motion", [depth, motion], [b])
    assert np.array_equal(a, b)
    assert (a[0] == 0).all() and (a[-1] == 0).all()  # unclamped 3x3 search: the 0 'depth' outside wins at the border


# ---------------------------------------------------------------------------------------------------------------------- auto exposure (N3)
def run_autoexposure(lib, prefix, img, steps):
    """steps: list of (elapsed_time, light_adaptation); returns the low-resolution luminance of the last step and the averages after each."""
    low = np.zeros((64, 64, 2), np.float32)
    avg = np.full((1, 1), 0.1, np.float32)  # the reference clears the 1x1 target to 0.1 (EpipolarLightScattering.cpp:892-905)
    seq = []
    for dt, adapt in steps:
        lib.call(prefix + "autoexposure", [img], [low, avg], fval=[dt], ival=[adapt])
        seq.append(float(avg[0, 0]))
    return low, seq


def test_autoexposure_oracle_vs_ref(oracle, ref):
    img = hdr_test_image(h=150, w=230, seed=21)
    steps = [(0.016, 1), (0.5, 1), (0.016, 0), (2.0, 1)]
    low_o, seq_o = run_autoexposure(oracle, "oracle_", img, steps)
    low_r, seq_r = run_autoexposure(ref, "ref_", img, steps)
    assert_close(low_o, low_r, rtol=1e-6, atol=1e-7, what="low-resolution luminance")
    np.testing.assert_allclose(seq_o, seq_r, rtol=2e-6)
    # without adaptation the average is the weighted geometric mean of the luminance of the 64x64 samples, whatever it was before
    lum = low_r[..., 0].astype(np.float64).mean() / max(low_r[..., 1].astype(np.float64).mean(), 1e-6)
    assert abs(seq_r[2] - np.exp(lum)) < 1e-4 * np.exp(lum)
    # with adaptation the first step moves 1 - exp(-dt) of the way from 0.1
    assert abs(seq_r[0] - (0.1 + (np.exp(lum) - 0.1) * (1 - np.exp(-0.016)))) < 1e-4 * seq_r[0]


def test_autoexposure_dark_frame_keeps_the_average(oracle):
    """All samples below MinLuminance: the weight is 0 and the previous average survives (saturate(LogLum_W.y / 1e-3) = 0)."""
    img = np.zeros((40, 40, 4), np.float32)
    img[..., :3] = 1e-3
    low, seq = run_autoexposure(oracle, "oracle_", img, [(1.0, 0)])
    assert seq[0] == pytest.approx(0.1) and float(np.abs(low[..., 1]).max()) == 0.0


# ------------------------------------------------------------------------------------------------ depth of field
def dof_frames(frames=(5, 6), w=96, h=64, lens=(12.0, 1.2, 135.0)):
    """Frames of the synthetic scene with a lens that blurs both the near and the far field, and an HDR colour buffer."""
    import torch
    from diligentfx_amd import synth
    from diligentfx_amd.binding import as_bytes

    scene = synth.Scene()
    out = []
    for fi in frames:
        f = synth.make_frame(scene, fi, w, h, torch.device("cpu"))
        cam = f["camera"]
        cam.fFocusDistance, cam.fFStop, cam.fFocalLength = lens
        color = synth.make_hdr_buffer(w, h, torch.device("cpu"), seed=100 + fi).numpy()
        out.append({"frame": fi, "depth": f["depth"].numpy(), "motion": f["motion"].numpy(), "cam": as_bytes(cam), "prev_cam": as_bytes(f["prev_camera"]),
                    "color": np.ascontiguousarray(color)})
    return out


def run_cpu_dof(lib, prefix, frames, attribs, flags):
    import cpu_chain

    chain = cpu_chain.CpuChain(lib, prefix)
    keeps = []
    for f in frames:
        cm = np.zeros_like(f["motion"])
        chain.call("closest_motion", [f["depth"], f["motion"]], [cm])
        pf = {"frame": f["frame"], "cam": f["cam"], "closest_motion": cm}
        keep = {}
        chain.dof(pf, f["color"], f["depth"], attribs, flags, keep)
        keeps.append(keep)
    return keeps


def flat(v):
    return list(v) if isinstance(v, (list, tuple)) else [v]


@pytest.mark.parametrize("rings,density", [(5, 7), (2, 2), (4, 3), (3, 7), (5, 2)])
def test_dof_host_tables_oracle_vs_ref_bit_exact(oracle, ref, rings, density):
    import cpu_chain

    a = cpu_chain.CpuChain(oracle, "oracle_").dof_tables(rings, density)
    b = cpu_chain.CpuChain(ref, "ref_").dof_tables(rings, density)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    n = 1 + density * (rings - 1) * rings // 2
    assert np.all(a[0][0, n:] == 0) and np.abs(np.hypot(a[0][0, :density * (rings - 1), 0], a[0][0, :density * (rings - 1), 1]) - 1).max() < 1e-6  # outer ring first
    assert abs(float(a[2].sum()) - 1.0) < 1e-6


@pytest.mark.parametrize("flags", [0, 1, 2, 3])
@pytest.mark.parametrize("lens", [(12.0, 1.2, 135.0), (10.0, 5.6, 50.0)])
def test_dof_oracle_vs_ref(oracle, ref, flags, lens):
    from diligentfx_amd.binding import DOFAttribs

    attribs = DOFAttribs.default()
    attribs.MaxCircleOfConfusion = 0.02
    attribs.AlphaInterpolation = 0.8
    frames = dof_frames(lens=lens)
    ka, kb = run_cpu_dof(oracle, "oracle_", frames, attribs, flags), run_cpu_dof(ref, "ref_", frames, attribs, flags)
    for fi, (a, b) in enumerate(zip(ka, kb)):
        for name in a:
            for lvl, (x, y) in enumerate(zip(flat(a[name]), flat(b[name]))):
                assert_close(x, y, rtol=1e-6, atol=1e-7, what=f"dof frame {fi} {name}[{lvl}] flags {flags}")
    out = ka[-1]
    # the frames exercise what they are meant to: both fields blurred, the temporal path blended, the combine pass changed the picture
    if lens[1] < 2.0:
        assert (out["dof_coc"] <= -0.999).any() and (out["dof_coc"] >= 0.999).any() and (out["dof_prefiltered"][0][..., 3] > 0.5).any()
        assert np.abs(out["dof_out"][..., :3] - frames[-1]["color"][..., :3]).max() > 0.1
    if flags & 1:
        assert not np.array_equal(out["dof_coc"], out["dof_coc_used"])


# ------------------------------------------------------------------------------------------------ environment-map background
def envmap_inputs(w=112, h=72, frame=9):
    import torch

    import chain_util
    from diligentfx_amd import synth
    from diligentfx_amd.binding import as_bytes

    f = synth.make_frame(synth.Scene(), frame, w, h, torch.device("cpu"))
    env = chain_util.box_mips(synth.make_sky_cube(32, torch.device("cpu")).clamp(max=500.0).numpy())
    return {"env": env, "depth": f["depth"].numpy(), "cam": as_bytes(f["camera"]), "prev": as_bytes(f["prev_camera"])}


def run_envmap(lib, prefix, inp, mode, gamma, mip=1.0, alpha=0.0, scale=(1.5, 1.0, 0.75)):
    h, w = inp["depth"].shape
    color, motion = np.full((h, w, 4), -7.0, np.float32), np.full((h, w, 2), -7.0, np.float32)
    attr = tone_mapping_attribs_bytes(mode)
    if prefix == "ref_":
        lib.call("ref_envmap_ldr" if mode else "ref_envmap", [inp["env"], inp["depth"]], [color, motion], cam0=inp["cam"], cam1=inp["prev"], attribs=attr, fval=[0.3, mip, alpha, *scale])
    else:
        lib.call("oracle_envmap", [inp["env"], inp["depth"]], [color, motion], cam0=inp["cam"], cam1=inp["prev"], attribs=attr, fval=[0.3, mip, alpha, *scale], ival=[gamma, 1])
    return color, motion


@pytest.mark.parametrize("mode,gamma,mip", [(0, 0, 1.0), (0, 0, 0.0), (0, 0, 2.4), (4, 1, 1.0)])
def test_envmap_oracle_vs_ref(oracle, ref, mode, gamma, mip):
    inp = envmap_inputs()
    a = run_envmap(oracle, "oracle_", inp, mode, gamma, mip)
    b = run_envmap(ref, "ref_", inp, mode, gamma, mip)
    assert_close(a[0], b[0], rtol=1e-5, atol=1e-7, what="env map colour")
    assert_close(a[1], b[1], rtol=1e-5, atol=1e-7, what="env map motion")
    bg = inp["depth"] >= 1.0
    assert 0.05 < bg.mean() < 0.95
    assert (a[0][~bg] == -7.0).all() and (a[1][~bg] == -7.0).all() and (a[0][bg][:, 3] == 0.0).all() and (a[0][bg][:, :3] >= 0).all()
    # the background at infinity moves with the camera rotation only: its motion vectors are small but not zero
    assert 0 < np.abs(a[1][bg]).max() < 0.2


def test_ssao_half_precision_depth_permutation(oracle, ref):
    """FEATURE_FLAG_HALF_PRECISION_DEPTH of A3: the self-occlusion offset (SSAO_ComputeAmbientOcclusion.fx:145-150); oracle flag vs the reference permutation."""
    import cpu_chain
    from diligentfx_amd.binding import SSAOAttribs

    f = small_frame(frame=5, w=120, h=72)
    depth, normal = f["depth"].numpy(), f["normal"].numpy()
    from diligentfx_amd.binding import as_bytes

    cam = as_bytes(f["camera"])
    ab = bytes(SSAOAttribs.default())
    co = cpu_chain.CpuChain(oracle, "oracle_")
    pf = co.postfx(5, depth, f["prev_depth"].numpy(), f["motion"].numpy(), cam, as_bytes(f["prev_camera"]), blue_noise_tables())
    pyr = [depth]
    for k in range(1, 5):
        o = np.zeros((max(72 >> k, 1), max(120 >> k, 1)), np.float32)
        oracle.call("oracle_ssao_prefiltered_depth_mip", [pyr[k - 1]], [o], cam0=cam, attribs=ab, ival=[k - 1])
        pyr.append(o)
    a, b, plain = np.ones((72, 120), np.float32), np.ones((72, 120), np.float32), np.ones((72, 120), np.float32)
    oracle.call("oracle_ssao_compute_ao_gtao", [pyr, normal, pf["noise_zw"]], [a], cam0=cam, attribs=ab, ival=[0, 0, 0, 0, 0, 1])
    ref.call("ref_ssao_compute_ao_gtao_halfprec", [pyr, normal, pf["noise_zw"]], [b], cam0=cam, attribs=ab)
    oracle.call("oracle_ssao_compute_ao_gtao", [pyr, normal, pf["noise_zw"]], [plain], cam0=cam, attribs=ab)
    assert_close(a, b, rtol=2e-4, atol=1e-6, max_outlier_frac=2e-3, what="A3 half-precision-depth permutation")
    assert np.abs(a - plain).max() > 1e-3


# ------------------------------------------------------------------------------------------------ equirectangular ("sphere") environment maps
def sphere_map_mips(h=32, seed=11):
    """A 2h x h equirectangular HDR map (smooth sky gradient + a bright sun + noise) with its box-filtered mip chain."""
    rng = np.random.default_rng(seed)
    w = 2 * h
    v, u = np.meshgrid((np.arange(h) + 0.5) / h, (np.arange(w) + 0.5) / w, indexing="ij")
    sky = np.stack([0.3 + 0.5 * v, 0.4 + 0.4 * v, 0.9 - 0.3 * v], -1)
    sun = 300.0 * np.exp(-(((u - 0.7) * 2) ** 2 + (v - 0.65) ** 2) / 0.002)[..., None]
    img = np.concatenate([sky + sun + 0.05 * rng.random((h, w, 3)), np.ones((h, w, 1))], -1).astype(np.float32)
    mips = [np.ascontiguousarray(img)]
    while mips[-1].shape[0] > 1:
        m = mips[-1]
        mips.append(np.ascontiguousarray(m.reshape(m.shape[0] // 2, 2, m.shape[1] // 2, 2, 4).mean(axis=(1, 3)).astype(np.float32)))
    return mips


def test_ibl_precompute_from_sphere_map_oracle_vs_ref(oracle, ref):
    env = sphere_map_mips()
    for roughness in (0.0, 0.35, 1.0):
        a, b = np.zeros((6 * 16, 16, 4), np.float32), np.zeros((6 * 16, 16, 4), np.float32)
        oracle.call("oracle_ibl_prefilter_env_map", [env], [a], ival=[48, 1], fval=[roughness])
        ref.call("ref_ibl_prefilter_env_map_sphere", [env], [b], ival=[48], fval=[roughness])
        assert_close(a, b, rtol=1e-4, atol=1e-6, max_outlier_frac=1e-3, what=f"prefilter from a sphere map, roughness {roughness}")
    a, b = np.zeros((6 * 8, 8, 4), np.float32), np.zeros((6 * 8, 8, 4), np.float32)
    oracle.call("oracle_ibl_irradiance_map", [env], [a], ival=[256, 1])
    ref.call("ref_ibl_irradiance_map_sphere", [env], [b], ival=[256])
    assert_close(a, b, rtol=1e-4, atol=1e-6, max_outlier_frac=1e-3, what="irradiance from a sphere map")
    # roughness 0 = the equirect -> cube conversion: every cube texel is the bilinear lookup of its direction; the sun ends up on the +X / -Z side
    conv = np.zeros((6 * 16, 16, 4), np.float32)
    oracle.call("oracle_ibl_prefilter_env_map", [env], [conv], ival=[8, 1], fval=[0.0])
    assert conv[..., :3].max() > 50.0 and conv[..., :3].min() >= 0.0


def test_envmap_sphere_oracle_vs_ref(oracle, ref):
    inp = envmap_inputs()
    env = sphere_map_mips()
    h, w = inp["depth"].shape
    outs = []
    for lib, name, kw in ((oracle, "oracle_envmap", {"ival": [0, 1, 1]}), (ref, "ref_envmap_sphere", {})):
        color, motion = np.full((h, w, 4), -7.0, np.float32), np.full((h, w, 2), -7.0, np.float32)
        lib.call(name, [env, inp["depth"]], [color, motion], cam0=inp["cam"], cam1=inp["prev"], attribs=tone_mapping_attribs_bytes(0), fval=[0.3, 1.5, 0.0, 1.0, 1.0, 1.0], **kw)
        outs.append((color, motion))
    assert_close(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-7, what="sphere env map colour")
    assert_close(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-7, what="sphere env map motion")


# ------------------------------------------------------------------------------------------------ specular-glossiness workflow of P1 (PBR_Shading.fxh:93-117, 390-403)
def specgloss_material(material, base_color, seed=11):
    """A PhysicalDesc plane for the frame: rgb = specular colour in sRGB space (dielectric 0.2 ... metal-like tinted 0.9, and values below the 0.04 floor of
    SolveMetallic), a = glossiness = 1 - roughness of the frame's material."""
    rng = np.random.default_rng(seed)
    h, w = material.shape[:2]
    spec = np.where(material[..., 1:2] > 0.5, 0.55 + 0.4 * base_color[..., :3], 0.12 + 0.25 * rng.random((h, w, 1), dtype=np.float32)).astype(np.float32)
    spec[: h // 8] *= 0.2  # very dark specular: perceived brightness below c_MinReflectance -> Metallic = 0
    return np.ascontiguousarray(np.concatenate([spec, 1.0 - material[..., 0:1]], -1).astype(np.float32))


def test_pbr_shade_specular_glossiness_oracle_vs_ref(oracle, ref):
    import chain_util
    from diligentfx_amd.binding import as_bytes

    f = small_frame(frame=4, w=128, h=80)
    ibl = chain_util.make_ibl(oracle, "oracle_")
    g = {k: f[k].numpy() for k in ("base_color", "normal", "material", "depth")}
    desc = specgloss_material(g["material"], g["base_color"])
    sa = chain_util.shade_attribs(len(ibl["prefiltered"]) - 1)
    sa.Workflow = 1  # PBR_WORKFLOW_SPECULAR_GLOSSINESS
    ins = [g["base_color"], g["normal"], desc, g["depth"], None, None, ibl["lut"], ibl["irradiance"], ibl["prefiltered"]]
    outs = []
    for lib, pfx in ((oracle, "oracle_"), (ref, "ref_")):
        rad, spec, mat = np.zeros((80, 128, 4), np.float32), np.zeros((80, 128, 4), np.float32), np.zeros((80, 128, 4), np.float32)
        lib.call(pfx + "pbr_shade", ins, [rad, spec], cam0=as_bytes(f["camera"]), attribs=bytes(sa), fval=[0.02, 0.03, 0.05, 0.0])
        lib.call(pfx + "specgloss_material", [g["base_color"], desc], [mat])
        outs.append((rad, spec, mat))
    for k, what in enumerate(("radiance", "specular IBL", "material target")):
        assert_close(outs[0][k], outs[1][k], rtol=2e-4, atol=1e-6, what=f"specular-glossiness {what}")
    mat = outs[1][2]
    assert np.allclose(mat[..., 0], np.clip(1.0 - desc[..., 3], 0, 1), atol=1e-6) and (mat[: 80 // 8, :, 1] == 0).all() and mat[..., 1].max() > 0.5
    # the workflow matters: the same planes read as metallic-roughness give a different picture
    sa.Workflow = 0
    other = np.zeros((80, 128, 4), np.float32)
    ref.call("ref_pbr_shade", ins, [other, np.zeros_like(other)], cam0=as_bytes(f["camera"]), attribs=bytes(sa), fval=[0.02, 0.03, 0.05, 0.0])
    assert np.abs(other - outs[1][0]).max() > 0.05


# ------------------------------------------------------------------------------------------------ PCF shadows of the punctual lights
@pytest.mark.parametrize("pcf", [2, 3, 5, 7])
def test_pbr_shade_with_shadows_oracle_vs_ref(oracle, ref, pcf):
    import chain_util

    f = small_frame(frame=4, w=128, h=80)
    ibl = chain_util.make_ibl(oracle, "oracle_")
    from diligentfx_amd.binding import as_bytes

    g = {k: f[k].numpy() for k in ("base_color", "normal", "material", "depth")}
    sa = chain_util.shadowed_shade_attribs(len(ibl["prefiltered"]) - 1)
    slices, infos = chain_util.make_shadow_inputs()
    ins = [g["base_color"], g["normal"], g["material"], g["depth"], None, None, ibl["lut"], ibl["irradiance"], ibl["prefiltered"], slices, infos.reshape(1, -1)]
    outs = {}
    for lib, name, kw in ((oracle, "oracle_pbr_shade", {"ival": [pcf]}), (ref, f"ref_pbr_shade_shadows{pcf}", {})):
        rad, spec = np.zeros((80, 128, 4), np.float32), np.zeros((80, 128, 4), np.float32)
        lib.call(name, ins, [rad, spec], cam0=as_bytes(f["camera"]), attribs=bytes(sa), fval=[0.02, 0.03, 0.05, 0.0], **kw)
        outs[name] = rad
    a, b = outs["oracle_pbr_shade"], outs[f"ref_pbr_shade_shadows{pcf}"]
    # "reference < texel" is a threshold on computed numbers: a tap exactly on it may flip between the two builds
    assert_close(a, b, rtol=1e-5, atol=1e-7, max_outlier_frac=1e-3, what=f"shadowed shade, PCF {pcf}")
    plain = np.zeros((80, 128, 4), np.float32)
    oracle.call("oracle_pbr_shade", ins[:9], [plain, np.zeros((80, 128, 4), np.float32)], cam0=as_bytes(f["camera"]), attribs=bytes(sa), fval=[0.02, 0.03, 0.05, 0.0])
    darker = (a[..., :3] < plain[..., :3] - 1e-4).any(-1).mean()
    assert 0.05 < darker < 0.9 and (a[..., :3] <= plain[..., :3] + 1e-5).all()  # shadows only remove light, and not everywhere
