"""Test infrastructure: builds small IBL inputs and runs the whole reference chain on the CPU through oracle/cpu_chain.py."""
import ctypes

import numpy as np
import torch

import cpu_chain
from diligentfx_amd import binding as B, synth
from util import blue_noise_tables


def box_mips(cube):
    """cube: (6*n, n, 4) -> list of mips down to 1x1 by 2x2 box averaging (how the application builds the env-map mip chain)."""
    n = cube.shape[1]
    mips = [np.ascontiguousarray(cube)]
    faces = cube.reshape(6, n, n, 4)
    while n > 1:
        faces = faces.reshape(6, n // 2, 2, n // 2, 2, 4).mean(axis=(2, 4)).astype(np.float32)
        n //= 2
        mips.append(np.ascontiguousarray(faces.reshape(6 * n, n, 4)))
    return mips


def make_ibl(lib, prefix, env_size=32, lut_size=32, irr_size=8, pref_size=16, lut_samples=64, irr_samples=128, pref_samples=32):
    env = synth.make_sky_cube(env_size, torch.device("cpu")).numpy()
    env = np.minimum(env, 200.0).astype(np.float32)  # tame the sun for the tiny sample counts used in tests
    env_mips = box_mips(env)
    lut = np.zeros((lut_size, lut_size, 2), np.float32)
    lib.call(prefix + "ibl_brdf_lut", [], [lut], ival=[lut_samples])
    irr = np.zeros((6 * irr_size, irr_size, 4), np.float32)
    lib.call(prefix + "ibl_irradiance_map", [env_mips], [irr], ival=[irr_samples])
    pref = []
    levels = int(np.log2(pref_size)) + 1
    for m in range(levels):
        s = pref_size >> m
        o = np.zeros((6 * s, s, 4), np.float32)
        lib.call(prefix + "ibl_prefilter_env_map", [env_mips], [o], ival=[pref_samples], fval=[m / (levels - 1)])
        pref.append(o)
    return {"env": env_mips, "lut": lut, "irradiance": [irr], "prefiltered": pref}


def shade_attribs(last_mip):
    a = synth.make_lights()
    a.PrefilteredCubeLastMip = float(last_mip)
    return a


def run_frame(chain: cpu_chain.CpuChain, scene, frame_index, w, h, ibl, keep=None, tonemap_mode=4):
    """One frame of the canonical chain (HnPostProcessTask order): shade -> prep -> SSR -> SSAO -> composite -> TAA -> Bloom -> ToneMap."""
    f = synth.make_frame(scene, frame_index, w, h, torch.device("cpu"), reversed_depth=getattr(chain, "reversed_depth", False))
    g = {k: v.numpy() for k, v in f.items() if isinstance(v, torch.Tensor)}
    return run_frame_inputs(chain, g, bytes(f["camera"]), bytes(f["prev_camera"]), frame_index, ibl, shade_attribs(len(ibl["prefiltered"]) - 1), keep, tonemap_mode)


def run_frame_inputs(chain: cpu_chain.CpuChain, g, cam, prev, frame_index, ibl, sa, keep=None, tonemap_mode=4, shade=None):
    """Same as run_frame, on caller-provided inputs (g: dict of numpy planes; cam / prev: CameraAttribs bytes; sa: PBRShadeAttribs).  shade: a callable (g, cam, sa) ->
    (radiance, specular IBL) in place of the default permutation of the shade (the layered permutations: tests/test_gpu_pbr_layers.py)."""
    h, w = g["depth"].shape
    radiance, spec_ibl = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    if shade is not None:
        radiance, spec_ibl = shade(g, cam, sa)
    else:
        chain.call("pbr_shade", [g["base_color"], g["normal"], g["material"], g["depth"], None, None, ibl["lut"], ibl["irradiance"], ibl["prefiltered"]],
                   [radiance, spec_ibl], cam0=cam, attribs=bytes(sa), fval=[0.02, 0.03, 0.05, 0.0])
    pf = chain.postfx(frame_index, g["depth"], g["prev_depth"], g["motion"], cam, prev, blue_noise_tables())
    ssr = chain.ssr(pf, radiance, g["depth"], g["normal"], g["material"], g["motion"], B.SSRAttribs.default(), keep)
    ssao = chain.ssao(pf, g["depth"], g["normal"], B.SSAOAttribs.default(), keep)
    comp = np.zeros((h, w, 4), np.float32)
    chain.call("composite", [radiance, spec_ibl, ssr, ssao, g["normal"], g["base_color"], g["material"], ibl["lut"]], [comp], cam0=cam, fval=[1.0, 1.0])
    taa = chain.taa(pf, comp, B.TAAAttribs.default(), keep)
    bloom = chain.bloom(taa, B.BloomAttribs.default(), keep)
    final = np.zeros((h, w, 4), np.float32)
    chain.call("tonemap", [bloom], [final], attribs=bytes(B.ToneMappingAttribs.default(tonemap_mode)), fval=[0.3], ival=[1])
    if keep is not None:
        keep.update({"gbuffer": g, "camera": cam, "prev_camera": prev, "radiance": radiance, "specular_ibl": spec_ibl, "postfx": pf, "composite": comp,
                     "final": final, "shade_attribs": sa})
    return final


def load_golden():
    """tests/golden/chain_golden.npz (generated from oracle/_ref by tests/golden/make_golden_chain.py)."""
    import os

    from util import GOLDEN

    z = np.load(os.path.join(GOLDEN, "chain_golden.npz"))
    levels = len([k for k in z.files if k.startswith("prefiltered")])
    ibl = {"lut": z["lut"], "irradiance": [z["irradiance"]], "prefiltered": [z[f"prefiltered{i}"] for i in range(levels)]}
    frames = []
    i = 0
    while f"f{i}_out_final" in z.files:
        frames.append({"in": {k: z[f"f{i}_in_{k}"] for k in ("depth", "normal", "base_color", "material", "motion", "prev_depth")},
                       "camera": z[f"f{i}_camera"].tobytes(), "prev_camera": z[f"f{i}_prev_camera"].tobytes(),
                       "out": {k: z[f"f{i}_out_{k}"] for k in ("radiance", "specular_ibl", "ssao_out", "ssr_out", "composite", "taa_out", "bloom_out", "final")}})
        i += 1
    raw = z["shade_attribs"].tobytes()  # (fixtures written before the Workflow field existed are shorter: zero = metallic-roughness)
    sa = B.PBRShadeAttribs.from_buffer_copy(raw + bytes(max(ctypes.sizeof(B.PBRShadeAttribs) - len(raw), 0)))
    return ibl, frames, sa


def make_shadow_inputs(size=64, seed=21):
    """Inputs of the shadowed shade (ENABLE_SHADOWS): a two-slice shadow-map array with depth relief around 0.5 and the PBRShadowMapInfo array (24 floats each: an
    orthographic world -> light-clip matrix over the synthetic scene, UV scale / bias, slice).  The maps are not rendered from the scene: parity of the comparison
    filter does not need them to be, only both outcomes of every comparison."""
    rng = np.random.default_rng(seed)
    v, u = np.meshgrid((np.arange(size) + 0.5) / size, (np.arange(size) + 0.5) / size, indexing="ij")
    slices = [np.ascontiguousarray((0.5 + 0.2 * np.sin(9.0 * u + k) * np.cos(7.0 * v - k) + 0.03 * rng.standard_normal((size, size))).astype(np.float32)) for k in range(2)]
    infos = np.zeros((2, 24), np.float32)
    for k in range(2):
        # rows of a row-vector matrix: x' = x / 14, y' = z / 14, z' = 0.5 + (y - 1.5) * 0.12 + 0.01 * x (a sheared height), w' = 1
        m = np.zeros((4, 4), np.float32)
        m[0, 0], m[2, 1], m[1, 2], m[0, 2], m[3, 2], m[3, 3] = 1 / 14.0, 1 / 14.0, 0.12, 0.01 * (1 - 2 * k), 0.5 - 1.5 * 0.12, 1.0
        infos[k, :16] = m.reshape(-1)
        infos[k, 16:20] = [0.9, 0.9, 0.05, 0.05]  # UVScale, UVBias
        infos[k, 20] = float(k)                    # ShadowMapSlice
    return slices, np.ascontiguousarray(infos)


def shadowed_shade_attribs(last_mip):
    """The lights of shade_attribs() with the directional light on shadow map 0 and a spot light on shadow map 1."""
    a = shade_attribs(last_mip)
    a.Lights[0].ShadowMapIndex = 0
    a.Lights[a.LightCount] = B.PBRLightAttribs(3, 2.0, 9.0, -3.0, 0.0, -1.0, 0.1, 1, 40.0, 36.0, 30.0, 30.0 ** 4, 2.0, -1.2, 0.0, 0.0)
    a.LightCount += 1
    return a
