"""Shared by the tests of the shade with material layers (test_gpu_pbr_layers.py on the device, test_host_kernel_layers.py on the host): the frame, the layer planes, the
checker call.  The checker is oracle/_ref only -- the reference's PBR_Shading.fxh / PBR_Common.fxh / Iridescence.fxh compiled per permutation (oracle/ref/ref_pl_*.cpp)."""
import numpy as np
import pytest
import torch

PERMUTATIONS = {"clearcoat": 1, "sheen": 2, "anisotropy": 4, "iridescence": 8, "transmission": 16, "all": 31}  # checker permutation -> MIFX_PBR_LAYER_* set
CASES = [("clearcoat", (160, 96), True), ("clearcoat", (131, 77), False), ("sheen", (160, 96), True), ("anisotropy", (160, 96), True), ("anisotropy", (131, 77), False),
         ("iridescence", (160, 96), True), ("transmission", (131, 77), True), ("all", (160, 96), True), ("all", (131, 77), False)]
# with shadow-mapped lights as well (ENABLE_SHADOWS): (checker permutation, MIFX_PBR_LAYER_* set, PCF_FILTER_SIZE, size, optional planes bound)
SHADOW_CASES = [("all_shadows3", 31, 3, (224, 128), True), ("sheen_shadows5", 2, 5, (160, 96), False)]
LAYER_ORDER = ("clearcoat", "clearcoat_normal", "sheen", "anisotropy", "tangent", "iridescence", "transmission")
IOR, ROTATION, BACKGROUND = 1.33, 0.7, (0.02, 0.03, 0.05, 0.0)


def ref_checker():
    import pyref

    r = pyref.ref_lib()
    if r is None or not r.has("ref_pbr_shade_layers_all"):
        pytest.skip("the layered shade is checked against oracle/_ref only (the reference's shader source compiled here)")
    return r


def make_layers(normal, seed):
    """Per-pixel layer inputs of the frame (numpy, c = 4 each) and the two sheen look-up tables (32 x 32)."""
    h, w, _ = normal.shape
    rng = np.random.default_rng(seed)
    u = lambda *s: rng.random(s, dtype=np.float32)  # noqa: E731
    n = normal[..., :3]
    unit = lambda v: v / np.maximum(np.linalg.norm(v, axis=-1, keepdims=True), 1e-6)  # noqa: E731
    z = np.zeros((h, w, 1), np.float32)
    cc = np.concatenate([u(h, w, 1), 0.05 + 0.95 * u(h, w, 1), z, z], -1)
    ccn = np.concatenate([unit(n + 0.3 * (u(h, w, 3) - 0.5)), z], -1).astype(np.float32)
    sheen = np.concatenate([u(h, w, 3), 0.05 + 0.95 * u(h, w, 1)], -1)
    ang = 2 * np.pi * u(h, w, 1)
    an = np.concatenate([np.cos(ang), np.sin(ang), u(h, w, 1), z], -1).astype(np.float32)
    t = u(h, w, 3) - 0.5
    tan = np.concatenate([unit(t - n * (t * n).sum(-1, keepdims=True)), z], -1).astype(np.float32)
    thick = 100.0 + 300.0 * u(h, w, 1)
    thick[:, : w // 8] = 0.0  # a thickness of exactly 0 switches the layer off (RenderPBR.psh:250-251)
    thick[:, w // 8: w // 4] *= 1e-4  # and the smoothstep that blends the film's IOR in below 0.03 nm (Iridescence.fxh:58)
    ir = np.concatenate([u(h, w, 1), thick.astype(np.float32), z, z], -1)
    tr = np.concatenate([u(h, w, 1), z, z, z], -1)
    gy, gx = np.meshgrid(np.linspace(0, 1, 32, dtype=np.float32), np.linspace(0, 1, 32, dtype=np.float32), indexing="ij")
    albedo = (0.8 * (1 - gx) ** 2 * (0.3 + 0.7 * gy) + 0.05 * u(32, 32)).astype(np.float32)  # E(cos theta, roughness) in [0, 1): the shape of the real table, not its values
    charlie = (0.4 * gy * (1 - 0.5 * gx) + 0.03 * u(32, 32)).astype(np.float32)
    planes = {"clearcoat": cc, "clearcoat_normal": ccn, "sheen": sheen, "anisotropy": an, "tangent": tan, "iridescence": ir, "transmission": tr}
    return {k: np.ascontiguousarray(v, np.float32) for k, v in planes.items()}, albedo, charlie


def make_case(perm, size, ibl_np, device, shadowed=False, reversed_depth=False):
    """The frame of one test case: (frame dict of synth, G-buffer as numpy, shade attribs, layer planes as numpy, the two tables).  shadowed: the lights of
    chain_util.shadowed_shade_attribs (two of them with a shadow map: chain_util.make_shadow_inputs)."""
    import chain_util
    from diligentfx_amd import binding as B, synth

    w, h = size
    f = synth.make_frame(synth.Scene(), 4, w, h, device, reversed_depth=reversed_depth)
    gn = {k: f[k].cpu().numpy() for k in ("base_color", "normal", "material", "depth")}
    gen = torch.Generator(device="cpu").manual_seed(3)
    gn["emissive"] = (torch.rand(h, w, 4, generator=gen) * 0.3).numpy()
    gn["occlusion"] = (0.3 + 0.7 * torch.rand(h, w, generator=gen)).numpy()
    sa = (chain_util.shadowed_shade_attribs if shadowed else chain_util.shade_attribs)(len(ibl_np["prefiltered"]) - 1)
    sa.OcclusionStrength, sa.EmissionScale = 0.8, 1.5
    sa.IBLScale[:] = [1.1, 0.9, 1.0, 1.0]
    sa.Lights[sa.LightCount] = B.PBRLightAttribs(3, 2.0, 6.0, -3.0, -0.2, -0.9, 0.3, -1, 40.0, 35.0, 30.0, 20.0 ** 4, 8.0, -6.8, 0.0, 0.0)  # a spot light
    sa.LightCount += 1
    sa.Lights[sa.LightCount] = B.PBRLightAttribs(2, -3.0, 4.0, 2.0, 0.0, -1.0, 0.0, -1, 25.0, 30.0, 35.0, 0.0, 0.0, 0.0, 0.0, 0.0)  # a point light without a range
    sa.LightCount += 1
    planes, albedo, charlie = make_layers(gn["normal"], seed=sum(map(ord, perm)))
    return f, gn, sa, planes, albedo, charlie


def checker_result(lib, perm, optional, f, gn, sa, planes, albedo, charlie, ibl_np, shadows=None, reversed_depth=False):
    """(radiance, specular IBL) of the reference's permutation `perm`; optional: the clear-coat normal and the tangent planes are bound; shadows: (slices, infos)."""
    h, w = gn["depth"].shape
    wr, ws = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    luts = [np.repeat(albedo[..., None], 4, -1).copy(), np.repeat(charlie[..., None], 4, -1).copy()]
    lib.call("ref_pbr_shade_layers_" + perm, [gn["base_color"], gn["normal"], gn["material"], gn["depth"], gn["emissive"], gn["occlusion"], ibl_np["lut"], ibl_np["irradiance"],
                                               ibl_np["prefiltered"], [planes[k] for k in LAYER_ORDER], luts] + ([list(shadows[0]), shadows[1].reshape(1, -1)] if shadows else []),
             [wr, ws], cam0=bytes(f["camera"]), attribs=bytes(sa), ival=[int(optional), int(optional), 0, 0, 0, 0, 0, int(reversed_depth)], fval=list(BACKGROUND) + [IOR, ROTATION])
    return wr, ws


GOLDEN_CASES = [("clearcoat", False), ("sheen", False), ("anisotropy", True), ("iridescence", False), ("transmission", False), ("all", True), ("all_shadows3", True)]
GOLDEN_FLAGS = {**PERMUTATIONS, "all_shadows3": 31}


def load_layers_golden():
    """tests/golden/layers_golden.npz (generated from oracle/_ref by tests/golden/make_golden_layers.py): inputs and the reference's outputs per permutation."""
    import os

    from diligentfx_amd import binding as B
    from util import GOLDEN

    z = np.load(os.path.join(GOLDEN, "layers_golden.npz"))
    levels = len([k for k in z.files if k.startswith("ibl_prefiltered")])
    ibl = {"lut": z["ibl_lut"], "irradiance": [z["ibl_irradiance"]], "prefiltered": [z[f"ibl_prefiltered{i}"] for i in range(levels)]}
    gn = {k[2:]: z[k] for k in z.files if k.startswith("g_")}
    planes = {k[6:]: z[k] for k in z.files if k.startswith("layer_")}
    attribs = {False: B.PBRShadeAttribs.from_buffer_copy(z["shade_attribs"].tobytes()), True: B.PBRShadeAttribs.from_buffer_copy(z["shade_attribs_shadowed"].tobytes())}
    out = {perm: (z[f"out_{perm}_radiance"], z[f"out_{perm}_specular_ibl"]) for perm, _ in GOLDEN_CASES}
    return dict(ibl=ibl, gn=gn, planes=planes, albedo=z["lut_albedo_scaling"], charlie=z["lut_charlie"], camera=z["camera"].tobytes(), attribs=attribs,
                shadows=([s for s in z["shadow_slices"]], z["shadow_infos"]), out=out)
