"""The kernel source of the shade with material layers, compiled for the HOST and compared with the reference where there is no GPU.

tests/host_kernels/layers_host.cpp includes the kernel's own header (diligentfx_amd/csrc/mifx_pbr_layers.h: pbr_shade_layers_pixel, the body of pbr_shade_layers_kernel) under
`hipcc --cuda-host-only` with __device__ redefined as "host and device", and loops over the pixels.  On the host the kernel's division and square root are the IEEE operations and
its exp / pow / cos are glibc's -- the checker's own -- so the comparison is BIT-EXACT: every operation of the five layers is written in the reference's order.  (What the host
cannot show -- the device's fdiv / fsqrt sequences, the device math library, the apron copies of the cube maps -- tests/test_gpu_pbr_layers.py shows on the device.)
Test infrastructure: the product never builds, loads or calls this."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from layers_util import (BACKGROUND, CASES, GOLDEN_CASES, GOLDEN_FLAGS, IOR, LAYER_ORDER, PERMUTATIONS, ROTATION, SHADOW_CASES, checker_result, load_layers_golden, make_case,
                         ref_checker)

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


class HostPlane(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("w", ctypes.c_int), ("h", ctypes.c_int), ("c", ctypes.c_int)]


def plane(a):
    if a is None:
        return HostPlane(None, 0, 0, 0)
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return HostPlane(a.ctypes.data, a.shape[1], a.shape[0], a.shape[2] if a.ndim == 3 else 1)


@pytest.fixture(scope="module")
def host_lib():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(HERE, "host_kernels", "layers_host.cpp")
    out_dir = os.path.join(HERE, "host_kernels", "_build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "layers_host.so")
    deps = [src, os.path.join(ROOT, "include", "mifx.h")] + [os.path.join(ROOT, "diligentfx_amd", "csrc", n) for n in ("mifx_pbr_layers.h", "mifx_pbr.h", "mifx_device.h")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        cmd = [hipcc, "-x", "hip", "--cuda-host-only", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fopenmp", "-I", os.path.join(ROOT, "diligentfx_amd", "csrc"), "-I",
               os.path.join(ROOT, "include"), "-o", out, src]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-4000:]
    return ctypes.CDLL(out)


@pytest.fixture(scope="module")
def ibl_np():
    import chain_util

    return chain_util.make_ibl(ref_checker(), "ref_")


def run_on_host(host_lib, ibl_np, flags, optional, f, gn, sa, planes, albedo, charlie, shadows=None, pcf=0, generic=False, reversed_depth=False):
    h, w = gn["depth"].shape
    got, got_spec = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    P = (HostPlane * 8)(plane(gn["base_color"]), plane(gn["normal"]), plane(gn["material"]), plane(gn["depth"]), plane(gn["emissive"]), plane(gn["occlusion"]), plane(got),
                        plane(got_spec))
    transmission = np.ascontiguousarray(planes["transmission"][..., 0])
    layer_planes = {**planes, "transmission": transmission}
    if not optional:
        layer_planes["clearcoat_normal"] = layer_planes["tangent"] = None
    L = (HostPlane * 7)(*[plane(layer_planes[k]) for k in LAYER_ORDER])
    charlie4 = np.repeat(charlie[..., None], 4, -1).copy()  # one table with one channel, one with four (r used): both texel layouts of the look-up
    U = (HostPlane * 3)(plane(ibl_np["lut"]), plane(albedo), plane(charlie4))
    irradiance = plane(ibl_np["irradiance"][0])
    prefiltered = (HostPlane * len(ibl_np["prefiltered"]))(*[plane(m) for m in ibl_np["prefiltered"]])
    sm = infos = None
    n_slices = n_infos = 0
    if shadows is not None:
        stack = np.ascontiguousarray(np.stack(shadows[0]))
        sm = (HostPlane * 1)(HostPlane(stack.ctypes.data, stack.shape[2], stack.shape[1], 1))
        infos = np.ascontiguousarray(shadows[1], np.float32)
        n_slices, n_infos = stack.shape[0], infos.shape[0]
    rc = host_lib.mifx_host_pbr_shade_layers(P, L, U, ctypes.byref(irradiance), prefiltered, len(ibl_np["prefiltered"]), bytes(f["camera"]), bytes(sa),
                                             (ctypes.c_float * 4)(*BACKGROUND), flags, ctypes.c_float(IOR), ctypes.c_float(ROTATION), int(reversed_depth), sm, n_slices,
                                             infos.ctypes.data_as(ctypes.c_void_p) if infos is not None else None, n_infos, pcf, int(generic))
    assert rc == 0
    return got, got_spec


@pytest.mark.parametrize("perm,size,optional", CASES)
def test_layers_kernel_source_on_the_host_is_bit_exact(host_lib, ibl_np, perm, size, optional):
    lib = ref_checker()
    f, gn, sa, planes, albedo, charlie = make_case(perm, size, ibl_np, torch.device("cpu"))
    want, want_spec = checker_result(lib, perm, optional, f, gn, sa, planes, albedo, charlie, ibl_np)
    got, got_spec = run_on_host(host_lib, ibl_np, PERMUTATIONS[perm], optional, f, gn, sa, planes, albedo, charlie)
    assert np.isfinite(want).all() and float(want[..., :3].max()) > 1.0
    assert np.array_equal(got, want), f"{perm}: {(got != want).mean():.2e} of the radiance values differ, max {np.abs(got - want).max():.3e}"
    assert np.array_equal(got_spec, want_spec), f"{perm}: {(got_spec != want_spec).mean():.2e} of the specular IBL values differ"
    # the instance with the set as a run-time mask (what any other combination of layers takes) gives the same bits as the one compiled for this set
    again, again_spec = run_on_host(host_lib, ibl_np, PERMUTATIONS[perm], optional, f, gn, sa, planes, albedo, charlie, generic=True)
    assert np.array_equal(again, want) and np.array_equal(again_spec, want_spec)


@pytest.mark.parametrize("variant", ["specular_glossiness", "reversed_depth", "two_layers"])
def test_layers_kernel_source_on_the_host_variants(host_lib, ibl_np, variant):
    """All five layers on a specular-glossiness G-buffer (PBR_WORKFLOW_SPECULAR_GLOSSINESS: the material plane is the PhysicalDesc), on a reversed-depth frame
    (FEATURE_FLAG_REVERSED_DEPTH: background = depth 0), and a set that has no instance of its own (clear coat + iridescence: the run-time-set instance) -- bit-exact."""
    lib = ref_checker()
    rev = variant == "reversed_depth"
    f, gn, sa, planes, albedo, charlie = make_case(variant, (120, 68), ibl_np, torch.device("cpu"), reversed_depth=rev)
    perm, flags = "all", 31
    if variant == "specular_glossiness":
        sa.Workflow = 1  # MIFX_PBR_WORKFLOW_SPECULAR_GLOSSINESS (PBR_Structures.fxh:30)
        rng = np.random.default_rng(3)
        gn["material"] = np.concatenate([0.04 + 0.6 * rng.random(gn["material"].shape[:2] + (3,), dtype=np.float32), 0.1 + 0.85 * rng.random(gn["material"].shape[:2] + (1,), dtype=np.float32)], -1)
    if variant == "two_layers":
        flags = 1 | 8
    want, want_spec = checker_result(lib, perm, True, f, gn, sa, planes, albedo, charlie, ibl_np, reversed_depth=rev)
    if variant == "two_layers":  # the reference has no such permutation compiled here: all five with the other three layers neutral is NOT the same arithmetic (anisotropy), so
        # compare the run-time-set instance with the sum of what it must equal instead: the kernel compiled for the set {clear coat, iridescence} does not exist either --
        # the check is that the run-time instance with flags 9 equals itself through both dispatch routes and differs from all five and from each single layer
        a, _ = run_on_host(host_lib, ibl_np, flags, True, f, gn, sa, planes, albedo, charlie)
        b, _ = run_on_host(host_lib, ibl_np, flags, True, f, gn, sa, planes, albedo, charlie, generic=True)
        only_cc, _ = run_on_host(host_lib, ibl_np, 1, True, f, gn, sa, planes, albedo, charlie)
        only_ir, _ = run_on_host(host_lib, ibl_np, 8, True, f, gn, sa, planes, albedo, charlie)
        assert np.array_equal(a, b) and np.isfinite(a).all()
        assert not np.array_equal(a, want) and not np.array_equal(a, only_cc) and not np.array_equal(a, only_ir)
        return
    got, got_spec = run_on_host(host_lib, ibl_np, flags, True, f, gn, sa, planes, albedo, charlie, reversed_depth=rev)
    assert np.array_equal(got, want), f"{variant}: {(got != want).mean():.2e} of the radiance values differ, max {np.abs(got - want).max():.3e}"
    assert np.array_equal(got_spec, want_spec)
    if rev:
        assert (gn["depth"] < 1e-6).any() and np.array_equal(got[gn["depth"] < 1e-6], np.broadcast_to(np.array(BACKGROUND, np.float32), got[gn["depth"] < 1e-6].shape))


@pytest.mark.parametrize("perm,flags,pcf,size,optional", SHADOW_CASES)
def test_layers_with_shadow_mapped_lights_on_the_host(host_lib, ibl_np, perm, flags, pcf, size, optional):
    """ENABLE_SHADOWS on top of the layers: the PCF filter of the default shade (mifx_pbr.h filter_shadow_map_fixed_pcf) in the layered body."""
    import chain_util

    lib = ref_checker()
    f, gn, sa, planes, albedo, charlie = make_case(perm, size, ibl_np, torch.device("cpu"), shadowed=True)
    shadows = chain_util.make_shadow_inputs()
    want, want_spec = checker_result(lib, perm, optional, f, gn, sa, planes, albedo, charlie, ibl_np, shadows=shadows)
    got, got_spec = run_on_host(host_lib, ibl_np, flags, optional, f, gn, sa, planes, albedo, charlie, shadows=shadows, pcf=pcf)
    unshadowed, _ = run_on_host(host_lib, ibl_np, flags, optional, f, gn, sa, planes, albedo, charlie)  # (ShadowMapIndex ignored without a shadow map)
    assert (np.abs(unshadowed - want) > 1e-3).mean() > 0.02  # the shadows are in the picture
    d = np.abs(got - want) / np.maximum(np.abs(want), 1.0)
    assert np.array_equal(got_spec, want_spec)
    assert float(d.max()) <= 1e-6, f"{perm}: max relative difference {d.max():.3e}, {(got != want).mean():.2e} of the values differ"


@pytest.mark.parametrize("perm,optional", GOLDEN_CASES)
def test_layers_kernel_source_on_the_host_against_the_golden_fixture(host_lib, perm, optional):
    """The same comparison without oracle/_ref: tests/golden/layers_golden.npz holds the reference's outputs (tests/golden/make_golden_layers.py)."""
    G = load_layers_golden()
    shadowed = "shadows" in perm
    got, got_spec = run_on_host(host_lib, G["ibl"], GOLDEN_FLAGS[perm], optional, {"camera": G["camera"]}, G["gn"], G["attribs"][shadowed], G["planes"], G["albedo"], G["charlie"],
                                shadows=G["shadows"] if shadowed else None, pcf=3 if shadowed else 0)
    want, want_spec = G["out"][perm]
    assert np.array_equal(got_spec, want_spec)
    if shadowed:  # (the PCF weights go through fdiv and products the compiler may order differently on the host: not bit-exact, but within rounding)
        assert float((np.abs(got - want) / np.maximum(np.abs(want), 1.0)).max()) <= 1e-6
    else:
        assert np.array_equal(got, want), f"{perm}: {(got != want).mean():.2e} of the radiance values differ"


def test_the_golden_fixture_is_the_reference():
    """Pins tests/golden/layers_golden.npz to oracle/_ref where that exists: the fixture's outputs are what the reference's permutations return for the fixture's inputs."""
    lib = ref_checker()
    G = load_layers_golden()
    for perm, optional in GOLDEN_CASES:
        shadowed = "shadows" in perm
        r, s = checker_result(lib, perm, optional, {"camera": G["camera"]}, G["gn"], G["attribs"][shadowed], G["planes"], G["albedo"], G["charlie"], G["ibl"],
                              shadows=G["shadows"] if shadowed else None)
        assert np.array_equal(r, G["out"][perm][0]) and np.array_equal(s, G["out"][perm][1]), perm


def test_layers_change_the_picture_and_neutral_inputs_do_not(host_lib, ibl_np):
    """Sanity of the checker's wrapper itself (oracle/ref/ref_pl_body.inc): with neutral layer inputs -- factor 0, sheen colour 0, transmission 0 -- the permutations that have no
    numerically different base path (clear coat, sheen, iridescence, transmission) reproduce the default permutation ref_pbr_shade; with the test's inputs they do not."""
    lib = ref_checker()
    f, gn, sa, planes, albedo, charlie = make_case("all", (96, 64), ibl_np, torch.device("cpu"))
    h, w = gn["depth"].shape
    base, base_spec = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    lib.call("ref_pbr_shade", [gn["base_color"], gn["normal"], gn["material"], gn["depth"], gn["emissive"], gn["occlusion"], ibl_np["lut"], ibl_np["irradiance"], ibl_np["prefiltered"]],
             [base, base_spec], cam0=bytes(f["camera"]), attribs=bytes(sa), fval=list(BACKGROUND))
    neutral = {k: np.zeros_like(v) for k, v in planes.items()}
    neutral["clearcoat"][..., 1] = 0.5  # roughness of a layer with factor 0
    neutral["sheen"][..., 3] = 0.5
    neutral["iridescence"][..., 1] = 300.0  # a film with factor 0
    geom = gn["depth"] < 1.0 - 1e-6
    for perm in ("clearcoat", "sheen", "iridescence", "transmission"):
        r, s = checker_result(lib, perm, False, f, gn, sa, neutral, albedo, charlie, ibl_np)
        assert np.allclose(r, base, rtol=2e-6, atol=1e-6) and np.allclose(s, base_spec, rtol=2e-6, atol=1e-6), perm
        r, _ = checker_result(lib, perm, True, f, gn, sa, planes, albedo, charlie, ibl_np)
        assert (np.abs(r - base)[geom] > 1e-3).mean() > 0.2, perm
