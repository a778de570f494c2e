"""Two header-only pieces of the timed chain, compiled for the HOST and compared with the reference bit for bit (the method of tests/test_host_kernel_layers.py):
ToneMap() in all twelve modes with and without the sRGB conversion (mifx_tonemap.h -- the body of tonemap_kernel and of the tail fused into Bloom's final up-sample and
the composite), and SSR's bilateral cleanup R7 (mifx_ssr_cleanup.h -- the body of ssr_bilateral_kernel and of the composite's fused variant) on the planes of a CPU chain
that has run three frames.  Test infrastructure: the product never builds, loads or calls this."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def ref_checker():
    import pyref

    r = pyref.ref_lib()
    if r is None:
        pytest.skip("compared with oracle/_ref (the reference's shader source compiled here)")
    return r


@pytest.fixture(scope="module")
def host_lib():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(HERE, "host_kernels", "chain_host.cpp")
    out_dir = os.path.join(HERE, "host_kernels", "_build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "chain_host.so")
    deps = [src, os.path.join(ROOT, "include", "mifx.h")] + [os.path.join(ROOT, "diligentfx_amd", "csrc", n) for n in ("mifx_tonemap.h", "mifx_ssr_cleanup.h", "mifx_composite.h", "mifx_ssr_temporal.h", "mifx_pbr.h", "mifx_effects.h", "mifx_device.h")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        cmd = [hipcc, "-x", "hip", "--cuda-host-only", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fopenmp", "-I", os.path.join(ROOT, "diligentfx_amd", "csrc"), "-I",
               os.path.join(ROOT, "include"), "-o", out, src]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-4000:]
    return ctypes.CDLL(out)


def fptr(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


@pytest.mark.parametrize("mode", range(12))
def test_tonemap_source_on_the_host_is_bit_exact(host_lib, mode):
    from diligentfx_amd import binding as B

    lib = ref_checker()
    rng = np.random.default_rng(100 + mode)
    h, w = 48, 64
    img = (rng.random((h, w, 4), dtype=np.float32) ** 3 * 24.0).astype(np.float32)  # HDR: most values small, a few above the white point
    img[0, :8, :3] = 0.0      # black
    img[1, :8, :3] *= -1.0    # negative input (clamped by the operator)
    img[2, :8, :3] = 1e-12    # the luminance floor
    attribs = B.ToneMappingAttribs.default(mode)
    for srgb in (0, 1):
        want, got = np.zeros_like(img), np.zeros_like(img)
        lib.call("ref_tonemap", [img], [want], attribs=bytes(attribs), fval=[0.3], ival=[srgb])
        assert host_lib.mifx_host_tonemap(fptr(img), fptr(got), w, h, ctypes.byref(attribs), ctypes.c_float(0.3), srgb) == 0
        # (a black pixel is 0 / 0 in the operators that divide by the pixel's luminance or take its logarithm: the reference returns NaN there, and so does the kernel)
        assert np.isfinite(want).mean() > 0.98
        assert np.array_equal(got, want, equal_nan=True), f"mode {mode} srgb {srgb}: {(got != want).mean():.2e} of the values differ"


def test_composite_source_on_the_host_is_bit_exact(host_lib):
    """M1 (mifx_composite.h) on the planes of a CPU chain, frame by frame: with the reflection read from R7's output plane, and with R7 evaluated in place from SSR's accumulated
    radiance / variance (the chain's fused instance) -- both give the reference's composite, bit for bit."""
    import chain_util
    import cpu_chain
    from diligentfx_amd import binding as B, synth

    lib = ref_checker()
    w, h = 128, 72
    ibl = chain_util.make_ibl(lib, "ref_")
    cpu = cpu_chain.CpuChain(lib, "ref_")
    scene = synth.Scene()
    a = B.SSRAttribs.default()
    lut = np.ascontiguousarray(ibl["lut"])
    null = ctypes.POINTER(ctypes.c_float)()
    for frame in range(3):
        keep = {}
        chain_util.run_frame(cpu, scene, frame, w, h, ibl, keep)
        g = keep["gbuffer"]
        cam = B.camera_from_bytes(keep["camera"])
        common = (fptr(keep["ssao_out"]), fptr(g["normal"]), fptr(g["base_color"]), fptr(g["material"]), fptr(lut), lut.shape[1], lut.shape[0], lut.shape[2])
        r7 = (fptr(g["depth"]), fptr(keep["ssr_roughness"]), fptr(keep["ssr_hist_rad"]), fptr(keep["ssr_hist_var"]), fptr(keep["ssr_mask"]), ctypes.c_float(a.RoughnessThreshold),
              ctypes.c_float(a.BilateralCleanupSpatialSigmaFactor), ctypes.c_float(a.AlphaInterpolation))
        for fused in (False, True):
            got = np.zeros((h, w, 4), np.float32)
            rc = host_lib.mifx_host_composite(fptr(keep["radiance"]), fptr(keep["specular_ibl"]), null if fused else fptr(keep["ssr_out"]), *common, fptr(got), w, h, ctypes.byref(cam),
                                              ctypes.c_float(1.0), ctypes.c_float(1.0), *r7)
            assert rc == 0
            want = keep["composite"]
            assert np.array_equal(got, want), f"frame {frame} fused {fused}: {(got != want).mean():.2e} of the values differ, max {np.abs(got - want).max():.3e}"
        assert (np.abs(keep["composite"] - keep["radiance"]) > 1e-3).mean() > 0.05  # reflections and occlusion are in the picture


def test_ssr_temporal_accumulation_source_on_the_host_is_bit_exact(host_lib):
    """R6 (mifx_ssr_temporal.h) on the planes of a CPU chain over five frames: the reprojection of both candidates, the disocclusion test, the 3x3 search of a disoccluded
    pixel, the clamp to the neighbourhood's statistics, and the texels outside the mask, which keep what the frame before last wrote."""
    import chain_util
    import cpu_chain
    from diligentfx_amd import binding as B, synth

    lib = ref_checker()
    w, h = 128, 72
    ibl = chain_util.make_ibl(lib, "ref_")
    cpu = cpu_chain.CpuChain(lib, "ref_")
    scene = synth.Scene()
    a = B.SSRAttribs.default()
    accumulated = 0
    for frame in range(5):
        cur = frame & 1
        before = None
        if cpu.ssr_hist is not None and frame > 0:
            before = (cpu.ssr_hist["rad"][cur].copy(), cpu.ssr_hist["var"][cur].copy())
        keep = {}
        chain_util.run_frame(cpu, scene, frame, w, h, ibl, keep)
        if before is None:
            continue  # (frame 0 allocates the history: nothing to compare the slot's previous content with)
        g, pf = keep["gbuffer"], keep["postfx"]
        got_rad, got_var = before[0].copy(), before[1].copy()
        prev_rad, prev_var = cpu.ssr_hist["rad"][cur ^ 1], cpu.ssr_hist["var"][cur ^ 1]  # (R6 of this frame wrote the other slot only)
        cam, prev_cam = B.camera_from_bytes(keep["camera"]), B.camera_from_bytes(keep["prev_camera"])
        rc = host_lib.mifx_host_ssr_temporal(fptr(g["motion"]), fptr(keep["ssr_res_depth"]), fptr(pf["reproj_depth"]), fptr(keep["ssr_res_rad"]), fptr(keep["ssr_res_var"]),
                                             fptr(pf["prev_depth"]), fptr(prev_rad), fptr(prev_var), fptr(keep["ssr_mask"]), fptr(got_rad), fptr(got_var), w, h, ctypes.byref(cam),
                                             ctypes.byref(prev_cam), ctypes.byref(a))
        assert rc == 0
        want_rad, want_var = keep["ssr_hist_rad"], keep["ssr_hist_var"]
        assert np.array_equal(got_rad, want_rad), f"frame {frame}: {(got_rad != want_rad).mean():.2e} of the radiance values differ, max {np.abs(got_rad - want_rad).max():.3e}"
        assert np.array_equal(got_var, want_var), f"frame {frame}: {(got_var != want_var).mean():.2e} of the variance values differ"
        m = keep["ssr_mask"] != 0
        accumulated += int(((want_var != 1.0) & m).sum())
    assert accumulated > 500  # pixels whose history was accepted (variance blended, not reset to 1)


def test_ssr_bilateral_cleanup_source_on_the_host_is_bit_exact(host_lib):
    import chain_util
    import cpu_chain
    from diligentfx_amd import binding as B, synth

    lib = ref_checker()
    w, h = 128, 72
    ibl = chain_util.make_ibl(lib, "ref_")
    cpu = cpu_chain.CpuChain(lib, "ref_")
    scene = synth.Scene()
    a = B.SSRAttribs.default()
    filtered = 0
    for frame in range(3):
        keep = {}
        chain_util.run_frame(cpu, scene, frame, w, h, ibl, keep)
        g = keep["gbuffer"]
        cam = B.camera_from_bytes(keep["camera"])
        proj = np.array(list(cam.mProj), np.float32)
        got = np.zeros((h, w, 4), np.float32)
        rc = host_lib.mifx_host_ssr_bilateral_cleanup(fptr(g["depth"]), fptr(g["normal"]), fptr(keep["ssr_roughness"]), fptr(keep["ssr_hist_rad"]), fptr(keep["ssr_hist_var"]),
                                                      fptr(keep["ssr_mask"]), fptr(got), w, h, fptr(proj), ctypes.c_float(a.RoughnessThreshold),
                                                      ctypes.c_float(a.BilateralCleanupSpatialSigmaFactor), ctypes.c_float(a.AlphaInterpolation), 0)
        assert rc == 0
        want = keep["ssr_out"]
        assert np.array_equal(got, want), f"frame {frame}: {(got != want).mean():.2e} of the values differ, max {np.abs(got - want).max():.3e}"
        filtered += int(((want != keep["ssr_hist_rad"] * np.array([1, 1, 1, a.AlphaInterpolation], np.float32)).any(-1) & (keep["ssr_mask"] != 0)).sum())
    assert filtered > 100  # the filter ran (pixels whose output is not the plain copy of the accumulated radiance)
