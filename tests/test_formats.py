"""Native texture formats at the boundary (csrc/formats.hip): import / export against the numpy restatement (oracle/format_ref.py).
Integer work (UNORM, binary16, float11 / float10 packing) is bit-exact; the sRGB curve goes through pow and is held to one code / 2e-6."""
import os

import numpy as np
import pytest
import torch

import format_ref as F

FORMATS = sorted(F.TEXEL)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_format_ref_known_answers():
    """The restatement itself against hand-computed values (CPU)."""
    assert list(F.float_to_unorm(np.array([0.0, 1.0, 0.5, -3.0, 7.0, np.nan, 0.00196], np.float32), 8)) == [0, 255, 128, 0, 255, 0, 0]
    assert list(F.float_to_unorm(np.array([0.00197, 1.0 / 255.0], np.float32), 8)) == [1, 1]
    # float11: 1.0 = exponent 15, mantissa 0; 65024 = the largest finite value (0x7bf); anything that rounds past it overflows to INF (0x7c0)
    assert list(F.float_to_ufloat(np.array([1.0, 65024.0, 65280.0, 70000.0, -1.0, 0.0, np.inf, -np.inf], np.float32), 6)) == [15 << 6, 0x7BF, 0x7C0, 0x7C0, 0, 0, 0x7C0, 0]
    assert F.float_to_ufloat(np.array([np.nan], np.float32), 6)[0] == (0x7C0 | 0x20)
    # smallest float11 subnormal = 2^-20; half of it ties to even (0), just above rounds up
    assert list(F.float_to_ufloat(np.array([2.0 ** -20, 2.0 ** -21, np.nextafter(np.float32(2.0 ** -21), np.float32(1)), 3 * 2.0 ** -21], np.float32), 6)) == [1, 0, 1, 2]
    v = np.arange(0x800, dtype=np.uint32)
    f = F.ufloat_to_float(v, 6)
    finite = v < 0x7C0
    assert np.array_equal(F.float_to_ufloat(f[finite], 6), v[finite])  # every finite code round-trips
    assert np.all(np.diff(f[finite]) > 0)
    v10 = np.arange(0x400, dtype=np.uint32)
    assert np.array_equal(F.float_to_ufloat(F.ufloat_to_float(v10, 5)[v10 < 0x3E0], 5), v10[v10 < 0x3E0])
    assert abs(float(F.linear_to_srgb(np.float32(0.5))) - 0.735357) < 1e-5 and abs(float(F.srgb_to_linear(np.float32(0.5))) - 0.214041) < 1e-5


def hdr_image(h=37, w=53, seed=3):
    rng = np.random.default_rng(seed)
    img = (np.exp2(rng.uniform(-24, 17, (h, w, 4))) * rng.choice([1.0, 1.0, 1.0, -1.0], (h, w, 4))).astype(np.float32)
    img[0, :8, 0] = [0.0, -0.0, np.inf, -np.inf, np.nan, 65504.0, 65520.0, 1e-8]
    img[1, :6, 1] = [65024.0, 65280.0, 2.0 ** -20, 2.0 ** -21, 6.1e-5, 1.0]
    return img


def ldr_image(h=37, w=53, seed=4):
    rng = np.random.default_rng(seed)
    img = rng.uniform(-0.1, 1.1, (h, w, 4)).astype(np.float32)
    img[0, :6, 0] = [0.0, 1.0, 0.5, np.nan, 0.0031308, 0.00197]
    codes = (np.arange(w) % 256).astype(np.float32)
    img[2, :, 1] = (codes + 0.5) / 255.0  # exactly on the rounding boundary of UNORM8
    return img


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", FORMATS)
def test_export_then_import(mifx_lib, fmt):
    from diligentfx_amd import api

    ctx = api.PostFXContext(0)
    img = hdr_image() if "FLOAT" in fmt else ldr_image()
    t = torch.from_numpy(img).to(ctx.device)
    raw = api.image_export(ctx, t, fmt)
    want = F.encode(img, fmt)
    got = raw.cpu().numpy()
    if fmt.endswith("_SRGB"):  # pow on both sides: a code may differ by one on a few values
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1 and (d != 0).mean() < 1e-3
    else:
        assert np.array_equal(got, want), f"{fmt}: {np.argwhere(got != want)[:5]}"
    # import what was exported (with a padded pitch) and compare with the restated decode
    texel = F.TEXEL[fmt]
    pitch = ((img.shape[1] * texel + 63) // 64) * 64
    padded = torch.zeros((img.shape[0], pitch), dtype=torch.uint8, device=ctx.device)
    padded[:, : img.shape[1] * texel] = raw
    back = api.image_import(ctx, padded, img.shape[1], fmt, 4).cpu().numpy()
    ref = F.decode(got, img.shape[1], fmt)
    if fmt.endswith("_SRGB"):
        np.testing.assert_allclose(back, ref, rtol=2e-6, atol=1e-9)
    else:
        assert np.array_equal(back.view(np.uint32), ref.view(np.uint32)) or np.array_equal(np.isnan(back), np.isnan(ref)) and np.array_equal(np.nan_to_num(back), np.nan_to_num(ref))
    # fewer destination channels keep the first ones; a one-channel source exports with (0, 0, 1) in the missing channels
    one = api.image_import(ctx, padded, img.shape[1], fmt, 1).cpu().numpy()
    if fmt.endswith("_SRGB"):
        np.testing.assert_allclose(one, ref[..., 0], rtol=2e-6, atol=1e-9)
    else:
        assert np.array_equal(np.nan_to_num(one), np.nan_to_num(ref[..., 0]))
    ctx.close()


@pytest.mark.gpu
def test_gbuffer_formats_round_trip_and_errors(mifx_lib):
    """The Hydrogent G-buffer (HnBeginFrameTask.cpp:63-69) through its native formats: what the chain receives after the import is the quantised
    G-buffer; UNORM / half values that are representable survive the round trip exactly."""
    from diligentfx_amd import api, binding as B, synth

    ctx = api.PostFXContext(0)
    f = synth.make_frame(synth.Scene(), 3, 96, 64, ctx.device)
    for name, fmt, ch in (("base_color", "RGBA8_UNORM", 4), ("normal", "RGBA16_FLOAT", 4), ("motion", "RG16_FLOAT", 2), ("depth", "R32_FLOAT", 1)):
        raw = api.image_export(ctx, f[name], fmt)
        back = api.image_import(ctx, raw, 96, fmt, ch)
        again = api.image_export(ctx, back, fmt)
        assert torch.equal(raw, again), name  # idempotent: the imported plane holds representable values only
        tol = {"RGBA8_UNORM": 0.5 / 255 + 1e-7, "RGBA16_FLOAT": 1e-3, "RG16_FLOAT": 1e-3, "R32_FLOAT": 0.0}[fmt]
        src = f[name] if ch != 2 else f[name]
        assert float((back - src).abs().max()) <= tol * max(1.0, float(src.abs().max())), name
    raw = api.image_export(ctx, f["depth"], "R32_FLOAT")
    with pytest.raises(B.MifxError, match="INVALID_ARG"):
        api.image_import(ctx, raw[:, :-4].contiguous(), 96, "R32_FLOAT", 1)  # pitch smaller than a row
    with pytest.raises(KeyError):
        api.image_import(ctx, raw, 96, "BC7_UNORM", 1)
    assert ctx.lib.mifx_native_format_texel_size(999) == 0
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["RGBA8_UNORM_SRGB", "RGBA8_UNORM", "RGBA16_FLOAT", "R11G11B10_FLOAT", "RGBA16_UNORM"])
def test_tone_map_into_native_target(mifx_lib, fmt):
    """mifx_tonemap_execute_native == mifx_tonemap_execute + mifx_image_export, bit for bit, for every operator and the sRGB flag
    (the copy-frame pass writing the swap chain's format, HnPostProcessTask.cpp:974-1000)."""
    from diligentfx_amd import api, binding as B

    ctx = api.PostFXContext(0)
    rng = np.random.default_rng(11)
    hdr = np.exp2(rng.uniform(-8, 6, (45, 67, 4))).astype(np.float32)
    hdr[..., 3] = rng.uniform(0, 1, hdr.shape[:2])
    t = torch.from_numpy(hdr).to(ctx.device)
    for mode in (0, 2, 4, 5, 8, 10, 11):
        for flags in (0, 1):
            attr = B.ToneMappingAttribs.default(mode)
            two_pass = api.image_export(ctx, ctx.tone_map(t, attr, 0.3, flags), fmt)
            fused = ctx.tone_map_native(t, attr, 0.3, fmt, flags)
            assert torch.equal(fused, two_pass), (fmt, mode, flags)
    # a padded pitch leaves the padding untouched
    ts = F.TEXEL[fmt]
    padded = ctx.tone_map_native(t, B.ToneMappingAttribs.default(4), 0.3, fmt, 0, pitch_bytes=67 * ts + 12)
    assert torch.equal(padded[:, :67 * ts], api.image_export(ctx, ctx.tone_map(t, B.ToneMappingAttribs.default(4), 0.3, 0), fmt)) and int(padded[:, 67 * ts:].max()) == 0


def test_fast_r11g11b10_paths_agree_with_the_codec_for_every_float(tmp_path):
    """quantize_ufloat / encode_quantized (csrc/mifx_ufloat.h: what the native-storage build's Bloom kernels run) against the integer codec the format tests pin to
    oracle/format_ref.py -- all 2^32 bit patterns, on the host (tools/check_ufloat.cpp)."""
    import shutil
    import subprocess

    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    exe = tmp_path / "check_ufloat"
    subprocess.run(["g++", "-O2", "-fopenmp", "-std=c++17", "-I", os.path.join(ROOT, "diligentfx_amd", "csrc"), os.path.join(ROOT, "tools", "check_ufloat.cpp"), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, r.stdout[-1000:]


def test_unorm_decode_without_a_division_is_correctly_rounded():
    """mifx_device.h reads an R8_UNORM / R16_UNORM code as q = c * fl(1 / N) followed by one residual step, fma(fma(-q, N, c), fl(1 / N), q): equal to the correctly rounded
    c / N for every code (the fused multiply-adds evaluated exactly in 80-bit arithmetic here: 24 x 24-bit products fit), and so is the float32 division pyref's stores use."""
    for n in (255.0, 65535.0):
        c = np.arange(int(n) + 1, dtype=np.float32)
        r = np.float32(1.0) / np.float32(n)
        q = (c * r).astype(np.float32)
        big = np.longdouble
        t = (big(c) - big(q) * big(n)).astype(np.float32)
        out = (big(t) * big(r) + big(q)).astype(np.float32)
        want = (c.astype(np.float64) / n).astype(np.float32)
        assert np.array_equal(out, want) and np.array_equal((c / np.float32(n)).astype(np.float32), want), n
