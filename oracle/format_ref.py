"""oracle/format_ref.py -- TEST INFRASTRUCTURE ONLY: numpy restatement of the texture-format conversions of csrc/formats.hip.

There is no reference source for these (they are what the texture unit / output merger does with the reference's TEX_FORMAT_* resources); the
rules are those of the Direct3D 11.3 functional specification, section 3.2.3 "Data conversion", as recalled -- the specification is not in
this container, so for UNORM / sRGB / float11 / float10 this file is "parity unpinned".  The binary16 path is pinned to numpy's IEEE
half-precision type (round to nearest even)."""
import numpy as np

TEXEL = {"R32_FLOAT": 4, "RG32_FLOAT": 8, "RGBA32_FLOAT": 16, "R16_FLOAT": 2, "RG16_FLOAT": 4, "RGBA16_FLOAT": 8, "R8_UNORM": 1, "RG8_UNORM": 2, "RGBA8_UNORM": 4,
         "RGBA8_UNORM_SRGB": 4, "R16_UNORM": 2, "RG16_UNORM": 4, "RGBA16_UNORM": 8, "R11G11B10_FLOAT": 4}
CHANNELS = {"R32_FLOAT": 1, "RG32_FLOAT": 2, "RGBA32_FLOAT": 4, "R16_FLOAT": 1, "RG16_FLOAT": 2, "RGBA16_FLOAT": 4, "R8_UNORM": 1, "RG8_UNORM": 2, "RGBA8_UNORM": 4,
            "RGBA8_UNORM_SRGB": 4, "R16_UNORM": 1, "RG16_UNORM": 2, "RGBA16_UNORM": 4, "R11G11B10_FLOAT": 3}


def float_to_unorm(c, bits):
    c = np.asarray(c, np.float32)
    c = np.where(np.isnan(c), np.float32(0), np.clip(c, np.float32(0), np.float32(1))).astype(np.float32)
    scale = np.float32(2 ** bits - 1)
    return (c * scale + np.float32(0.5)).astype(np.float32).astype(np.uint32)  # truncation of a non-negative value


def unorm_to_float(v, bits):
    return (v.astype(np.float32) / np.float32(2 ** bits - 1)).astype(np.float32)


def srgb_to_linear(c):
    c = np.asarray(c, np.float32)
    return np.where(c <= np.float32(0.04045), c / np.float32(12.92), np.power((c + np.float32(0.055)) / np.float32(1.055), np.float32(2.4))).astype(np.float32)


def linear_to_srgb(c):
    c = np.asarray(c, np.float32)
    c = np.where(np.isnan(c), np.float32(0), np.clip(c, np.float32(0), np.float32(1))).astype(np.float32)
    return np.where(c <= np.float32(0.0031308), np.float32(12.92) * c, np.float32(1.055) * np.power(c, np.float32(1.0 / 2.4)) - np.float32(0.055)).astype(np.float32)


def float_to_ufloat(x, mbits):
    """float32 -> unsigned float with 5 exponent bits and `mbits` mantissa bits: negative -> 0, NaN stays NaN, round to nearest even, overflow -> +INF."""
    f = np.asarray(x, np.float32).view(np.uint32).astype(np.int64)
    sign, e, m = f >> 31, (f >> 23) & 0xFF, f & 0x7FFFFF
    E = e - 127 + 15
    inf = np.int64(31 << mbits)
    sub = E <= 0
    mant = np.where(sub, m | 0x800000, (np.maximum(E, 0) << 23) | m)
    shift = np.where(sub, 23 - mbits + 1 - E, 23 - mbits)
    shift = np.clip(shift, 1, 62)
    q, rem, half = mant >> shift, mant & ((np.int64(1) << shift) - 1), np.int64(1) << (shift - 1)
    r = q + ((rem > half) | ((rem == half) & ((q & 1) == 1))).astype(np.int64)
    r = np.where(sub & (E < -mbits), 0, r)
    r = np.where(E >= 31, inf, r)
    r = np.where(sign == 1, 0, r)
    r = np.where(e == 255, np.where(m != 0, inf | (1 << (mbits - 1)), np.where(sign == 1, 0, inf)), r)
    return r.astype(np.uint32)


def ufloat_to_float(v, mbits):
    v = v.astype(np.uint32)
    e, m = v >> mbits, v & ((1 << mbits) - 1)
    normal = (((e + 112) << 23) | (m << (23 - mbits))).astype(np.uint32).view(np.float32)
    sub = (m.astype(np.float32) * np.float32(1.0 / (1 << (14 + mbits)))).astype(np.float32)
    special = (np.uint32(0x7F800000) | (m << (23 - mbits))).astype(np.uint32).view(np.float32)
    return np.where(e == 31, special, np.where(e == 0, sub, normal)).astype(np.float32)


def encode(img, fmt):
    """img: float32 (H, W, 4) -> uint8 (H, W * texel) in `fmt` (channels beyond the format's are dropped)."""
    img = np.asarray(img, np.float32)
    h, w = img.shape[:2]
    n = CHANNELS[fmt]
    if fmt.endswith("32_FLOAT"):
        out = np.ascontiguousarray(img[..., :n])
    elif fmt.endswith("16_FLOAT"):
        with np.errstate(over="ignore"):
            out = np.ascontiguousarray(img[..., :n]).astype(np.float16)
    elif fmt == "R11G11B10_FLOAT":
        out = (float_to_ufloat(img[..., 0], 6) | (float_to_ufloat(img[..., 1], 6) << 11) | (float_to_ufloat(img[..., 2], 5) << 22)).astype(np.uint32)
    else:
        bits = 8 if "8_" in fmt else 16
        src = img[..., :n].copy()
        if fmt.endswith("_SRGB"):
            src[..., :3] = linear_to_srgb(src[..., :3])
        out = float_to_unorm(src, bits).astype(np.uint8 if bits == 8 else np.uint16)
    return np.ascontiguousarray(out).view(np.uint8).reshape(h, w * TEXEL[fmt])


def decode(raw, width, fmt):
    """uint8 (H, >= W * texel) -> float32 (H, W, 4); channels the format lacks are (0, 0, 0, 1)."""
    h = raw.shape[0]
    n = CHANNELS[fmt]
    body = np.ascontiguousarray(raw[:, : width * TEXEL[fmt]])
    out = np.zeros((h, width, 4), np.float32)
    out[..., 3] = 1.0
    if fmt.endswith("32_FLOAT"):
        out[..., :n] = body.view(np.float32).reshape(h, width, n)
    elif fmt.endswith("16_FLOAT"):
        out[..., :n] = body.view(np.float16).reshape(h, width, n).astype(np.float32)
    elif fmt == "R11G11B10_FLOAT":
        v = body.view(np.uint32).reshape(h, width)
        out[..., 0], out[..., 1], out[..., 2] = ufloat_to_float(v & 0x7FF, 6), ufloat_to_float((v >> 11) & 0x7FF, 6), ufloat_to_float(v >> 22, 5)
    else:
        bits = 8 if "8_" in fmt else 16
        v = body.view(np.uint8 if bits == 8 else np.uint16).reshape(h, width, n)
        out[..., :n] = unorm_to_float(v, bits)
        if fmt.endswith("_SRGB"):
            out[..., :3] = srgb_to_linear(out[..., :3])
    return out
