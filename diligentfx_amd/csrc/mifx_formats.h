// mifx_formats.h -- device-side codec of the reference's native texture formats (see formats.hip for the rules and their source): decode_texel / encode_texel and the
// NativeImg view, shared by the import / export kernels (formats.hip) and by passes that write a native format directly (tonemap.hip).
#pragma once
#include "mifx.h"
#include "mifx_device.h"

namespace mifx
{
typedef unsigned mifx_u2 __attribute__((ext_vector_type(2)));
MIFX_D unsigned float_to_unorm(float c, float scale)
{
    c = c != c ? 0.0f : fminf(fmaxf(c, 0.0f), 1.0f);
    return unsigned(c * scale + 0.5f);
}
// code / N (N = 255, 65535) correctly rounded without a division: q = code * fl(1 / N) and one residual step -- equal to the IEEE quotient for every code (checked
// exhaustively in numpy: all 256 / 65536 codes); 4 instructions instead of the 8 of fdiv
template <unsigned N> MIFX_D float unorm_to_float(unsigned code)
{
    const float c = float(code), r = 1.0f / float(N), q = c * r;
    return __builtin_fmaf(__builtin_fmaf(-q, float(N), c), r, q);
}
MIFX_D float srgb_to_linear_exact(float c) { return c <= 0.04045f ? c / 12.92f : powf((c + 0.055f) / 1.055f, 2.4f); }
MIFX_D float linear_to_srgb_exact(float c)
{
    c = c != c ? 0.0f : fminf(fmaxf(c, 0.0f), 1.0f);
    return c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
}
MIFX_D float    half_to_float(unsigned short h) { return float(__builtin_bit_cast(_Float16, h)); }
MIFX_D unsigned float_to_half(float f) { return unsigned(__builtin_bit_cast(unsigned short, _Float16(f))); } // v_cvt_f16_f32: round to nearest even

// (float_to_ufloat / ufloat_to_float: mifx_device.h, shared with the storage types of the native-storage build)
MIFX_D v4 decode_texel(const unsigned char* p, unsigned fmt)
{
    const MIFX_GLOBAL unsigned char* g = (const MIFX_GLOBAL unsigned char*)p;
    v4 r{0.0f, 0.0f, 0.0f, 1.0f};
    switch (fmt)
    {
        case MIFX_NATIVE_FORMAT_R32_FLOAT: r.x = *(const MIFX_GLOBAL float*)g; break;
        case MIFX_NATIVE_FORMAT_RG32_FLOAT: { const mifx_f2 t = *(const MIFX_GLOBAL mifx_f2*)g; r.x = t.x; r.y = t.y; break; }
        case MIFX_NATIVE_FORMAT_RGBA32_FLOAT: { const mifx_f4 t = *(const MIFX_GLOBAL mifx_f4*)g; r = v4{t.x, t.y, t.z, t.w}; break; }
        case MIFX_NATIVE_FORMAT_R16_FLOAT: r.x = half_to_float(*(const MIFX_GLOBAL unsigned short*)g); break;
        case MIFX_NATIVE_FORMAT_RG16_FLOAT: { const unsigned t = *(const MIFX_GLOBAL unsigned*)g; r.x = half_to_float(t & 0xffffu); r.y = half_to_float(t >> 16); break; }
        case MIFX_NATIVE_FORMAT_RGBA16_FLOAT:
        {
            const mifx_u2 t = *(const MIFX_GLOBAL mifx_u2*)g;
            r = v4{half_to_float(t.x & 0xffffu), half_to_float(t.x >> 16), half_to_float(t.y & 0xffffu), half_to_float(t.y >> 16)};
            break;
        }
        case MIFX_NATIVE_FORMAT_R8_UNORM: r.x = unorm_to_float<255>(*g); break;
        case MIFX_NATIVE_FORMAT_RG8_UNORM: { const unsigned t = *(const MIFX_GLOBAL unsigned short*)g; r.x = unorm_to_float<255>(t & 0xffu); r.y = unorm_to_float<255>(t >> 8); break; }
        case MIFX_NATIVE_FORMAT_RGBA8_UNORM:
        case MIFX_NATIVE_FORMAT_RGBA8_UNORM_SRGB:
        {
            const unsigned t = *(const MIFX_GLOBAL unsigned*)g;
            r = v4{unorm_to_float<255>(t & 0xffu), unorm_to_float<255>((t >> 8) & 0xffu), unorm_to_float<255>((t >> 16) & 0xffu), unorm_to_float<255>(t >> 24)};
            if (fmt == MIFX_NATIVE_FORMAT_RGBA8_UNORM_SRGB) { r.x = srgb_to_linear_exact(r.x); r.y = srgb_to_linear_exact(r.y); r.z = srgb_to_linear_exact(r.z); }
            break;
        }
        case MIFX_NATIVE_FORMAT_R16_UNORM: r.x = unorm_to_float<65535>(*(const MIFX_GLOBAL unsigned short*)g); break;
        case MIFX_NATIVE_FORMAT_RG16_UNORM: { const unsigned t = *(const MIFX_GLOBAL unsigned*)g; r.x = unorm_to_float<65535>(t & 0xffffu); r.y = unorm_to_float<65535>(t >> 16); break; }
        case MIFX_NATIVE_FORMAT_RGBA16_UNORM:
        {
            const mifx_u2 t = *(const MIFX_GLOBAL mifx_u2*)g;
            r = v4{unorm_to_float<65535>(t.x & 0xffffu), unorm_to_float<65535>(t.x >> 16), unorm_to_float<65535>(t.y & 0xffffu), unorm_to_float<65535>(t.y >> 16)};
            break;
        }
        case MIFX_NATIVE_FORMAT_R11G11B10_FLOAT:
        {
            const unsigned t = *(const MIFX_GLOBAL unsigned*)g;
            r.x = ufloat_to_float<6>(t & 0x7ffu); r.y = ufloat_to_float<6>((t >> 11) & 0x7ffu); r.z = ufloat_to_float<5>(t >> 22);
            break;
        }
        default: break;
    }
    return r;
}
MIFX_D void encode_texel(unsigned char* p, unsigned fmt, v4 c)
{
    MIFX_GLOBAL unsigned char* g = (MIFX_GLOBAL unsigned char*)p;
    switch (fmt)
    {
        case MIFX_NATIVE_FORMAT_R32_FLOAT: *(MIFX_GLOBAL float*)g = c.x; break;
        case MIFX_NATIVE_FORMAT_RG32_FLOAT: *(MIFX_GLOBAL mifx_f2*)g = mifx_f2{c.x, c.y}; break;
        case MIFX_NATIVE_FORMAT_RGBA32_FLOAT: *(MIFX_GLOBAL mifx_f4*)g = mifx_f4{c.x, c.y, c.z, c.w}; break;
        case MIFX_NATIVE_FORMAT_R16_FLOAT: *(MIFX_GLOBAL unsigned short*)g = (unsigned short)float_to_half(c.x); break;
        case MIFX_NATIVE_FORMAT_RG16_FLOAT: *(MIFX_GLOBAL unsigned*)g = float_to_half(c.x) | (float_to_half(c.y) << 16); break;
        case MIFX_NATIVE_FORMAT_RGBA16_FLOAT: *(MIFX_GLOBAL mifx_u2*)g = mifx_u2{float_to_half(c.x) | (float_to_half(c.y) << 16), float_to_half(c.z) | (float_to_half(c.w) << 16)}; break;
        case MIFX_NATIVE_FORMAT_R8_UNORM: *g = (unsigned char)float_to_unorm(c.x, 255.0f); break;
        case MIFX_NATIVE_FORMAT_RG8_UNORM: *(MIFX_GLOBAL unsigned short*)g = (unsigned short)(float_to_unorm(c.x, 255.0f) | (float_to_unorm(c.y, 255.0f) << 8)); break;
        case MIFX_NATIVE_FORMAT_RGBA8_UNORM_SRGB: c.x = linear_to_srgb_exact(c.x); c.y = linear_to_srgb_exact(c.y); c.z = linear_to_srgb_exact(c.z); [[fallthrough]];
        case MIFX_NATIVE_FORMAT_RGBA8_UNORM:
            *(MIFX_GLOBAL unsigned*)g = float_to_unorm(c.x, 255.0f) | (float_to_unorm(c.y, 255.0f) << 8) | (float_to_unorm(c.z, 255.0f) << 16) | (float_to_unorm(c.w, 255.0f) << 24);
            break;
        case MIFX_NATIVE_FORMAT_R16_UNORM: *(MIFX_GLOBAL unsigned short*)g = (unsigned short)float_to_unorm(c.x, 65535.0f); break;
        case MIFX_NATIVE_FORMAT_RG16_UNORM: *(MIFX_GLOBAL unsigned*)g = float_to_unorm(c.x, 65535.0f) | (float_to_unorm(c.y, 65535.0f) << 16); break;
        case MIFX_NATIVE_FORMAT_RGBA16_UNORM:
            *(MIFX_GLOBAL mifx_u2*)g = mifx_u2{float_to_unorm(c.x, 65535.0f) | (float_to_unorm(c.y, 65535.0f) << 16), float_to_unorm(c.z, 65535.0f) | (float_to_unorm(c.w, 65535.0f) << 16)};
            break;
        case MIFX_NATIVE_FORMAT_R11G11B10_FLOAT: *(MIFX_GLOBAL unsigned*)g = float_to_ufloat<6>(c.x) | (float_to_ufloat<6>(c.y) << 11) | (float_to_ufloat<5>(c.z) << 22); break;
        default: break;
    }
}

struct NativeImg
{
    unsigned char* p;
    int            w, h, pitch;
    unsigned       fmt, texel;
};
mifx_status to_native(const mifx_native_image* im, const char* what, NativeImg& out); // validates a borrowed native image (formats.hip)
} // namespace mifx
