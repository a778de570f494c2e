// formats.hip -- the reference's native texture formats at the boundary (first step): conversion between linear-layout images in
// the formats the Hydrogent G-buffer and the post-processing targets use (HnBeginFrameTask.cpp:63-69: RGBA16_FLOAT scene colour / normal / IBL,
// RG16_FLOAT motion, RGBA8_UNORM base colour, RG8_UNORM material, R32_FLOAT depth; R11G11B10_FLOAT / R16_FLOAT / R16_UNORM effect outputs;
// RGBA8_UNORM_SRGB swap chain) and the fp32 planes of this library.  The kernels themselves keep computing on fp32 planes.
//
// Conversion rules = what a D3D11-class texture unit / output merger does (Direct3D 11.3 functional specification, section 3.2.3 "Data
// conversion", recalled -- the specification is not in this container):
//   UNORM -> float   c / (2^n - 1)
//   float -> UNORM   NaN -> 0, clamp to [0, 1], c * (2^n - 1) + 0.5, drop the fraction
//   sRGB             the exact piecewise curve on top of UNORM8 (alpha stays linear)
//   float16          IEEE binary16, round to nearest even (the export path of the hardware; the specification also permits round to zero)
//   float11 / 10     unsigned 5-bit-exponent floats: negative -> 0, NaN stays NaN, round to nearest even, overflow -> +INF
// Missing source channels read as (0, 0, 0, 1) as in D3D.
#include "mifx_host.h"

namespace mifx
{
typedef unsigned mifx_u2 __attribute__((ext_vector_type(2)));
MIFX_D unsigned float_to_unorm(float c, float scale)
{
    c = c != c ? 0.0f : fminf(fmaxf(c, 0.0f), 1.0f);
    return unsigned(c * scale + 0.5f);
}
MIFX_D float srgb_to_linear_exact(float c) { return c <= 0.04045f ? c / 12.92f : powf((c + 0.055f) / 1.055f, 2.4f); }
MIFX_D float linear_to_srgb_exact(float c)
{
    c = c != c ? 0.0f : fminf(fmaxf(c, 0.0f), 1.0f);
    return c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
}
MIFX_D float    half_to_float(unsigned short h) { return float(__builtin_bit_cast(_Float16, h)); }
MIFX_D unsigned float_to_half(float f) { return unsigned(__builtin_bit_cast(unsigned short, _Float16(f))); } // v_cvt_f16_f32: round to nearest even

// unsigned small float with 5 exponent bits and M mantissa bits (float11: M = 6, float10: M = 5), integer arithmetic only
template <int M> MIFX_D unsigned float_to_ufloat(float x)
{
    const unsigned f = __builtin_bit_cast(unsigned, x);
    const unsigned e = (f >> 23) & 0xffu, m = f & 0x7fffffu;
    if (e == 255u) return m ? ((31u << M) | (1u << (M - 1))) : ((f >> 31) ? 0u : (31u << M)); // NaN stays NaN; -INF -> 0, +INF stays
    if (f >> 31) return 0u;                                                                   // negative values clamp to 0
    const int E = int(e) - 127 + 15;
    if (E >= 31) return 31u << M; // overflow -> +INF
    unsigned mant, shift;
    if (E <= 0)
    {
        if (E < -M) return 0u;             // below half of the smallest subnormal (ties at E == -M round to even = 0 or up below)
        mant  = m | 0x800000u;             // implicit one
        shift = unsigned(23 - M + 1 - E);  // 18 .. 24 + M
    }
    else
    {
        mant  = (unsigned(E) << 23) | m;   // exponent and mantissa as one integer: a mantissa carry increments the exponent
        shift = unsigned(23 - M);
    }
    const unsigned q = mant >> shift, rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
    return q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u);
}
template <int M> MIFX_D float ufloat_to_float(unsigned v)
{
    const unsigned e = v >> M, m = v & ((1u << M) - 1u);
    if (e == 31u) return __builtin_bit_cast(float, 0x7f800000u | (m << (23 - M)));
    if (e == 0u) return float(m) * (1.0f / float(1u << (14 + M))); // subnormal: m * 2^-14 / 2^M
    return __builtin_bit_cast(float, ((e + 112u) << 23) | (m << (23 - M)));
}

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
        case MIFX_NATIVE_FORMAT_R8_UNORM: r.x = float(*g) / 255.0f; break;
        case MIFX_NATIVE_FORMAT_RG8_UNORM: { const unsigned t = *(const MIFX_GLOBAL unsigned short*)g; r.x = float(t & 0xffu) / 255.0f; r.y = float(t >> 8) / 255.0f; break; }
        case MIFX_NATIVE_FORMAT_RGBA8_UNORM:
        case MIFX_NATIVE_FORMAT_RGBA8_UNORM_SRGB:
        {
            const unsigned t = *(const MIFX_GLOBAL unsigned*)g;
            r = v4{float(t & 0xffu) / 255.0f, float((t >> 8) & 0xffu) / 255.0f, float((t >> 16) & 0xffu) / 255.0f, float(t >> 24) / 255.0f};
            if (fmt == MIFX_NATIVE_FORMAT_RGBA8_UNORM_SRGB) { r.x = srgb_to_linear_exact(r.x); r.y = srgb_to_linear_exact(r.y); r.z = srgb_to_linear_exact(r.z); }
            break;
        }
        case MIFX_NATIVE_FORMAT_R16_UNORM: r.x = float(*(const MIFX_GLOBAL unsigned short*)g) / 65535.0f; break;
        case MIFX_NATIVE_FORMAT_RG16_UNORM: { const unsigned t = *(const MIFX_GLOBAL unsigned*)g; r.x = float(t & 0xffffu) / 65535.0f; r.y = float(t >> 16) / 65535.0f; break; }
        case MIFX_NATIVE_FORMAT_RGBA16_UNORM:
        {
            const mifx_u2 t = *(const MIFX_GLOBAL mifx_u2*)g;
            r = v4{float(t.x & 0xffffu) / 65535.0f, float(t.x >> 16) / 65535.0f, float(t.y & 0xffffu) / 65535.0f, float(t.y >> 16) / 65535.0f};
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
__global__ __launch_bounds__(256) void image_import_kernel(NativeImg src, Img dst, int dstChannels)
{
    int x, y;
    if (!pixel_xy(dst, x, y)) return;
    const v4 c = decode_texel(src.p + size_t(y) * src.pitch + size_t(x) * src.texel, src.fmt);
    if (dstChannels == 1) st<float>(dst, x, y, c.x);
    else if (dstChannels == 2) st<v2>(dst, x, y, v2{c.x, c.y});
    else st<v4>(dst, x, y, c);
}
__global__ __launch_bounds__(256) void image_export_kernel(Img src, int srcChannels, NativeImg dst)
{
    int x, y;
    if (!pixel_xy(src, x, y)) return;
    v4 c{0.0f, 0.0f, 0.0f, 1.0f};
    if (srcChannels == 1) c.x = ld<float>(src, x, y);
    else if (srcChannels == 2) { const v2 t = ld<v2>(src, x, y); c.x = t.x; c.y = t.y; }
    else c = ld<v4>(src, x, y);
    encode_texel(dst.p + size_t(y) * dst.pitch + size_t(x) * dst.texel, dst.fmt, c);
}

uint32_t native_texel_size(uint32_t fmt)
{
    switch (fmt)
    {
        case MIFX_NATIVE_FORMAT_R8_UNORM: return 1;
        case MIFX_NATIVE_FORMAT_R16_FLOAT: case MIFX_NATIVE_FORMAT_RG8_UNORM: case MIFX_NATIVE_FORMAT_R16_UNORM: return 2;
        case MIFX_NATIVE_FORMAT_R32_FLOAT: case MIFX_NATIVE_FORMAT_RG16_FLOAT: case MIFX_NATIVE_FORMAT_RGBA8_UNORM: case MIFX_NATIVE_FORMAT_RGBA8_UNORM_SRGB:
        case MIFX_NATIVE_FORMAT_RG16_UNORM: case MIFX_NATIVE_FORMAT_R11G11B10_FLOAT: return 4;
        case MIFX_NATIVE_FORMAT_RG32_FLOAT: case MIFX_NATIVE_FORMAT_RGBA16_FLOAT: case MIFX_NATIVE_FORMAT_RGBA16_UNORM: return 8;
        case MIFX_NATIVE_FORMAT_RGBA32_FLOAT: return 16;
        default: return 0;
    }
}
static mifx_status to_native(const mifx_native_image* im, const char* what, NativeImg& out)
{
    MIFX_REQUIRE(im != nullptr && im->data != nullptr && im->width > 0 && im->height > 0, "%s: null or empty native image", what);
    const uint32_t ts = native_texel_size(im->format);
    MIFX_REQUIRE(ts != 0, "%s: unknown native format %u", what, im->format);
    MIFX_REQUIRE(im->pitch_bytes >= im->width * ts && im->pitch_bytes % (ts < 4 ? ts : 4) == 0 && (reinterpret_cast<uintptr_t>(im->data) % (ts < 16 ? ts : 16)) == 0,
                 "%s: pitch %u / alignment does not fit %u texels of %u bytes", what, im->pitch_bytes, im->width, ts);
    out = NativeImg{static_cast<unsigned char*>(im->data), int(im->width), int(im->height), int(im->pitch_bytes), im->format, ts};
    return MIFX_OK;
}
static int channels_of(uint32_t fmt) { return fmt == MIFX_FORMAT_F32 ? 1 : (fmt == MIFX_FORMAT_F32X2 ? 2 : 4); }

mifx_status launch_image_import(hipStream_t s, const mifx_native_image* src, const mifx_image2d* dst)
{
    NativeImg n;
    MIFX_CHECK(to_native(src, "mifx_image_import", n));
    MIFX_REQUIRE(dst != nullptr, "mifx_image_import: null destination");
    Img d;
    MIFX_CHECK(to_img_wh(dst, dst->format, src->width, src->height, "dst", d));
    const dim3 block(64, 4, 1);
    hipLaunchKernelGGL(image_import_kernel, grid2d(d, block), block, 0, s, n, d, channels_of(dst->format));
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
mifx_status launch_image_export(hipStream_t s, const mifx_image2d* src, const mifx_native_image* dst)
{
    NativeImg n;
    MIFX_CHECK(to_native(dst, "mifx_image_export", n));
    MIFX_REQUIRE(src != nullptr, "mifx_image_export: null source");
    Img i;
    MIFX_CHECK(to_img_wh(src, src->format, dst->width, dst->height, "src", i));
    const dim3 block(64, 4, 1);
    hipLaunchKernelGGL(image_export_kernel, grid2d(i, block), block, 0, s, i, channels_of(src->format), n);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
} // namespace mifx
