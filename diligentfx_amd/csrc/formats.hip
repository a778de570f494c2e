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
#include "mifx_formats.h"

namespace mifx
{
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
mifx_status to_native(const mifx_native_image* im, const char* what, NativeImg& out)
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
