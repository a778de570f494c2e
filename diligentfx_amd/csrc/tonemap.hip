// tonemap.hip -- M2: full-screen application of ToneMap() (+ optional LinearToSRGB), the copy-frame pass of the
// reference chain (Hydrogent/shaders/HnCopyFrame.psh:27-36,61-63).  Pure streaming: 16 B in + 16 B out per pixel
// (32 B/px).  One float4 texel per lane => 1 KiB coalesced per wave per load/store.
#include "mifx_host.h"
#include "mifx_tonemap.h"
#include "mifx_formats.h"

namespace mifx
{
template <int MODE, bool SRGB> __global__ __launch_bounds__(256) void tonemap_kernel(Img in, Img out, ToneMapK a, const float* aveLum)
{
    int x, y;
    if (!pixel_xy(out, x, y)) return;
    if (aveLum) a.aveLogLum = fmaxf(0.05f, *aveLum); // GetAverageSceneLuminance (AtmosphereShadersCommon.fxh:188-195) of the auto-exposure plane
    v4 c = ld_hdr(in, x, y, a.packedIn);
    v3 t = tone_map<MODE>(xyz(c), a);
    if (SRGB) t = linear_to_srgb(t);
    st<v4>(out, x, y, mk4(t, c.w));
}

// The copy-frame pass as the reference runs it: the render target is the swap chain's format (RGBA8_UNORM_SRGB or another TEX_FORMAT_*), so the output merger
// converts what the shader returns -- here the conversion is the tail of the kernel (mifx_formats.h), 4-8 bytes written per pixel instead of 16 and no export pass.
template <int MODE, bool SRGB> __global__ __launch_bounds__(256) void tonemap_native_kernel(Img in, NativeImg out, ToneMapK a, const float* aveLum)
{
    const int x = int(blockIdx.x * blockDim.x + threadIdx.x), y = int(blockIdx.y * blockDim.y + threadIdx.y);
    if (x >= out.w || y >= out.h) return;
    if (aveLum) a.aveLogLum = fmaxf(0.05f, *aveLum);
    const v4 c = ld_hdr(in, x, y, a.packedIn);
    v3 t = tone_map<MODE>(xyz(c), a);
    if (SRGB) t = linear_to_srgb(t);
    encode_texel(out.p + size_t(y) * out.pitch + size_t(x) * out.texel, out.fmt, mk4(t, c.w));
}

__global__ __launch_bounds__(256) void fill_f32_kernel(Img plane, int floats_per_row, float value)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x < floats_per_row) reinterpret_cast<float*>(plane.p + size_t(y) * plane.pitch)[x] = value;
}

// a clear colour, one value per channel (PostFXContext::ClearRenderTarget): `channels` floats -- or halves -- per texel
template <class T> __global__ __launch_bounds__(256) void clear_texels_kernel(Img plane, int channels, mifx_f4 color)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= plane.w * channels) return;
    const int   c = x % channels;
    const float v = c == 0 ? color.x : c == 1 ? color.y : c == 2 ? color.z : color.w;
    reinterpret_cast<T*>(plane.p + size_t(y) * plane.pitch)[x] = T(v);
}

// ------------------------------------------------------------------------------------------------ diagnostics: the fp32 helpers, element-wise
__global__ __launch_bounds__(256) void eval_math_kernel(unsigned op, const float* a, const float* b, float* out, unsigned long long n)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b ? b[i] : 0.0f;
    float r;
    switch (op)
    {
        case MIFX_MATH_FDIV: r = fdiv(x, y); break;
        case MIFX_MATH_FSQRT: r = fsqrt(x); break;
        case MIFX_MATH_SIN_BOUNDED: r = m_sin_bounded(x); break;
        case MIFX_MATH_COS_BOUNDED: r = m_cos_bounded(x); break;
        case MIFX_MATH_EXP: r = m_exp(x); break;
        default: r = m_pow(x, y); break;
    }
    out[i] = r;
}
mifx_status launch_eval_math(hipStream_t s, unsigned op, const float* a, const float* b, float* out, unsigned long long n)
{
    if (n == 0) return MIFX_OK;
    hipLaunchKernelGGL(eval_math_kernel, dim3(unsigned((n + 255) / 256), 1, 1), dim3(256, 1, 1), 0, s, op, a, b, out, n);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}

// ------------------------------------------------------------------------------------------------ diagnostics: the device's streaming rate
// One 16-byte load and one 16-byte store per lane, one texel of a float4 plane per lane like the chain's streaming passes, grid-stride over the buffer: what a
// pure copy of this library's own access pattern reaches (the "achievable" HBM rate bench.py quotes beside the 8 TB/s peak).
__global__ __launch_bounds__(256) void stream_copy_kernel(const mifx_f4* __restrict__ src, mifx_f4* __restrict__ dst, unsigned long long n)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
mifx_status launch_stream_copy(hipStream_t s, const void* src, void* dst, unsigned long long bytes)
{
    const unsigned long long n = bytes / 16u;
    if (n == 0) return MIFX_OK;
    const unsigned long long want = (n + 255u) / 256u;
    hipLaunchKernelGGL(stream_copy_kernel, dim3(unsigned(want < 65536u ? want : 65536u), 1, 1), dim3(256, 1, 1), 0, s, static_cast<const mifx_f4*>(src), static_cast<mifx_f4*>(dst), n);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}

mifx_status launch_clear_texels(hipStream_t s, Img plane, int channels, bool halves, const float color[4])
{
    const dim3 block(256, 1, 1), grid((plane.w * channels + 255) / 256, plane.h, 1);
    const mifx_f4 c{color[0], color[1], color[2], color[3]};
    if (halves) hipLaunchKernelGGL(clear_texels_kernel<_Float16>, grid, block, 0, s, plane, channels, c);
    else hipLaunchKernelGGL(clear_texels_kernel<float>, grid, block, 0, s, plane, channels, c);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}

mifx_status launch_fill_f32(hipStream_t s, Img plane, int floats_per_texel, float value)
{
    const int n = plane.w * floats_per_texel;
    dim3 block(256, 1, 1), grid((n + 255) / 256, plane.h, 1);
    hipLaunchKernelGGL(fill_f32_kernel, grid, block, 0, s, plane, n, value);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}

mifx_status launch_tonemap(hipStream_t s, Img in, Img out, const mifx_tone_mapping_attribs& attr, float ave_log_lum, uint32_t flags, const float* aveLum, bool packedIn)
{
    ToneMapK a = make_tonemapk(attr, ave_log_lum);
    a.packedIn = packedIn ? 1 : 0;
    const dim3 block(64, 4, 1);
    const dim3 grid = grid2d(out, block);
    const bool srgb = (flags & MIFX_TONEMAP_FLAG_CONVERT_OUTPUT_TO_SRGB) != 0;
#define MIFX_TM_LAUNCH(M)                                                                              \
    if (srgb) hipLaunchKernelGGL((tonemap_kernel<M, true>), grid, block, 0, s, in, out, a, aveLum);    \
    else hipLaunchKernelGGL((tonemap_kernel<M, false>), grid, block, 0, s, in, out, a, aveLum)
    MIFX_TONEMAP_DISPATCH(attr.iToneMappingMode, MIFX_TM_LAUNCH)
#undef MIFX_TM_LAUNCH
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
mifx_status launch_tonemap_native(hipStream_t s, Img in, const mifx_native_image* ldr_out, const mifx_tone_mapping_attribs& attr, float ave_log_lum, uint32_t flags, const float* aveLum,
                                  bool packedIn)
{
    NativeImg out;
    MIFX_CHECK(to_native(ldr_out, "tone map output", out));
    MIFX_REQUIRE(out.w == in.w && out.h == in.h, "tone map output: %dx%d, input %dx%d", out.w, out.h, in.w, in.h);
    ToneMapK a = make_tonemapk(attr, ave_log_lum);
    a.packedIn = packedIn ? 1 : 0;
    const dim3 block(64, 4, 1);
    const dim3 grid = grid2d(in.w, in.h, block);
    const bool srgb = (flags & MIFX_TONEMAP_FLAG_CONVERT_OUTPUT_TO_SRGB) != 0;
#define MIFX_TM_LAUNCH(M)                                                                                   \
    if (srgb) hipLaunchKernelGGL((tonemap_native_kernel<M, true>), grid, block, 0, s, in, out, a, aveLum);  \
    else hipLaunchKernelGGL((tonemap_native_kernel<M, false>), grid, block, 0, s, in, out, a, aveLum)
    MIFX_TONEMAP_DISPATCH(attr.iToneMappingMode, MIFX_TM_LAUNCH)
#undef MIFX_TM_LAUNCH
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
} // namespace mifx
