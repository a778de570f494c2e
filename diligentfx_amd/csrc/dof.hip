// dof.hip -- depth of field: the eleven reference passes of PostProcess/DepthOfField as nine kernels.
//   dof_coc_kernel            D1  DOF_ComputeCircleOfConfusion.fx:24-39
//   dof_temporal_coc_kernel   D2  DOF_ComputeTemporalCircleOfConfusion.fx:48-92
//   dof_dilation_*_kernel     D3 + D4: DOF_ComputeSeparatedCircleOfConfusion.fx:5-11 is folded into the load of the first dilation level
//                             (DOF_ComputeDilationCircleOfConfusion.fx:8-52); with dimensions divisible by 8 the three levels are one launch
//   dof_blur_kernel           D5 horizontal + vertical (DOF_ComputeBlurredCircleOfConfusion.fx:8-28) through an LDS tile
//   dof_prefilter_kernel      D6  DOF_ComputePrefilteredTexture.fx:23-52
//   dof_bokeh_gather_kernel   D7  DOF_ComputeBokehFirstPass.fx:49-104 (template: DOF_OPTION_KARIS_INVERSE)
//   dof_bokeh_fill_kernel     D8  DOF_ComputeBokehSecondPass.fx:40-85
//   dof_postfilter_kernel     D9  DOF_ComputePostfilteredTexture.fx:26-48
//   dof_combine_kernel        D10 DOF_ComputeCombinedTexture.fx:36-46
// All planes fp32 (the reference: R16F / R16_UNORM CoC, RGBA16F half-resolution colour, R11G11B10F result -- native formats are SURVEY 8f N4).
//
// The two bokeh passes compare the interpolated alpha (far CoC) of every tap with the interpolated alpha of the centre ("a >= CoCFar"); over
// regions of constant CoC (sky, clamped CoC) both sides are the same number up to the rounding of the bilinear weights, so the comparison is
// decided by that rounding.  The alpha channel is therefore interpolated with the reference's separate multiplies and adds in the reference's
// order (sample_rgb_alpha_strict); only the colour channels use fused multiply-adds.
#include "mifx_host.h"
#include "mifx_pyramid.h"

namespace mifx
{
MIFX_D v2 dof_pixel_uv(int x, int y, int w, int h) // NormalizedDeviceXYToTexUV(f2NormalizedXY) of the texel centre
{
    const v2 ndc{2.0f * fdiv(float(x) + 0.5f, float(w)) - 1.0f, 1.0f - 2.0f * fdiv(float(y) + 0.5f, float(h))};
    return ndc_to_uv(ndc);
}
MIFX_D float sdr_weight(v3 c) { return fdiv(1.0f, 1.0f + luminance601(c)); } // ComputeSDRWeight, DOF_Common.fx:14-17
MIFX_D float hdr_weight(v3 c) { return 1.0f + luminance601(c); }              // ComputeHDRWeight, DOF_Common.fx:9-12

// one bilinear tap of a float4 plane: rgb with fused multiply-adds, alpha in the reference's operation order (see the header)
struct Tap4 { v3 rgb; float a; };
MIFX_D Tap4 sample_rgb_alpha_strict(const Img& im, float u, float v)
{
    const BilinearTaps b = bilinear_taps<kV4Bytes>(im, u, v);
    const v4 t00 = ld_at<v4>(im, b.o00), t10 = ld_at<v4>(im, b.o10), t01 = ld_at<v4>(im, b.o01), t11 = ld_at<v4>(im, b.o11);
    Tap4 r;
    r.a = t00.w * b.w00 + t10.w * b.w10 + t01.w * b.w01 + t11.w * b.w11;
    {
        MIFX_FMA_BLOCK
        r.rgb = v3{t00.x * b.w00 + t10.x * b.w10 + t01.x * b.w01 + t11.x * b.w11, t00.y * b.w00 + t10.y * b.w10 + t01.y * b.w01 + t11.y * b.w11,
                   t00.z * b.w00 + t10.z * b.w10 + t01.z * b.w01 + t11.z * b.w11};
    }
    return r;
}

// the colour of one bilinear tap only (the near-field loops do not read the alpha of their taps): 12-byte loads, a quarter less L1 traffic in the
// L1-bandwidth-bound gather
typedef float mifx_f3 __attribute__((ext_vector_type(3)));
MIFX_D v3 sample_rgb(const Img& im, float u, float v)
{
    const BilinearTaps b = bilinear_taps<kV4Bytes>(im, u, v);
#ifdef MIFX_STORAGE_H4 // (binary16 texels are 8 bytes: one load of the whole texel)
    const v4 t00 = ld_at<v4>(im, b.o00), t10 = ld_at<v4>(im, b.o10), t01 = ld_at<v4>(im, b.o01), t11 = ld_at<v4>(im, b.o11);
#else
    const mifx_f3 t00 = *(const MIFX_GLOBAL mifx_f3*)(im.p + b.o00), t10 = *(const MIFX_GLOBAL mifx_f3*)(im.p + b.o10), t01 = *(const MIFX_GLOBAL mifx_f3*)(im.p + b.o01),
                  t11 = *(const MIFX_GLOBAL mifx_f3*)(im.p + b.o11);
#endif
    {
        MIFX_FMA_BLOCK
        return v3{t00.x * b.w00 + t10.x * b.w10 + t01.x * b.w01 + t11.x * b.w11, t00.y * b.w00 + t10.y * b.w10 + t01.y * b.w01 + t11.y * b.w11,
                  t00.z * b.w00 + t10.z * b.w10 + t01.z * b.w01 + t11.z * b.w11};
    }
}

// ------------------------------------------------------------------------------------------------ D1
struct DofCocK
{
    float p10, p11, p14, p15; // mProj elements of DepthToCameraZ (ShaderUtilities.fxh:29-40)
    float K;                  // f * f / (N * (F - f)), f = focal length in metres -- uniform, evaluated by the host in the reference's order
    float focus, denom;       // fFocusDistance; fSensorWidth * MaxCircleOfConfusion
};
__global__ __launch_bounds__(256) void dof_coc_kernel(Img depth, Img out, DofCocK k)
{
    int x, y;
    if (!pixel_xy(out, x, y)) return;
    const float d   = ld<float>(depth, x, y);
    const float z   = fdiv(k.p14 - d * k.p15, d * k.p11 - k.p10);
    const float coc = fdiv(k.K * (z - k.focus), fmaxf(z, 1e-4f));
    st<coc_t>(out, x, y, clampf(fdiv(1000.0f * coc, k.denom), -1.0f, 1.0f));
}

// ------------------------------------------------------------------------------------------------ D2
__global__ __launch_bounds__(256) void dof_temporal_coc_kernel(Img curr, Img prev, Img motionTex, Img out, float vw, float vh, float ivw, float ivh, float stability)
{
    int x, y;
    if (!pixel_xy(out, x, y)) return;
    const v2    m       = ld<cm_t>(motionTex, x, y);
    const float px      = float(x) + 0.5f, py = float(y) + 0.5f;
    const float prevX   = px - (m.x * 0.5f) * vw, prevY = py - (m.y * -0.5f) * vh; // F3NDC_XYZ_TO_UVD_SCALE.xy
    const float cocCurr = ld<coc_t>(curr, x, y);
    if (!(prevX >= 0.0f && prevY >= 0.0f && prevX < vw && prevY < vh)) // IsInsideScreen, PostFX_Common.fxh:121-127
    {
        st<coc_t>(out, x, y, cocCurr);
        return;
    }
    const float cocPrev = sample_linear_clamp_f_taps<coc_t>(prev, prevX * ivw, prevY * ivh);
    float m1 = 0.0f, m2 = 0.0f; // ComputePixelStatistic :48-72; the point-clamp sampler at texel centres is a clamped load
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
        {
            const float c = ld_clamp<coc_t>(curr, x + dx, y + dy);
            m1 += c;
            m2 += c * c;
        }
    const float mean = fdiv(m1, 9.0f), variance = fdiv(m2, 9.0f) - mean * mean, stdDev = fsqrt(fmaxf(variance, 0.0f));
    const float lo = mean - 2.5f * stdDev, hi = mean + 2.5f * stdDev; // DOF_TEMPORAL_VARIANCE_GAMMA
    st<coc_t>(out, x, y, lerpf(cocCurr, clampf(cocPrev, lo, hi), stability));
}

// ------------------------------------------------------------------------------------------------ D3 + D4
// abs(CoC) * float(CoC < 0.0), as D3's target keeps it (the separated circle of confusion is a target of its own in the reference, DepthOfField.cpp:227-240: R16_UNORM in the
// native-storage build; the levels below are maxima of such values and the stores of dil_t keep them exactly)
MIFX_D float near_coc(float c) { return quantize_as<dil_t>(c < 0.0f ? -c : 0.0f); }
struct DilationOp
{
    using T = float;
    Img src;    // the signed CoC (full resolution): the first level reads it through near_coc()
    Img dst[3]; // dilation levels 1..3
    MIFX_D float load(int x, int y) const { return near_coc(ld<coc_t>(src, x, y)); }
    MIFX_D void  quad(int x, int y, float& a, float& b, float& c, float& d) const { a = load(2 * x, 2 * y); b = load(2 * x, 2 * y + 1); c = load(2 * x + 1, 2 * y); d = load(2 * x + 1, 2 * y + 1); }
    MIFX_D float reduce(float a, float b, float c, float d) const { return fmaxf(fmaxf(a, b), fmaxf(c, d)); }
    MIFX_D float stored(float v) const { return v; }
    MIFX_D bool  inside(int l, int x, int y) const { return x < dst[l - 1].w && y < dst[l - 1].h; }
    MIFX_D int   first_row() const { return 0; }
    MIFX_D void  store(int l, int x, int y, float v) const { st<dil_t>(dst[l - 1], x, y, v); }
};
__global__ __launch_bounds__(256) void dof_dilation_levels_kernel(DilationOp op, int nl) { pyramid_reduce_levels(op, nl); }

// one level, any source size (an odd dimension widens the footprint to three texels, clamped)
template <bool FROM_COC> __global__ __launch_bounds__(256) void dof_dilation_level_kernel(Img last, Img out)
{
    int x, y;
    if (!pixel_xy(out, x, y)) return;
    auto S = [&](int ox, int oy) {
        return FROM_COC ? near_coc(ld_clamp<coc_t>(last, 2 * x + ox, 2 * y + oy)) : ld_clamp<dil_t>(last, 2 * x + ox, 2 * y + oy);
    };
    float m = fmaxf(fmaxf(S(0, 0), S(0, 1)), fmaxf(S(1, 0), S(1, 1)));
    const bool oddW = (last.w & 1) != 0, oddH = (last.h & 1) != 0;
    if (oddW) m = fmaxf(m, fmaxf(S(2, 0), S(2, 1)));
    if (oddH) m = fmaxf(m, fmaxf(S(0, 2), S(1, 2)));
    if (oddW && oddH) m = fmaxf(m, S(2, 2));
    st<dil_t>(out, x, y, m);
}

// ------------------------------------------------------------------------------------------------ D5 (both directions)
constexpr int kGaussRadius = 6; // DOF_GAUSS_KERNEL_RADIUS
struct GaussK { float w[2 * kGaussRadius + 1]; };
constexpr int kBlurBX = 32, kBlurBY = 8, kBlurTW = kBlurBX + 2 * kGaussRadius, kBlurTH = kBlurBY + 2 * kGaussRadius;
__global__ __launch_bounds__(256) void dof_blur_kernel(Img in, Img out, GaussK g)
{
    __shared__ float src[kBlurTH][kBlurTW];  // source texels, clamp addressing (ClampScreenCoord)
    __shared__ float rowBlur[kBlurTH][kBlurBX]; // the horizontally blurred rows the block's columns need
    const int tid = int(threadIdx.y) * kBlurBX + int(threadIdx.x);
    const int bx0 = int(blockIdx.x) * kBlurBX, by0 = int(blockIdx.y) * kBlurBY;
    for (int i = tid; i < kBlurTW * kBlurTH; i += kBlurBX * kBlurBY)
    {
        const int c = i % kBlurTW, r = i / kBlurTW;
        src[r][c] = ld_clamp<dil_t>(in, bx0 - kGaussRadius + c, by0 - kGaussRadius + r);
    }
    __syncthreads();
    for (int i = tid; i < kBlurBX * kBlurTH; i += kBlurBX * kBlurBY)
    {
        const int c = i % kBlurBX, r = i / kBlurBX;
        float sum = 0.0f;
#pragma unroll
        for (int s = 0; s <= 2 * kGaussRadius; ++s) sum += src[r][c + s] * g.w[s];
        rowBlur[r][c] = quantize_as<dil_t>(sum); // (the reference stores the horizontal pass in a target of its own, DepthOfField.cpp:243-253)
    }
    __syncthreads();
    const int x = bx0 + int(threadIdx.x), y = by0 + int(threadIdx.y);
    if (x >= out.w || y >= out.h) return;
    float sum = 0.0f;
#pragma unroll
    for (int s = 0; s <= 2 * kGaussRadius; ++s) sum += rowBlur[int(threadIdx.y) + s][threadIdx.x] * g.w[s];
    st<dil_t>(out, x, y, sum);
}

// ------------------------------------------------------------------------------------------------ D6
__global__ __launch_bounds__(256) void dof_prefilter_kernel(Img color, Img coc, Img dilation, Img outNear, Img outFar)
{
    int x, y;
    if (!pixel_xy(outNear, x, y)) return;
    const v2 uv = dof_pixel_uv(x, y, outNear.w, outNear.h);
    float cocMax = -3.402823466e+38f;
    v4    sum    = mk4(0.0f);
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        const int   lx = 2 * x + (i & 1), ly = 2 * y + (i >> 1); // always inside: the targets are (W / 2) x (H / 2)
        const v3    c  = xyz(ld<v4>(color, lx, ly));
        const float w  = sdr_weight(c);
        cocMax = fmaxf(cocMax, ld<coc_t>(coc, lx, ly));
        sum += mk4(c, 1.0f) * w;
    }
    const float fgAlpha = sample_linear_clamp_f_taps<dil_t>(dilation, uv.x, uv.y);
    const float bgAlpha = cocMax > 0.0f ? cocMax : 0.0f; // abs(CoCMax) * float(CoCMax > 0.0)
    const v3    rgb     = xyz(sum) / fmaxf(sum.w, 1.e-5f);
    st<v4>(outNear, x, y, mk4(rgb, fgAlpha));
    st<v4>(outFar, x, y, mk4(rgb, bgAlpha));
}

// ------------------------------------------------------------------------------------------------ D7 / D8
struct BokehK
{
    const float* kernel; // sampleCount x float2, device memory (uniform index: scalar loads)
    int          sampleCount;
    float        maxCoC, aspect;
};
template <bool KARIS> __global__ __launch_bounds__(256) void dof_bokeh_gather_kernel(Img nearTex, Img farTex, Img radiance, Img outNear, Img outFar, BokehK k)
{
    int x, y;
    if (!tiled_xy(outNear, x, y)) return;
    const v2    uv      = dof_pixel_uv(x, y, outNear.w, outNear.h);
    const float cocNear = sample_rgb_alpha_strict(nearTex, uv.x, uv.y).a;
    const float cocFar  = sample_rgb_alpha_strict(farTex, uv.x, uv.y).a;
    v4 fg = mk4(0.0f), bg = mk4(0.0f);
    if (cocNear > 0.0f)
        for (int i = 0; i < k.sampleCount; ++i)
        {
            const float spx = ((0.5f * k.kernel[2 * i]) * cocNear) * k.maxCoC, spy = ((0.5f * k.kernel[2 * i + 1]) * cocNear) * k.maxCoC;
            const float su = uv.x + spx, sv = uv.y + k.aspect * spy;
            const v3    t  = sample_rgb(nearTex, su, sv);
            const float w  = KARIS ? hdr_weight(sample_rgb(radiance, su, sv)) : 1.0f;
            {
                MIFX_FMA_BLOCK
                fg = v4{fg.x + t.x * w, fg.y + t.y * w, fg.z + t.z * w, fg.w + w};
            }
        }
    if (cocFar > 0.0f)
        for (int i = 0; i < k.sampleCount; ++i)
        {
            const float spx = ((0.5f * k.kernel[2 * i]) * cocFar) * k.maxCoC, spy = ((0.5f * k.kernel[2 * i + 1]) * cocFar) * k.maxCoC;
            const float su = uv.x + spx, sv = uv.y + k.aspect * spy;
            const Tap4  t  = sample_rgb_alpha_strict(farTex, su, sv);
            const float w  = (KARIS ? hdr_weight(sample_rgb(radiance, su, sv)) : 1.0f) * (t.a >= cocFar ? 1.0f : 0.0f);
            {
                MIFX_FMA_BLOCK
                bg = v4{bg.x + t.rgb.x * w, bg.y + t.rgb.y * w, bg.z + t.rgb.z * w, bg.w + w};
            }
        }
    st<v4>(outNear, x, y, mk4(xyz(fg) * rcpf(fg.w + (fg.w == 0.0f ? 1.0f : 0.0f)), cocNear));
    st<v4>(outFar, x, y, mk4(xyz(bg) * rcpf(bg.w + (bg.w == 0.0f ? 1.0f : 0.0f)), cocFar));
}

__global__ __launch_bounds__(256) void dof_bokeh_fill_kernel(Img nearTex, Img farTex, Img outNear, Img outFar, BokehK k)
{
    int x, y;
    if (!tiled_xy(outNear, x, y)) return;
    const v2   uv = dof_pixel_uv(x, y, outNear.w, outNear.h);
    const Tap4 cn = sample_rgb_alpha_strict(nearTex, uv.x, uv.y), cf = sample_rgb_alpha_strict(farTex, uv.x, uv.y);
    const float cocNear = cn.a, cocFar = cf.a;
    v3 fg = cn.rgb, bg = cf.rgb;
    if (cocNear > 0.0f)
        for (int i = 0; i < k.sampleCount; ++i)
        {
            const float spx = ((0.25f * k.kernel[2 * i]) * cocNear) * k.maxCoC, spy = ((0.25f * k.kernel[2 * i + 1]) * cocNear) * k.maxCoC;
            fg = max3(sample_rgb(nearTex, uv.x + spx, uv.y + k.aspect * spy), fg);
        }
    if (cocFar > 0.0f)
        for (int i = 0; i < k.sampleCount; ++i)
        {
            const float spx = ((0.25f * k.kernel[2 * i]) * cocFar) * k.maxCoC, spy = ((0.25f * k.kernel[2 * i + 1]) * cocFar) * k.maxCoC;
            const Tap4  t   = sample_rgb_alpha_strict(farTex, uv.x + spx, uv.y + k.aspect * spy);
            bg = max3(t.rgb * (t.a >= cocFar ? 1.0f : 0.0f), bg);
        }
    st<v4>(outNear, x, y, mk4(fg, cocNear));
    st<v4>(outFar, x, y, mk4(bg, cocFar));
}

// ------------------------------------------------------------------------------------------------ D9
__global__ __launch_bounds__(256) void dof_postfilter_kernel(Img nearTex, Img farTex, Img outNear, Img outFar)
{
    int x, y;
    if (!pixel_xy(outNear, x, y)) return;
    const v2 uv = dof_pixel_uv(x, y, outNear.w, outNear.h);
    const v2 ts{fdiv(1.0f, float(nearTex.w)), fdiv(1.0f, float(nearTex.h))}; // rcp(g_TextureColorCoCNear dimensions), for both textures
    auto tent = [&](const Img& im) {
        const v4 A = sample_linear_clamp_v4_taps(im, uv.x + ts.x * -0.5f, uv.y + ts.y * -0.5f), B = sample_linear_clamp_v4_taps(im, uv.x + ts.x * -0.5f, uv.y + ts.y * 0.5f),
                 C = sample_linear_clamp_v4_taps(im, uv.x + ts.x * 0.5f, uv.y + ts.y * -0.5f), D = sample_linear_clamp_v4_taps(im, uv.x + ts.x * 0.5f, uv.y + ts.y * 0.5f);
        return 0.25f * (A + B + C + D);
    };
    st<v4>(outNear, x, y, tent(nearTex));
    st<v4>(outFar, x, y, tent(farTex));
}

// ------------------------------------------------------------------------------------------------ D10
MIFX_D float smoothstep01(float a, float b, float x)
{
    const float t = saturate(fdiv(x - a, b - a));
    return t * t * (3.0f - 2.0f * t);
}
__global__ __launch_bounds__(256) void dof_combine_kernel(Img color, Img nearTex, Img farTex, Img out, float alpha)
{
    int x, y;
    if (!pixel_xy(out, x, y)) return;
    const v2 uv  = dof_pixel_uv(x, y, out.w, out.h);
    const v4 src = ld<v4>(color, x, y);
    const v4 n = sample_linear_clamp_v4_taps(nearTex, uv.x, uv.y), f = sample_linear_clamp_v4_taps(farTex, uv.x, uv.y);
    v3 r = xyz(src);
    r = lerp3(r, xyz(f), smoothstep01(0.1f, 1.0f, f.w));
    r = lerp3(r, xyz(n), smoothstep01(0.1f, 1.0f, n.w));
#ifdef MIFX_STORAGE_H4
    st<bloom_t>(out, x, y, mk4(lerp3(xyz(src), r, alpha), 1.0f)); // an R11G11B10_FLOAT target (DepthOfField.cpp:281-289): a 4-byte plane like Bloom's levels, alpha reads as 1
#else
    st<v4>(out, x, y, mk4(lerp3(xyz(src), r, alpha), src.w)); // alpha: carried through (the reference target has no alpha)
#endif
}

// ------------------------------------------------------------------------------------------------ launchers
static const dim3 kLinearBlock(64, 4, 1);

mifx_status launch_dof_coc(hipStream_t s, Img depth, Img out, const mifx_camera_attribs& cam, float maxCoC)
{
    // uniform part of ComputeCircleOfConfusionPS (:29-33) in the shader's operation order
    const float f = cam.fFocalLength / 1000.0f;
    DofCocK k;
    k.p10 = cam.mProj[10]; k.p11 = cam.mProj[11]; k.p14 = cam.mProj[14]; k.p15 = cam.mProj[15];
    k.K     = f * f / (cam.fFStop * (cam.fFocusDistance - f));
    k.focus = cam.fFocusDistance;
    k.denom = cam.fSensorWidth * maxCoC;
    hipLaunchKernelGGL(dof_coc_kernel, grid2d(out, kLinearBlock), kLinearBlock, 0, s, depth, out, k);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
mifx_status launch_dof_temporal_coc(hipStream_t s, Img curr, Img prev, Img motion, Img out, const mifx_camera_attribs& cam, float stability)
{
    hipLaunchKernelGGL(dof_temporal_coc_kernel, grid2d(out, kLinearBlock), kLinearBlock, 0, s, curr, prev, motion, out, cam.f4ViewportSize[0], cam.f4ViewportSize[1],
                       cam.f4ViewportSize[2], cam.f4ViewportSize[3], stability);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
// levels[0..2] = dilation levels 1..3 ((W >> k) x (H >> k)); `coc` = the signed CoC that the (never materialised) level 0 is derived from
mifx_status launch_dof_dilation(hipStream_t s, Img coc, const Img levels[3])
{
    const int fused = pyramid_fusable_levels(coc.w, coc.h, 3);
    if (fused > 0)
    {
        DilationOp op{coc, {levels[0], levels[1], levels[2]}};
        hipLaunchKernelGGL(dof_dilation_levels_kernel, dim3((levels[0].w + 15) / 16, (levels[0].h + 15) / 16, 1), dim3(256, 1, 1), 0, s, op, fused);
    }
    for (int l = fused; l < 3; ++l)
    {
        if (l == 0) hipLaunchKernelGGL(dof_dilation_level_kernel<true>, grid2d(levels[0], kLinearBlock), kLinearBlock, 0, s, coc, levels[0]);
        else hipLaunchKernelGGL(dof_dilation_level_kernel<false>, grid2d(levels[l], kLinearBlock), kLinearBlock, 0, s, levels[l - 1], levels[l]);
    }
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
mifx_status launch_dof_blur(hipStream_t s, Img in, Img out, const float weights[13])
{
    GaussK g;
    std::memcpy(g.w, weights, sizeof(g.w));
    hipLaunchKernelGGL(dof_blur_kernel, dim3((out.w + kBlurBX - 1) / kBlurBX, (out.h + kBlurBY - 1) / kBlurBY, 1), dim3(kBlurBX, kBlurBY, 1), 0, s, in, out, g);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
mifx_status launch_dof_prefilter(hipStream_t s, Img color, Img coc, Img dilation, Img outNear, Img outFar)
{
    hipLaunchKernelGGL(dof_prefilter_kernel, grid2d(outNear, kLinearBlock), kLinearBlock, 0, s, color, coc, dilation, outNear, outFar);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
mifx_status launch_dof_bokeh_gather(hipStream_t s, Img nearTex, Img farTex, Img radiance, Img outNear, Img outFar, const float* kernel, int sampleCount, float maxCoC, float aspect,
                                    bool karis)
{
    const BokehK k{kernel, sampleCount, maxCoC, aspect};
    if (karis) hipLaunchKernelGGL(dof_bokeh_gather_kernel<true>, tiled_grid(outNear), dim3(256, 1, 1), 0, s, nearTex, farTex, radiance, outNear, outFar, k);
    else hipLaunchKernelGGL(dof_bokeh_gather_kernel<false>, tiled_grid(outNear), dim3(256, 1, 1), 0, s, nearTex, farTex, radiance, outNear, outFar, k);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
mifx_status launch_dof_bokeh_fill(hipStream_t s, Img nearTex, Img farTex, Img outNear, Img outFar, const float* kernel, int sampleCount, float maxCoC, float aspect)
{
    const BokehK k{kernel, sampleCount, maxCoC, aspect};
    hipLaunchKernelGGL(dof_bokeh_fill_kernel, tiled_grid(outNear), dim3(256, 1, 1), 0, s, nearTex, farTex, outNear, outFar, k);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
mifx_status launch_dof_postfilter(hipStream_t s, Img nearTex, Img farTex, Img outNear, Img outFar)
{
    hipLaunchKernelGGL(dof_postfilter_kernel, grid2d(outNear, kLinearBlock), kLinearBlock, 0, s, nearTex, farTex, outNear, outFar);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
mifx_status launch_dof_combine(hipStream_t s, Img color, Img nearTex, Img farTex, Img out, float alpha)
{
    hipLaunchKernelGGL(dof_combine_kernel, grid2d(out, kLinearBlock), kLinearBlock, 0, s, color, nearTex, farTex, out, alpha);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
} // namespace mifx
