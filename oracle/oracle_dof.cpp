// oracle_dof.cpp -- TEST INFRASTRUCTURE ONLY: hand-written CPU restatement of the depth-of-field effect
//   D1  circle of confusion        Shaders/PostProcess/DepthOfField/private/DOF_ComputeCircleOfConfusion.fx:24-39
//   D2  temporal CoC               DOF_ComputeTemporalCircleOfConfusion.fx:48-92
//   D3  separated (near) CoC       DOF_ComputeSeparatedCircleOfConfusion.fx:5-11
//   D4  dilation level             DOF_ComputeDilationCircleOfConfusion.fx:8-52
//   D5  Gauss blur of the CoC      DOF_ComputeBlurredCircleOfConfusion.fx:8-28
//   D6  prefiltered near / far     DOF_ComputePrefilteredTexture.fx:23-52
//   D7  bokeh gather               DOF_ComputeBokehFirstPass.fx:49-104
//   D8  bokeh flood fill           DOF_ComputeBokehSecondPass.fx:40-85
//   D9  post filter                DOF_ComputePostfilteredTexture.fx:26-48
//   D10 combine                    DOF_ComputeCombinedTexture.fx:36-46
// and of the two host-side tables (PostProcess/DepthOfField/src/DepthOfField.cpp:49-94).  Same argument blocks as the oracle/_ref wrappers
// (oracle/ref/ref_d*.cpp); pinned against them by tests/test_oracle_vs_ref.py.  All planes fp32 (the reference stores R16 / RGBA16F /
// R11G11B10F intermediates; native formats are SURVEY 8f N4).
#include "oracle_kit.h"

using namespace ok;

namespace
{
struct DofAttribs // DepthOfFieldStructures.fxh:31-56
{
    float   MaxCircleOfConfusion, TemporalStabilityFactor;
    int32_t BokehKernelRingCount, BokehKernelRingDensity;
    float   AlphaInterpolation, Padding0, Padding1, Padding2;
};
static_assert(sizeof(DofAttribs) == 32, "DepthOfFieldAttribs layout");
inline DofAttribs load_attribs(const ref_args* a) { DofAttribs k; std::memcpy(&k, a->attribs, sizeof(k)); return k; }

inline f2 pixel_uv(int x, int y, int w, int h) { return ndc_to_uv({2.0f * ((float(x) + 0.5f) / float(w)) - 1.0f, 1.0f - 2.0f * ((float(y) + 0.5f) / float(h))}); }
inline int sample_count(int rings, int density) { return 1 + density * ((rings - 1) * rings >> 1); } // DOF_Common.fx:4-7
inline float sdr_weight(f3 c) { return 1.0f / (1.0f + luminance601(c)); }                            // DOF_Common.fx:14-17
inline float hdr_weight(f3 c) { return 1.0f + luminance601(c); }                                     // DOF_Common.fx:9-12
inline f3 sample_linear_clamp3(const Img& im, float u, float v) // Texture2D<float3>::SampleLevel on a plane with >= 3 floats per texel
{
    Bilinear b = bilinear_uc(u * float(im.w()), v * float(im.h()), im.w(), im.h());
    return im.ld3(b.x0, b.y0) * b.w00 + im.ld3(b.x1, b.y0) * b.w10 + im.ld3(b.x0, b.y1) * b.w01 + im.ld3(b.x1, b.y1) * b.w11;
}
} // namespace

extern "C" {

// Host tables.  out[0]: kernel points (n x 1, c=2), ival[0] = ring count, ival[1] = ring density (GenerateKernelPoints, DepthOfField.cpp:49-74);
// the rest of the row is zero (KernelData.resize(128), :110)
int oracle_dof_kernel_points(const ref_args* a)
{
    const ref_img& o = a->out[0];
    const int rings = a->ival[0], density = a->ival[1];
    const int count = 1 + density * (rings - 1) * rings / 2;
    if (o.c != 2 || o.h != 1 || o.w < count || rings < 2) return -1;
    std::memset(o.data, 0, sizeof(float) * 2 * size_t(o.w));
    const float radiusInc = 1.0f / (float(rings) - 1.0f);
    int n = 0;
    for (int i = rings - 1; i >= 0; --i)
    {
        const int   points   = std::max(density * i, 1);
        const float radius   = float(i) * radiusInc;
        const float thetaInc = 2.0f * 3.14159265358979323846f / float(points); // PI_F
        const float offset   = 0.1f * float(i);
        for (int j = 0; j < points; ++j, ++n)
        {
            const float theta = offset + float(j) * thetaInc;
            o.data[2 * n + 0] = radius * std::cos(theta);
            o.data[2 * n + 1] = radius * std::sin(theta);
        }
    }
    return n == count ? 0 : -1;
}
// out[0]: (2 * radius + 1) x 1 weights; ival[0] = radius, fval[0] = sigma (GenerateGaussKernel, DepthOfField.cpp:76-94)
int oracle_dof_gauss_kernel(const ref_args* a)
{
    const ref_img& o = a->out[0];
    const int   radius = a->ival[0];
    const float sigma  = a->fval[0];
    if (o.c != 1 || o.h != 1 || o.w != 2 * radius + 1) return -1;
    float sum = 0.0f;
    for (int i = -radius; i <= radius; ++i)
    {
        const float v = std::exp(-float(i * i) / (2.0f * sigma * sigma));
        o.data[i + radius] = v;
        sum += v;
    }
    for (int i = 0; i < o.w; ++i) o.data[i] /= sum;
    return 0;
}

// D1.  in[0]: depth; cam0; attribs; out[0]: signed CoC in [-1, 1] (near < 0 < far)
int oracle_dof_coc(const ref_args* a)
{
    const Camera     cam = load_camera(a->cam0);
    const DofAttribs k   = load_attribs(a);
    const Img depth = in_img(a, 0), out = out_img(a, 0);
#pragma omp parallel for
    for (int y = 0; y < out.h(); ++y)
        for (int x = 0; x < out.w(); ++x)
        {
            const float linearDepth = depth_to_camera_z(depth.ld1z(x, y), cam.proj);
            const float f   = cam.focalLength / 1000.0f;
            const float K   = f * f / (cam.fStop * (cam.focusDistance - f));
            const float coc = K * (linearDepth - cam.focusDistance) / fmax2(linearDepth, 1e-4f);
            out.st1(x, y, clampf(1000.0f * coc / (cam.sensorWidth * k.MaxCircleOfConfusion), -1.0f, 1.0f));
        }
    return 0;
}

// D2.  in: 0 current CoC, 1 previous temporal CoC, 2 closest motion (c=2); cam0; attribs; out[0]
int oracle_dof_temporal_coc(const ref_args* a)
{
    const Camera     cam = load_camera(a->cam0);
    const DofAttribs k   = load_attribs(a);
    const Img curr = in_img(a, 0), prev = in_img(a, 1), motionTex = in_img(a, 2), out = out_img(a, 0);
    const float vw = cam.viewport[0], vh = cam.viewport[1], ivw = cam.viewport[2], ivh = cam.viewport[3];
#pragma omp parallel for
    for (int y = 0; y < out.h(); ++y)
        for (int x = 0; x < out.w(); ++x)
        {
            const f2 pos{float(x) + 0.5f, float(y) + 0.5f};
            const f2 motion  = motionTex.ld2z(x, y) * f2{0.5f, -0.5f}; // F3NDC_XYZ_TO_UVD_SCALE.xy
            const f2 prevPos = pos - motion * f2{vw, vh};
            const float cocCurr = curr.ld1z(x, y);
            if (!(prevPos.x >= 0.0f && prevPos.y >= 0.0f && prevPos.x < vw && prevPos.y < vh)) // IsInsideScreen, PostFX_Common.fxh:121-127
            {
                out.st1(x, y, cocCurr);
                continue;
            }
            const float cocPrev = sample_linear_clamp1(prev, prevPos.x * ivw, prevPos.y * ivh);
            float m1 = 0.0f, m2 = 0.0f; // ComputePixelStatistic :48-72 (point-clamp sampler at texel centres = clamped Load)
            for (int dx = -1; dx <= 1; ++dx)
                for (int dy = -1; dy <= 1; ++dy)
                {
                    const float c = curr.ld1c(x + dx, y + dy);
                    m1 += c;
                    m2 += c * c;
                }
            const float mean = m1 / 9.0f, variance = (m2 / 9.0f) - (mean * mean), stdDev = std::sqrt(fmax2(variance, 0.0f));
            const float cocMin = mean - 2.5f * stdDev, cocMax = mean + 2.5f * stdDev; // DOF_TEMPORAL_VARIANCE_GAMMA
            out.st1(x, y, lerp(cocCurr, clampf(cocPrev, cocMin, cocMax), k.TemporalStabilityFactor));
        }
    return 0;
}

// D3.  in[0]: signed CoC; out[0]: |CoC| of the near field, 0 elsewhere
int oracle_dof_separated_coc(const ref_args* a)
{
    const Img coc = in_img(a, 0), out = out_img(a, 0);
#pragma omp parallel for
    for (int y = 0; y < out.h(); ++y)
        for (int x = 0; x < out.w(); ++x)
        {
            const float c = coc.ld1z(x, y);
            out.st1(x, y, std::fabs(c) * (c < 0.0f ? 1.0f : 0.0f));
        }
    return 0;
}

// D4.  in[0]: previous level; out[0]: max over the 2x2 footprint, extended to 3 texels along an odd dimension
int oracle_dof_dilation_coc(const ref_args* a)
{
    const Img last = in_img(a, 0), out = out_img(a, 0);
    const bool oddW = (last.w() & 1) != 0, oddH = (last.h() & 1) != 0;
#pragma omp parallel for
    for (int y = 0; y < out.h(); ++y)
        for (int x = 0; x < out.w(); ++x)
        {
            auto S = [&](int ox, int oy) { return last.ld1c(2 * x + ox, 2 * y + oy); }; // ClampScreenCoord
            float m = fmax2(fmax2(S(0, 0), S(0, 1)), fmax2(S(1, 0), S(1, 1)));
            if (oddW) m = fmax2(m, fmax2(S(2, 0), S(2, 1)));
            if (oddH) m = fmax2(m, fmax2(S(0, 2), S(1, 2)));
            if (oddW && oddH) m = fmax2(m, S(2, 2));
            out.st1(x, y, m);
        }
    return 0;
}

// D5.  in: 0 CoC, 1 Gauss kernel (13 x 1); ival[0]: 0 = horizontal, 1 = vertical; out[0]
int oracle_dof_blur(const ref_args* a)
{
    const Img coc = in_img(a, 0), kernel = in_img(a, 1), out = out_img(a, 0);
    const bool vertical = a->ival[0] != 0;
    const int  radius   = 6; // DOF_GAUSS_KERNEL_RADIUS
    if (kernel.w() != 2 * radius + 1) return -1;
#pragma omp parallel for
    for (int y = 0; y < out.h(); ++y)
        for (int x = 0; x < out.w(); ++x)
        {
            float sum = 0.0f;
            for (int i = -radius; i <= radius; ++i)
                sum += (vertical ? coc.ld1c(x, y + i) : coc.ld1c(x + i, y)) * kernel.ld1(i + radius, 0);
            out.st1(x, y, sum);
        }
    return 0;
}

// D6.  in: 0 colour, 1 signed CoC, 2 blurred dilation CoC (last level); out: 0 near (rgb, dilated near CoC), 1 far (rgb, far CoC); half resolution
int oracle_dof_prefilter(const ref_args* a)
{
    const Img color = in_img(a, 0), coc = in_img(a, 1), dilation = in_img(a, 2), out0 = out_img(a, 0), out1 = out_img(a, 1);
#pragma omp parallel for
    for (int y = 0; y < out0.h(); ++y)
        for (int x = 0; x < out0.w(); ++x)
        {
            const f2 uv = pixel_uv(x, y, out0.w(), out0.h());
            float cocMax = -3.402823466e+38f;
            f4    sum    = splat4(0.0f);
            for (int i = 0; i < 4; ++i)
            {
                const int lx = 2 * x + (i & 1), ly = 2 * y + (i >> 1);
                const f3    c = color.inside(lx, ly) ? color.ld3(lx, ly) : splat3(0.0f);
                const float w = sdr_weight(c);
                cocMax = fmax2(cocMax, coc.ld1z(lx, ly));
                sum += mk4(c, 1.0f) * w;
            }
            const float fgAlpha = sample_linear_clamp1(dilation, uv.x, uv.y);
            const float bgAlpha = std::fabs(cocMax) * (cocMax > 0.0f ? 1.0f : 0.0f);
            const f3    rgb     = xyz(sum) / fmax2(sum.w, 1.e-5f);
            out0.st4(x, y, mk4(rgb, fgAlpha));
            out1.st4(x, y, mk4(rgb, bgAlpha));
        }
    return 0;
}

// D7.  in: 0 near, 1 far, 2 kernel points (c=2), 3 radiance (Karis variant); cam0; attribs; ival[0]: DOF_OPTION_KARIS_INVERSE; out: 0 near, 1 far
int oracle_dof_bokeh_first(const ref_args* a)
{
    const Camera     cam = load_camera(a->cam0);
    const DofAttribs k   = load_attribs(a);
    const bool karis = a->ival[0] != 0;
    const Img nearTex = in_img(a, 0), farTex = in_img(a, 1), kernel = in_img(a, 2), out0 = out_img(a, 0), out1 = out_img(a, 1);
    const Img radiance = karis ? in_img(a, 3) : nearTex;
    const float aspect = cam.viewport[0] * cam.viewport[3];
    const int   count  = sample_count(k.BokehKernelRingCount, k.BokehKernelRingDensity);
    if (count > kernel.w()) return -1;
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < out0.h(); ++y)
        for (int x = 0; x < out0.w(); ++x)
        {
            const f2    uv      = pixel_uv(x, y, out0.w(), out0.h());
            const float cocNear = sample_linear_clamp4(nearTex, uv.x, uv.y).w;
            const float cocFar  = sample_linear_clamp4(farTex, uv.x, uv.y).w;
            f4 fg = splat4(0.0f), bg = splat4(0.0f);
            if (cocNear > 0.0f)
                for (int i = 0; i < count; ++i)
                {
                    const f2 sp = 0.5f * kernel.ld2(i, 0) * cocNear * k.MaxCircleOfConfusion;
                    const f2 st{sp.x, aspect * sp.y};
                    const f4 c = sample_linear_clamp4(nearTex, uv.x + st.x, uv.y + st.y);
                    const float w = karis ? hdr_weight(sample_linear_clamp3(radiance, uv.x + st.x, uv.y + st.y)) : 1.0f;
                    fg += mk4(xyz(c), 1.0f) * w;
                }
            if (cocFar > 0.0f)
                for (int i = 0; i < count; ++i)
                {
                    const f2 sp = 0.5f * kernel.ld2(i, 0) * cocFar * k.MaxCircleOfConfusion;
                    const f2 st{sp.x, aspect * sp.y};
                    const f4 c = sample_linear_clamp4(farTex, uv.x + st.x, uv.y + st.y);
                    const float w = karis ? hdr_weight(sample_linear_clamp3(radiance, uv.x + st.x, uv.y + st.y)) : 1.0f;
                    bg += mk4(xyz(c), 1.0f) * w * (c.w >= cocFar ? 1.0f : 0.0f);
                }
            out0.st4(x, y, mk4(xyz(fg) * (1.0f / (fg.w + (fg.w == 0.0f ? 1.0f : 0.0f))), cocNear));
            out1.st4(x, y, mk4(xyz(bg) * (1.0f / (bg.w + (bg.w == 0.0f ? 1.0f : 0.0f))), cocFar));
        }
    return 0;
}

// D8.  in: 0 near, 1 far (D7 outputs), 2 small kernel (c=2); cam0; attribs; out: 0 near, 1 far
int oracle_dof_bokeh_second(const ref_args* a)
{
    const Camera     cam = load_camera(a->cam0);
    const DofAttribs k   = load_attribs(a);
    const Img nearTex = in_img(a, 0), farTex = in_img(a, 1), kernel = in_img(a, 2), out0 = out_img(a, 0), out1 = out_img(a, 1);
    const float aspect = cam.viewport[0] * cam.viewport[3];
    const int   count  = sample_count(3, 5); // DOF_BOKEH_KERNEL_SMALL_RING_COUNT / _DENSITY
    if (count > kernel.w()) return -1;
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < out0.h(); ++y)
        for (int x = 0; x < out0.w(); ++x)
        {
            const f2 uv = pixel_uv(x, y, out0.w(), out0.h());
            f4 fg = sample_linear_clamp4(nearTex, uv.x, uv.y), bg = sample_linear_clamp4(farTex, uv.x, uv.y);
            const float cocNear = fg.w, cocFar = bg.w;
            f3 fgc = xyz(fg), bgc = xyz(bg);
            if (cocNear > 0.0f)
                for (int i = 0; i < count; ++i)
                {
                    const f2 sp = 0.25f * kernel.ld2(i, 0) * cocNear * k.MaxCircleOfConfusion;
                    const f4 c  = sample_linear_clamp4(nearTex, uv.x + sp.x, uv.y + aspect * sp.y);
                    fgc = max3(xyz(c), fgc);
                }
            if (cocFar > 0.0f)
                for (int i = 0; i < count; ++i)
                {
                    const f2 sp = 0.25f * kernel.ld2(i, 0) * cocFar * k.MaxCircleOfConfusion;
                    const f4 c  = sample_linear_clamp4(farTex, uv.x + sp.x, uv.y + aspect * sp.y);
                    bgc = max3(xyz(c) * (c.w >= cocFar ? 1.0f : 0.0f), bgc);
                }
            out0.st4(x, y, mk4(fgc, cocNear));
            out1.st4(x, y, mk4(bgc, cocFar));
        }
    return 0;
}

// D9.  in: 0 near, 1 far; out: 0 near, 1 far
int oracle_dof_postfilter(const ref_args* a)
{
    for (int t = 0; t < 2; ++t)
    {
        const Img in = in_img(a, t), out = out_img(a, t);
        const f2  ts{1.0f / float(in_img(a, 0).w()), 1.0f / float(in_img(a, 0).h())}; // g_TextureColorCoCNear.GetDimensions
#pragma omp parallel for
        for (int y = 0; y < out.h(); ++y)
            for (int x = 0; x < out.w(); ++x)
            {
                const f2 uv = pixel_uv(x, y, out.w(), out.h());
                auto S = [&](float ox, float oy) { return sample_linear_clamp4(in, uv.x + ts.x * ox, uv.y + ts.y * oy); };
                const f4 A = S(-0.5f, -0.5f), B = S(-0.5f, +0.5f), C = S(+0.5f, -0.5f), D = S(+0.5f, +0.5f);
                out.st4(x, y, 0.25f * (A + B + C + D));
            }
    }
    return 0;
}

// D10.  in: 0 colour, 1 CoC (unused), 2 near, 3 far (D9 outputs); attribs; out[0]: rgb, a = alpha of the colour input
int oracle_dof_combine(const ref_args* a)
{
    const DofAttribs k = load_attribs(a);
    const Img color = in_img(a, 0), nearTex = in_img(a, 2), farTex = in_img(a, 3), out = out_img(a, 0);
#pragma omp parallel for
    for (int y = 0; y < out.h(); ++y)
        for (int x = 0; x < out.w(); ++x)
        {
            const f2 uv  = pixel_uv(x, y, out.w(), out.h());
            const f3 src = color.ld3(x, y);
            const f4 n = sample_linear_clamp4(nearTex, uv.x, uv.y), f = sample_linear_clamp4(farTex, uv.x, uv.y);
            f3 r = src;
            r = lerp(r, xyz(f), smoothstep(0.1f, 1.0f, f.w));
            r = lerp(r, xyz(n), smoothstep(0.1f, 1.0f, n.w));
            out.st4(x, y, mk4(lerp(src, r, k.AlphaInterpolation), color.im->c > 3 ? color.px(x, y)[3] : 1.0f));
        }
    return 0;
}

} // extern "C"
