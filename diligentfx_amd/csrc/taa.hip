// taa.hip -- TemporalAntiAliasing (T1).
//   T1 Shaders/PostProcess/TemporalAntiAliasing/private/TAA_ComputeTemporalAccumulation.fx:34-262
// The history is sampled with linear CLAMP (TemporalAntiAliasing.cpp:234), reproduced in software with exact fp32 weights (mifx_device.h).
#include "mifx_host.h"
#include "mifx_composite.h"

namespace mifx
{
mifx_status make_lutk(const mifx_image2d* im, LutK& k); // pbr.hip

// ------------------------------------------------------------------------------------------------ T1
template <bool YCOCG> MIFX_D v3 rgb_to_ycocg(v3 c) // :34-49
{
    if (!YCOCG) return c;
    const float co = c.x - c.z;
    const float t  = c.z + 0.5f * co;
    const float cg = c.y - t;
    const float yy = t + 0.5f * cg;
    return v3{yy, co, cg};
}
template <bool YCOCG> MIFX_D v3 ycocg_to_rgb(v3 c) // :51-66
{
    if (!YCOCG) return c;
    const float t = c.x - 0.5f * c.z;
    const float g = c.z + t;
    const float b = t - 0.5f * c.y;
    const float r = b + c.y;
    return v3{r, g, b};
}
MIFX_D v3 hdr_to_sdr(v3 c) { return c * (mk3(1.0f) / (mk3(1.0f) + c)); }                          // :68-71  Color * rcp(1 + Color)
MIFX_D v3 sdr_to_hdr(v3 c) { return c * (mk3(1.0f) / (mk3(1.0f) - c + mk3(5.960464478e-8f))); }  // :73-76  Color * rcp(1 - Color + FLT_EPS)

// Workgroup = 32x8 output texels.  The 3x3 colour statistic needs SDR(YCoCg(max(colour, 0))) of nine texels per pixel -- three divisions and
// the colour transform each; the block converts its 34x10 footprint once into LDS (clamp addressing applied at fill time) and the statistic
// reads the tile: same per-texel arithmetic, 1.3 conversions per pixel instead of 9.
// (Round 4, measured and not taken: the history taps out of LDS as well -- per block the min / max of floor(prevPos - 0.5) over its pixels (wave shuffles + one LDS
//  exchange), the window [min - 2, max + 3]^2 of the accumulated colour and of the previous depth copied in with row-contiguous loads (48 x 16 texels: 20.8 KB with the
//  colour tile), the 20 + 9 taps as LDS reads, blocks with incoherent motion on the gather path; bit-identical, all 16 TAA comparisons and the chain / sharding suites
//  green.  The idea: a CU's vector L1 serves one tag look-up per clock (tools/microbench/tcp_gather_rate.hip) and this pass' 74 M look-ups per launch are 120 of its
//  169 us.  Result: 215 us with the window, 191 us for the same kernel with the window switched off, 169 us for this one (profiles/r04_ab_taa_window.txt): the block
//  reduction, the second barrier and the fill cost more than the look-ups they replace, as for SSR's R5 (ssr.hip).  Also seen on the way: a branch per tap instead of one
//  per group of taps serialises the 20 loads (+50 us), and five instead of seven resident workgroups per CU cost 30 %.)
constexpr int kTaaBX = 32, kTaaBY = 8, kTaaTW = kTaaBX + 2, kTaaTH = kTaaBY + 2;
// Round 5: the chain's composite (M1, with SSR's cleanup R7 inside: mifx_composite.h) evaluated HERE, for the 34 x 10 texels of the block's colour tile, instead of a
// pass of its own that writes a plane this kernel is the only reader of.  COMPOSITE = false: `currColor` is read (the stand-alone effect, the chain with the fusion
// off, TAA's placeholder frame).  COMPOSITE = true: `currColor` is not read; SampleCurrColor(texel) = max(composite_pixel(texel).rgb, 0) with exactly the stored value
// (quantize_v4: nothing in the fp32 build, the RGBA16_FLOAT rounding of the plane in the native-storage build).  One plane less written and read per frame
// (32 B/px), one launch less, and -- what the three-lane schedule could not arrange from outside -- the composite's streaming loads and this pass' gathers share
// a CU workgroup by workgroup.  The price: the 84 halo texels of a block are composited twice (340 evaluations per 256 pixels).
struct TaaCompositeIn
{
    Img   color, specIBL, ssao, normalTex, baseColor, material;
    LutK  lut;
    CamK  cam;
    float ssrScale, ssaoScale;
    int   outW, outH;
    SsrCleanupIn r7;
};
template <bool GAUSS, bool BICUBIC, bool YCOCG, bool COMPOSITE>
#ifndef MIFX_TAA_WAVES
#define MIFX_TAA_WAVES 5
#endif
__global__ __launch_bounds__(256) MIFX_WAVES(MIFX_TAA_WAVES) void taa_kernel(Img currColor, Img prevColor, Img motionTex, Img currDepth /*reprojected*/, Img prevDepth, Img out, CamK cur, CamK prev,
                                                  float stability, int reset, int skipRejection, TaaCompositeIn ci)
{
    __shared__ v4 tile[kTaaTH * kTaaTW];
    const int by0 = int(blockIdx.y) * kTaaBY + out.y0; // first row of this block (row window of `out`)
    const int x = blockIdx.x * kTaaBX + threadIdx.x;
    const int y = by0 + int(threadIdx.y);
    const int W = int(cur.vw), H = int(cur.vh);
    auto sample_curr = [&](int px, int py) { // SampleCurrColor :78-81
        if (COMPOSITE)
        {
            v4 c;
            composite_pixel<MIFX_TONE_MAPPING_MODE_NONE, true>(c, px, py, ci.color, ci.specIBL, Img{}, ci.ssao, ci.normalTex, ci.baseColor, ci.material, ci.lut, ci.outW, ci.outH, ci.cam, ci.ssrScale,
                                                               ci.ssaoScale, ToneMapK{}, ci.r7);
            return max3(xyz(quantize_v4(c)), 0.0f);
        }
        return max3(xyz(ld<v4>(currColor, px, py)), 0.0f);
    };
    // Round 5: the kernel is a chain of dependent memory round trips (tools/isa_roundtrips.py), and until now five and a half of them: the motion vector (the compiler
    // had sunk its first use into the `inImage` branch: a full wait on the first load of the kernel), the tile texels one per loop iteration (two for a third of the
    // threads, which the barrier makes everybody's), the 20 history taps, and -- sunk below the uniform SkipRejection branch, i.e. behind the last history tap -- the
    // nine previous-depth taps.  Now: motion, depth and BOTH tile texels of the thread are requested together and branch-free (clamped coordinates for threads outside
    // the image, whose values nobody uses); the nine depth taps are requested as soon as the motion vector is there, before the tile is converted, and pinned in front
    // of the barrier; the history taps follow it.  Same loads, same arithmetic, three round trips.
    const bool inImage = x < out.w && y < row_end(out);
    v2    m;
    float cd;
    v3    ownColor = mk3(0.0f); // COMPOSITE: SampleCurrColor of this thread's own pixel (the tile holds it converted only)
    float dtap[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}; // (!COMPOSITE) the previous-depth taps around int(PrevPosition), dy outer / dx inner
    {
        const int ox = blockIdx.x * kTaaBX - 1, oy = by0 - 1;
        if (COMPOSITE)
        {
            m  = inImage ? ld<cm_t>(motionTex, x, y) : v2{0.0f, 0.0f};
            cd = inImage ? ld<float>(currDepth, x, y) : 0.0f;
            // every thread its own texel first (kept for the paths below that write the current colour as it is), then the 84 texels of the one-texel frame around
            // the block: top row, bottom row, left column, right column
            auto fill = [&](int tx, int ty) {
                const v3 c = sample_curr(clampi(ox + tx, 0, W - 1), clampi(oy + ty, 0, H - 1));
                tile[ty * kTaaTW + tx] = mk4(rgb_to_ycocg<YCOCG>(hdr_to_sdr(c)), 0.0f);
                return c;
            };
            ownColor = fill(int(threadIdx.x) + 1, int(threadIdx.y) + 1);
            const int hh = int(threadIdx.y) * kTaaBX + int(threadIdx.x);
            if (hh < 2 * kTaaTW + 2 * kTaaBY)
            {
                const int tx = hh < kTaaTW ? hh : hh < 2 * kTaaTW ? hh - kTaaTW : hh < 2 * kTaaTW + kTaaBY ? 0 : kTaaTW - 1;
                const int ty = hh < kTaaTW ? 0 : hh < 2 * kTaaTW ? kTaaTH - 1 : hh < 2 * kTaaTW + kTaaBY ? 1 + hh - 2 * kTaaTW : 1 + hh - 2 * kTaaTW - kTaaBY;
                (void)fill(tx, ty);
            }
        }
        else
        {
            const int xc = min(x, out.w - 1), yc = min(y, row_end(out) - 1);
            m  = ld_once<cm_t>(motionTex, xc, yc);
            cd = ld_once<float>(currDepth, xc, yc);
            // tile texels i0 = thread index and i1 = i0 + 256 (the second only for the first 84 threads; the others repeat their first: one more hit on a line they
            // have just asked for, no branch between the loads)
            const int  i0 = int(threadIdx.y) * kTaaBX + int(threadIdx.x);
            const bool two = i0 + kTaaBX * kTaaBY < kTaaTW * kTaaTH;
            const int  i1 = two ? i0 + kTaaBX * kTaaBY : i0;
            v4 raw0 = ld<v4>(currColor, clampi(ox + i0 % kTaaTW, 0, W - 1), clampi(oy + i0 / kTaaTW, 0, H - 1));
            v4 raw1 = ld<v4>(currColor, clampi(ox + i1 % kTaaTW, 0, W - 1), clampi(oy + i1 / kTaaTW, 0, H - 1));
            keep_here(m); // (the motion vector is the oldest request: this waits for it alone)
            {
                const v2 mo{m.x * 0.5f, m.y * -0.5f};
                const v2 pp{(float(x) + 0.5f) - mo.x * cur.vw, (float(y) + 0.5f) - mo.y * cur.vh};
                const int pxi = int(pp.x), pyi = int(pp.y);
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) dtap[(dy + 1) * 3 + dx + 1] = ld_zero_f_nb(prevDepth, pxi + dx, pyi + dy);
            }
            keep_here(raw0);
            keep_here(raw1);
            tile[i0] = mk4(rgb_to_ycocg<YCOCG>(hdr_to_sdr(max3(xyz(raw0), 0.0f))), 0.0f);
            if (two) tile[i1] = mk4(rgb_to_ycocg<YCOCG>(hdr_to_sdr(max3(xyz(raw1), 0.0f))), 0.0f);
        }
#ifdef MIFX_TAA_LDS_BARRIER
        // the workgroup exchanges LDS data only: wait for this wave's LDS writes and meet the others -- __syncthreads() would also wait for every global load in flight
        // (its workgroup-scope fence covers global memory), i.e. for the depth taps just requested
        if (!COMPOSITE) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else __syncthreads();
#else
        __syncthreads();
#endif
        if (!COMPOSITE)
            for (int i = 0; i < 9; ++i) keep_here(dtap[i]); // (requested above: pinned here so that the loads are not sunk to their use)
    }
    if (x >= out.w || y >= row_end(out)) return;
    auto tile_at = [&](int dx, int dy) { return xyz(tile[(int(threadIdx.y) + 1 + dy) * kTaaTW + int(threadIdx.x) + 1 + dx]); };
    const v2 pos{float(x) + 0.5f, float(y) + 0.5f};
    const v2 motion{m.x * 0.5f, m.y * -0.5f};
    const v2 prevPos{pos.x - motion.x * cur.vw, pos.y - motion.y * cur.vh};

    const bool inside = prevPos.x >= 0.0f && prevPos.y >= 0.0f && prevPos.x < cur.vw && prevPos.y < cur.vh;
    if (!inside || reset)
    {
        st<v4>(out, x, y, mk4(COMPOSITE ? ownColor : sample_curr(x, y), 0.5f));
        return;
    }
    const float aspect       = cur.vw * cur.ivh;
    const float motionFactor = saturate(1.0f - length(v2{motion.x * aspect, motion.y}) * 256.0f); // TAA_MOTION_VECTOR_DIFF_FACTOR

    // ComputeDepthDisocclusion :117-136 (3x3 around int(PrevPosition), unclamped loads -> 0).  Only the threshold on the maximum is used:
    //   max_i exp(-|lc - lp_i| / max(lc, lp_i, 1e-6)) > 0.9   <=>   exists i : |lc - lp_i| < k * max(lc, lp_i, 1e-6),  k = -ln(0.9)
    //                                                          <=>   exists i : lc * (1 - k) < lp_i < lc / (1 - k)
    // and, camera z being a monotonic function of the stored depth (DepthToCameraZ, ShaderUtilities.fxh:33-40; the 0 of an out-of-bounds load included),
    //                                                          <=>   exists i : the previous DEPTH d_i lies strictly between the depths of those two camera z.
    // Two conversions per pixel instead of nine, no exp, no division per tap; the forms can only disagree for a tap within one rounding error of a bound (the
    // camera z the reference computes from d_i carries the same rounding).
    bool similar = false;
    {
        const int   pxi = int(prevPos.x), pyi = int(prevPos.y);
        const float zc  = depth_to_camera_z(cd, cur.proj);
        constexpr float k = 0.105360515657826f;
        const float da = camera_z_to_depth(zc * (1.0f - k), prev.proj), db = camera_z_to_depth(fdiv(zc, 1.0f - k), prev.proj);
        const float dlo = fminf(da, db), dhi = fmaxf(da, db);
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx)
            {
                const float pd = COMPOSITE ? ld_zero_f_nb(prevDepth, pxi + dx, pyi + dy) : dtap[(dy + 1) * 3 + dx + 1];
                similar = similar || (pd > dlo && pd < dhi);
            }
    }
    const float depthFactor = similar ? 1.0f : 0.0f; // TAA_DEPTH_DISOCCLUSION_THRESHOLD = 0.9
    v4 prevRGBA;
    if (BICUBIC)
    {
        // SamplePrevColorCatmullRom :138-173 (5 bilinear taps)
        const v2 texel{cur.ivw, cur.ivh};
        const v2 centre{floorf(prevPos.x - 0.5f) + 0.5f, floorf(prevPos.y - 0.5f) + 0.5f};
        const v2 f = prevPos - centre, f2 = f * f, f3 = f2 * f;
        const v2 w0 = -0.5f * f3 + f2 - 0.5f * f;
        const v2 w1 = 1.5f * f3 - 2.5f * f2 + 1.0f;
        const v2 w2 = -1.5f * f3 + 2.0f * f2 + 0.5f * f;
        const v2 w3 = 0.5f * f3 - 0.5f * f2;
        const v2 w12 = w1 + w2;
        const v2 tp0  = (centre - 1.0f) * texel;
        const v2 tp3  = (centre + 2.0f) * texel;
        const v2 tp12 = (centre + w2 / w12) * texel;
        const float p0 = w12.x * w0.y, p1 = w0.x * w12.y, p2 = w12.x * w12.y, p3 = w3.x * w12.y, p4 = w12.x * w3.y;
        v4 r = mk4(0.0f);
        r += sample_linear_clamp_v4(prevColor, tp12.x, tp0.y) * p0;
        r += sample_linear_clamp_v4(prevColor, tp0.x, tp12.y) * p1;
        r += sample_linear_clamp_v4(prevColor, tp12.x, tp12.y) * p2;
        r += sample_linear_clamp_v4(prevColor, tp3.x, tp12.y) * p3;
        r += sample_linear_clamp_v4(prevColor, tp12.x, tp3.y) * p4;
        prevRGBA = max4(r * fdiv(1.0f, p0 + p1 + p2 + p3 + p4), 0.0f);
    }
    else
    {
        prevRGBA = max4(sample_linear_clamp_v4(prevColor, prevPos.x * cur.ivw, prevPos.y * cur.ivh), 0.0f); // SamplePrevColorBilinear :175-178
    }

    const v3 currY = tile_at(0, 0); // rgb_to_ycocg(hdr_to_sdr(SampleCurrColor(x, y)))
    const v3 prevY = rgb_to_ycocg<YCOCG>(hdr_to_sdr(xyz(prevRGBA)));
    auto corrected_alpha = [&](float a) { return fminf(stability, saturate(fdiv(1.0f, 2.0f - a))); }; // ComputeCorrectedAlpha :224-227

    if (skipRejection)
    {
        const v3 o = sdr_to_hdr(ycocg_to_rgb<YCOCG>(lerp3(currY, prevY, prevRGBA.w)));
        st<v4>(out, x, y, mk4(o, corrected_alpha(prevRGBA.w)));
        return;
    }
    const float varianceGamma = lerpf(0.75f, 2.5f, motionFactor * motionFactor); // TAA_MIN/MAX_VARIANCE_GAMMA

    // ComputePixelStatisticYCoCgSDR :191-222 (3x3, clamped, x outer / y inner)
    float wsum = 0.0f;
    v3 m1 = mk3(0.0f), m2 = mk3(0.0f);
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
        {
            const v3    sdr = tile_at(dx, dy);
            const float w   = GAUSS ? expf(-3.0f * float(dx * dx + dy * dy) / ((1.0f + 1.0f) * (1.0f + 1.0f))) : 1.0f;
            m1 += sdr * w;
            m2 += sdr * sdr * w;
            wsum += w;
        }
    const v3 mean = m1 / wsum;
    const v3 var  = m2 / wsum - (mean * mean);
    const v3 sd   = sqrt3(max3(var, 0.0f));

    // ClipToAABB :98-106 (relies on min() ignoring NaN when the colour delta is zero)
    const float maxT = 10.0f; // TAA_VARIANCE_INTERSECTION_MAX_T
    const v3 extents = varianceGamma * sd;
    const v3 dir     = currY - prevY;
    const v3 sgn{signf(dir.x), signf(dir.y), signf(dir.z)};
    const v3 isect   = ((mean - sgn * extents) - prevY) / dir;
    auto sel = [&](float i) { float ge = i >= 0.0f ? 1.0f : 0.0f; return (maxT + 1.0f) + ge * (i - (maxT + 1.0f)); }; // lerp(MaxT+1, Intersection, GreaterEqual(Intersection, 0))
    const v3 possible{sel(isect.x), sel(isect.y), sel(isect.z)};
    const float T = fminf(maxT, fminf(possible.x, fminf(possible.y, possible.z)));
    const float lt = T < maxT ? 1.0f : 0.0f;
    const v3 clamped = prevY + lt * ((prevY + dir * T) - prevY); // lerp(ColorPrev, ColorPrev + Direction * T, Less(T, MaxT))

    const float alpha = prevRGBA.w * motionFactor * depthFactor;
    const v3 o = sdr_to_hdr(ycocg_to_rgb<YCOCG>(lerp3(currY, clamped, alpha)));
    st<v4>(out, x, y, mk4(o, corrected_alpha(alpha)));
}

mifx_status launch_taa(hipStream_t s, Img currColor, Img prevColor, Img motion, Img reprojDepth, Img prevDepth, Img out, const CamK& cur, const CamK& prev,
                       const mifx_taa_attribs& a, uint32_t flags, const TaaFusedComposite* fused)
{
    const dim3 block(kTaaBX, kTaaBY, 1), grid = grid2d(out, block);
    TaaCompositeIn ci{};
    if (fused)
    {
        // what launch_composite (composite.hip) validates and builds for the pass on its own
        const mifx_composite_attribs& ca = *fused->attribs;
        const uint32_t W = uint32_t(out.w), H = uint32_t(out.h);
        MIFX_CHECK(to_img_wh(ca.color, MIFX_FORMAT_F32X4, W, H, "color", ci.color));
        MIFX_CHECK(to_img_wh(ca.specular_ibl, MIFX_FORMAT_F32X4, W, H, "specular_ibl", ci.specIBL));
        MIFX_CHECK(to_img_wh(ca.ssao, MIFX_PLANE_AO, W, H, "ssao", ci.ssao));
        MIFX_CHECK(to_img_wh(ca.normal, MIFX_FORMAT_F32X4, W, H, "normal", ci.normalTex));
        MIFX_CHECK(to_img_wh(ca.base_color, MIFX_FORMAT_F32X4, W, H, "base_color", ci.baseColor));
        MIFX_CHECK(to_img_wh(ca.material, MIFX_FORMAT_F32X4, W, H, "material", ci.material));
        MIFX_REQUIRE(ca.camera != nullptr && fused->r7 != nullptr, "fused composite: camera and cleanup inputs must not be null");
        MIFX_REQUIRE(ca.tone_mapping == nullptr || ca.tone_mapping->iToneMappingMode == MIFX_TONE_MAPPING_MODE_NONE, "fused composite: the chain composites without a tone map while TAA is on");
        MIFX_CHECK(make_lutk(ca.brdf_lut, ci.lut));
        ci.cam = make_camk(*ca.camera);
        ci.ssrScale = ca.ssr_scale; ci.ssaoScale = ca.ssao_scale;
        ci.outW = out.w; ci.outH = out.h;
        ci.r7 = *fused->r7;
    }
#define MIFX_TAA_C(G, B, Y, C) hipLaunchKernelGGL((taa_kernel<G, B, Y, C>), grid, block, 0, s, currColor, prevColor, motion, reprojDepth, prevDepth, out, cur, prev, \
                                                  a.TemporalStabilityFactor, a.ResetAccumulation, a.SkipRejection, ci)
#define MIFX_TAA(G, B, Y) do { if (fused) MIFX_TAA_C(G, B, Y, true); else MIFX_TAA_C(G, B, Y, false); } while (0)
    switch (flags & 7u)
    {
        case 0: MIFX_TAA(false, false, false); break;
        case 1: MIFX_TAA(true, false, false); break;
        case 2: MIFX_TAA(false, true, false); break;
        case 3: MIFX_TAA(true, true, false); break;
        case 4: MIFX_TAA(false, false, true); break;
        case 5: MIFX_TAA(true, false, true); break;
        case 6: MIFX_TAA(false, true, true); break;
        default: MIFX_TAA(true, true, true); break;
    }
#undef MIFX_TAA
#undef MIFX_TAA_C
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
} // namespace mifx
