// ssr.hip -- ScreenSpaceReflection passes R1..R7 (AMD-SSSR-derived stochastic screen-space reflections).
// Math follows Shaders/PostProcess/ScreenSpaceReflection/private/SSR_*.fx; host sequence in api_ssr.cpp.
//
// Masking: the reference marks reflection samples in a D16 depth target and depth-tests R4-R7 against it
// (ScreenSpaceReflection.cpp:47-51,550,626,...).  Here the mask is a float plane (1 = reflection sample) and every masked pass writes 0
// to masked-out texels (the reference clears R4/R7 targets to 0 and leaves R5/R6 targets stale; stale data is undefined, 0 is our contract).
#include "mifx_host.h"
#include "mifx_effects.h"
#include "mifx_pyramid.h"
#include "mifx_pbr.h"

namespace mifx
{


// ------------------------------------------------------------------------------------------------ R1: Hi-Z mip (SSR_ComputeHierarchicalDepthBuffer.fx:24-71)
__global__ __launch_bounds__(256) void ssr_hiz_mip_kernel(Img src, Img dst, int reversed)
{
    int x, y;
    if (!pixel_xy(dst, x, y)) return;
    const int  rx = 2 * x, ry = 2 * y;
    const bool oddW = (src.w & 1) != 0, oddH = (src.h & 1) != 0;
    float m = reversed ? 0.0f : 1.0f; // DepthFarPlane
    auto  tap = [&](int ox, int oy) { m = closest_depth(m, ld_clamp<float>(src, rx + ox, ry + oy), reversed != 0); };
    tap(0, 0); tap(0, 1); tap(1, 0); tap(1, 1);
    if (oddW) { tap(2, 0); tap(2, 1); }
    if (oddH) { tap(0, 2); tap(1, 2); }
    if (oddW && oddH) tap(2, 2);
    st<float>(dst, x, y, m);
}

struct HizOp
{
    using T = float;
    int reversed;           // SSR_OPTION_INVERTED_DEPTH: closest = largest depth, far plane = 0
    Img src, dst[4], copy0; // copy0.p != null: the source level is also written out (level 0 of the hierarchy = a copy of the depth buffer)
    MIFX_D float load(int x, int y) const
    {
        const float v = ld<float>(src, x, y);
        if (copy0.p) st<float>(copy0, x, y, v); // every source texel is read by exactly one thread (even dimensions)
        return v;
    }
    MIFX_D float reduce(float a, float b, float c, float d) const // DepthFarPlane = 1 (0 when reversed)
    {
        return reversed ? fmaxf(fmaxf(fmaxf(fmaxf(0.0f, a), b), c), d) : fminf(fminf(fminf(fminf(1.0f, a), b), c), d);
    }
    MIFX_D bool  inside(int l, int x, int y) const { return x < dst[l - 1].w && y < row_end(dst[l - 1]); }
    MIFX_D int   first_block_row() const { return dst[0].y0 >> 4; }
    MIFX_D void  store(int l, int x, int y, float v) const { st<float>(dst[l - 1], x, y, v); }
};
__global__ __launch_bounds__(256) void ssr_hiz_levels_kernel(HizOp op, int nl) { pyramid_reduce_levels(op, nl); }

// ------------------------------------------------------------------------------------------------ R2: mask + roughness (SSR_ComputeStencilMaskAndExtractRoughness.fx:13-40)
__global__ __launch_bounds__(256) void ssr_mask_roughness_kernel(Img material, Img depthTex, Img roughnessOut, Img maskOut, SsrK k)
{
    int x, y;
    if (!pixel_xy(maskOut, x, y)) return;
    const v4 m = ld<v4>(material, x, y);
    const v4 sel{k.RoughnessChannel == 0u ? 1.0f : 0.0f, k.RoughnessChannel == 1u ? 1.0f : 0.0f, k.RoughnessChannel == 2u ? 1.0f : 0.0f, k.RoughnessChannel == 3u ? 1.0f : 0.0f};
    float r = dot(m, sel);
    if (!k.IsRoughnessPerceptual) r = fsqrt(r);
    const float d = ld<float>(depthTex, x, y);
    st<float>(roughnessOut, x, y, r); // every texel (the reference leaves non-sample texels stale)
    st<float>(maskOut, x, y, is_reflection_sample(r, d, k.RoughnessThreshold, k.ReversedDepth != 0) ? 1.0f : 0.0f);
}

// ------------------------------------------------------------------------------------------------ R3: half-resolution mask (SSR_ComputeDownsampledStencilMask.fx:13-61)
__global__ __launch_bounds__(256) void ssr_downsampled_mask_kernel(Img roughnessTex, Img depthTex, Img maskOut, SsrK k)
{
    int x, y;
    if (!pixel_xy(maskOut, x, y)) return;
    const bool rev = k.ReversedDepth != 0, oddW = (depthTex.w & 1) != 0, oddH = (depthTex.h & 1) != 0;
    float minDepth = rev ? 0.0f : 1.0f, maxRough = 0.0f; // DepthFarPlane
    auto tap = [&](int ox, int oy) {
        const int lx = clampi(2 * x + ox, 0, depthTex.w - 1), ly = clampi(2 * y + oy, 0, depthTex.h - 1); // ClampScreenCoord
        minDepth = closest_depth(minDepth, ld<float>(depthTex, lx, ly), rev);
        maxRough = fmaxf(maxRough, ld<float>(roughnessTex, lx, ly));
    };
    tap(0, 0); tap(1, 0); tap(0, 1); tap(1, 1);
    if (oddW) { tap(2, 0); tap(2, 1); }
    if (oddH) { tap(0, 2); tap(1, 2); }
    if (oddW && oddH) tap(2, 2);
    st<float>(maskOut, x, y, is_reflection_sample(maxRough, minDepth, k.RoughnessThreshold, rev) ? 1.0f : 0.0f);
}

// ------------------------------------------------------------------------------------------------ R5: spatial reconstruction (SSR_ComputeSpatialReconstruction.fx:60-175)
static constexpr float c_ssr_poisson[8][3] = {{-0.4706069f, -0.4427112f, +0.6461146f}, {-0.9057375f, +0.3003471f, +0.9542373f}, {-0.3487388f, +0.4037880f, +0.5335386f},
                                          {+0.1023042f, +0.6439373f, +0.6520134f}, {+0.5699277f, +0.3513750f, +0.6695386f}, {+0.2939128f, -0.1131226f, +0.3149309f},
                                          {+0.7836658f, -0.4208784f, +0.8895339f}, {+0.1564120f, -0.8198990f, +0.8346850f}};

// HALF = SSR_OPTION_HALF_RESOLUTION: the Poisson taps address the half-size ray textures (:153-154)
template <bool HALF> __global__ __launch_bounds__(256) void ssr_spatial_kernel(Img roughnessTex, Img normalTex, Img depthTex, Img dirPdfTex, Img specTex, Img mask, Img outRad, Img outVar,
                                                          Img outDepth, CamK cam, SsrK k)
{
    int x, y;
    if (!pixel_xy(outRad, x, y)) return;
    if (ld<float>(mask, x, y) == 0.0f)
    {
        st<v4>(outRad, x, y, mk4(0.0f));
        st<float>(outVar, x, y, 0.0f);
        st<float>(outDepth, x, y, 0.0f);
        return;
    }
    const int W = int(cam.vw), H = int(cam.vh);
    const v2 pos{float(x) + 0.5f, float(y) + 0.5f};
    const v3 camPos{cam.pos[0], cam.pos[1], cam.pos[2]};
    const v3 posWS  = inv_project_position(v3{pos.x * cam.ivw, pos.y * cam.ivh, ld<float>(depthTex, x, y)}, cam.viewProjInv);
    const v3 N      = xyz(ld<v4>(normalTex, x, y));
    const v3 V      = normalize(camPos - posWS);
    const float NdotV = saturate(dot(N, V));
    const float rough = ld<float>(roughnessTex, x, y);
    const float radius = lerpf(0.0f, k.SpatialReconstructionRadius, saturate(5.0f * rough)); // SSR_SPATIAL_RECONSTRUCTION_ROUGHNESS_FACTOR
    const float angle = 2.0f * MIFX_PI * bayer4x4(unsigned(x), unsigned(y), cam.frameIndex);
    // note: ComputeBlurKernelRotation uses M_PI (3.14159265358979) -- same fp32 value as MIFX_PI
    float sinA, cosA;
    m_sincos(angle, sinA, cosA); // angle in [0, 2 pi)
    const v4 rot{cosA, sinA, -sinA, cosA};

    const float alpha = rough * rough;
    const float visV  = smith_ggx_visibility_v_term(NdotV, alpha); // per-pixel factor of the visibility term, hoisted out of the 8-sample loop
    v4    colorSum = mk4(0.0f);
    float weightSum = 0.0f, variance = 0.0f, mean = 0.0f;
    float nearestHit = 0.0f;
#pragma unroll
    for (int s = 0; s < 8; ++s)
    {
        const v2  xi = rotate_vector(rot, v2{c_ssr_poisson[s][0], c_ssr_poisson[s][1]});
        const int sx = HALF ? clampi(int(0.5f * (floorf(pos.x) + radius * xi.x) + 0.5f), 0, int(0.5f * cam.vw) - 1) : clampi(int(pos.x + radius * xi.x), 0, W - 1);
        const int sy = HALF ? clampi(int(0.5f * (floorf(pos.y) + radius * xi.y) + 0.5f), 0, int(0.5f * cam.vh) - 1) : clampi(int(pos.y + radius * xi.y), 0, H - 1);
        const float ws = spatial_weight_const(c_ssr_poisson[s][2] * c_ssr_poisson[s][2], 0.9f);
        // ComputeWeightRayLength :60-88
        float wgt, rayLen;
        {
            const v4    dp  = ld<v4>(dirPdfTex, sx, sy);
            const float len = length(xyz(dp));
            if (len < 1e-6f) { wgt = 1e-6f; rayLen = 1e-6f; }
            else
            {
                const v3    L = xyz(dp) / len;
                const v3    Hh = normalize(L + V);
                const float NdotH = saturate(dot(N, Hh)), NdotL = saturate(dot(N, L));
                const float vis = smith_ggx_visibility_correlated_v(NdotL, NdotV, alpha, visV);
                const float D   = normal_distribution_ggx(NdotH, alpha);
                float brdf = vis * D * NdotL;
                brdf *= ws;
                wgt    = fmaxf(fdiv(brdf, fmaxf(dp.w, 1e-5f)), 1e-6f);
                rayLen = len;
            }
        }
        const v4 c = ld<v4>(specTex, sx, sy);
        // ComputeWeightedVariance :90-100
        colorSum  = colorSum + wgt * c;
        weightSum += wgt;
        const float value = luminance601(xyz(c));
        const float prevMean = mean;
        mean += wgt * fdiv(1.0f, weightSum) * (value - prevMean);
        variance += wgt * (value - prevMean) * (value - mean);
        if (wgt > 1.0e-6f) nearestHit = fmaxf(rayLen, nearestHit);
    }
    st<v4>(outRad, x, y, colorSum / fmaxf(weightSum, 1e-6f));
    st<float>(outVar, x, y, fdiv(variance, fmaxf(weightSum, 1e-6f)));
    // ComputeResolvedDepth :102-106
    st<float>(outDepth, x, y, camera_z_to_depth(length(camPos - posWS) + nearestHit, cam.proj));
}

// ------------------------------------------------------------------------------------------------ R6: temporal accumulation (SSR_ComputeTemporalAccumulation.fx:104-275)
MIFX_D float ssr_disocclusion(float a, float b) // ComputeDisocclusion :113-118
{
    a = fabsf(a); b = fabsf(b);
    return m_exp(fdiv(-fabsf(a - b), fmaxf(fmaxf(a, b), 1e-6f)));
}
__global__ __launch_bounds__(256) MIFX_WAVES(6) void ssr_temporal_kernel(Img motionTex, Img hitDepthTex, Img currDepth /*reprojected*/, Img currRad, Img currVar, Img prevDepth, Img prevRad,
                                                           Img prevVar, Img mask, Img outRad, Img outVar, CamK cur, CamK prev, SsrK k)
{
    int x, y;
    if (!pixel_xy(outRad, x, y)) return;
    if (ld<float>(mask, x, y) == 0.0f)
    {
        st<v4>(outRad, x, y, mk4(0.0f));
        st<float>(outVar, x, y, 0.0f);
        return;
    }
    const int W = int(cur.vw), H = int(cur.vh);
    const v2 pos{float(x) + 0.5f, float(y) + 0.5f};
    // ComputePixelStatistic :122-145
    v4 m1 = mk4(0.0f), m2 = mk4(0.0f);
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
        {
            const v4 c = ld<v4>(currRad, clampi(x + dx, 0, W - 1), clampi(y + dy, 0, H - 1));
            m1 += c;
            m2 += c * c;
        }
    const v4 mean = m1 / 9.0f;
    const v4 sd   = sqrt4(max4((m2 / 9.0f) - (mean * mean), 0.0f));

    const float depth    = ld<float>(currDepth, x, y);
    const float hitDepth = ld<float>(hitDepthTex, x, y);
    const v2 mraw = ld<v2>(motionTex, x, y);
    const v2 motion{mraw.x * 0.5f, mraw.y * -0.5f};
    const v2 prevIncident{pos.x - motion.x * cur.vw, pos.y - motion.y * cur.vh};
    // ComputeReflectionHitPosition :104-110
    v2 prevHit;
    {
        const v2 tc{(float(x) + 0.5f) * cur.ivw + 0.5f * cur.jx, (float(y) + 0.5f) * cur.ivh + -0.5f * cur.jy};
        const v3 pw = inv_project_position(v3{tc.x, tc.y, hitDepth}, cur.viewProjInv);
        const v3 pc = project_position(pw, prev.viewProj);
        prevHit = v2{(pc.x - 0.5f * prev.jx) * cur.vw, (pc.y - -0.5f * prev.jy) * cur.vh};
    }
    auto sample_prev_rad = [&](v2 p) { return sample_linear_clamp_v4(prevRad, p.x * cur.ivw, p.y * cur.ivh); };
    const v4 cInc = sample_prev_rad(prevIncident), cHit = sample_prev_rad(prevHit);
    const float meanLum = luminance601(xyz(mean));
    const float dInc = fabsf(luminance601(xyz(cInc)) - meanLum), dHit = fabsf(luminance601(xyz(cHit)) - meanLum);
    const v2 prevCoord = dInc < dHit ? prevIncident : prevHit;

    // ComputeReprojection :147-222
    const float currCamZ = depth_to_camera_z(depth, cur.proj);
    v2   rCoord = prevCoord;
    v4   rColor = sample_prev_rad(prevCoord);
    bool success;
    {
        const float pz = depth_to_camera_z(ld_zero_f(prevDepth, int(prevCoord.x), int(prevCoord.y)), prev.proj);
        success = ssr_disocclusion(currCamZ, pz) > 0.9f; // SSR_DISOCCLUSION_THRESHOLD
    }
    if (!success)
    {
        v4 bestW = mk4(0.0f);
        int bx0 = 0, by0 = 0, bx1 = 0, by1 = 0;
        float bestTotal = 0.0f;
        bool  done = false;
        for (int dy = -1; dy <= 1 && !done; ++dy)
        {
            for (int dx = -1; dx <= 1; ++dx)
            {
                const v2 loc{prevCoord.x + float(dx), prevCoord.y + float(dy)};
                const Bilinear b = bilinear_uc(loc.x, loc.y, currDepth.w, currDepth.h);
                auto ok = [&](int px, int py) { return ssr_disocclusion(currCamZ, depth_to_camera_z(ld<float>(prevDepth, px, py), prev.proj)) > (0.9f / 2.0f) ? 1.0f : 0.0f; };
                const v4 w{b.w00 * ok(b.x0, b.y0), b.w10 * ok(b.x1, b.y0), b.w01 * ok(b.x0, b.y1), b.w11 * ok(b.x1, b.y1)};
                const float total = dot(w, mk4(1.0f));
                if (total > bestTotal)
                {
                    bestTotal = total; bestW = w; bx0 = b.x0; by0 = b.y0; bx1 = b.x1; by1 = b.y1;
                    rCoord = loc;
                    if (bestTotal > 0.9f) break; // BestTotalWeightEarlyExitThreshold
                }
            }
            if (bestTotal > 0.9f) done = true;
        }
        success = bestTotal > 0.1f;
        if (success)
            rColor = (ld<v4>(prevRad, bx0, by0) * bestW.x + ld<v4>(prevRad, bx1, by0) * bestW.y + ld<v4>(prevRad, bx0, by1) * bestW.z + ld<v4>(prevRad, bx1, by1) * bestW.w) / bestTotal;
    }
    success = success && (rCoord.x >= 0.0f && rCoord.y >= 0.0f && rCoord.x < cur.vw && rCoord.y < cur.vh);

    if (success)
    {
        const v4 cmin = mean - 2.5f * sd, cmax = mean + 2.5f * sd; // SSR_TEMPORAL_VARIANCE_GAMMA
        const v4 pr   = min4(max4(rColor, cmin), cmax);
        const float pv = sample_linear_clamp_f(prevVar, rCoord.x * cur.ivw, rCoord.y * cur.ivh);
        st<v4>(outRad, x, y, lerp4(ld<v4>(currRad, x, y), pr, k.TemporalRadianceStabilityFactor));
        st<float>(outVar, x, y, lerpf(ld<float>(currVar, x, y), pv, k.TemporalVarianceStabilityFactor));
    }
    else
    {
        st<v4>(outRad, x, y, ld<v4>(currRad, x, y));
        st<float>(outVar, x, y, 1.0f);
    }
}

// ------------------------------------------------------------------------------------------------ R7: bilateral cleanup (SSR_ComputeBilateralCleanup.fx:49-103)
__global__ __launch_bounds__(256) void ssr_bilateral_kernel(Img depthTex, Img normalTex, Img roughnessTex, Img radTex, Img varTex, Img mask, Img out, CamK cam, SsrK k)
{
    int x, y;
    if (!pixel_xy(out, x, y)) return;
    if (ld<float>(mask, x, y) == 0.0f)
    {
        st<v4>(out, x, y, mk4(0.0f)); // target cleared to 0 (ScreenSpaceReflection.cpp:1099)
        return;
    }
    const int W = int(cam.vw), H = int(cam.vh);
    const float rough = ld<float>(roughnessTex, x, y);
    const float var   = ld<float>(varTex, x, y);
    const v3    N     = xyz(ld<v4>(normalTex, x, y));
    const float camZ  = depth_to_camera_z(ld<float>(depthTex, x, y), cam.proj);
    // ddx/ddy of CameraZ (:57): fine derivatives inside the 2x2 pixel quad (right - left, bottom - top); quad lanes outside the image
    // replicate the nearest pixel.  Same convention as the oracle's quad emulation.
    auto cz = [&](int px, int py) { return depth_to_camera_z(ld<float>(depthTex, px < W ? px : W - 1, py < H ? py : H - 1), cam.proj); };
    const int qx = x & ~1, qy = y & ~1;
    const v2  grad{cz(qx + 1, y) - cz(qx, y), cz(x, qy + 1) - cz(x, qy)};

    const float roughTarget = saturate(8.0f * rough); // SSR_BILATERAL_ROUGHNESS_FACTOR
    const float radius = lerpf(0.0f, var > 0.001f ? 2.0f : 0.0f, roughTarget); // SSS_BILATERAL_VARIANCE_ESTIMATE_THRESHOLD
    const float sigma  = k.BilateralCleanupSpatialSigmaFactor;
    const int   er     = int(fminf(2.0f * sigma, radius));
    v4 result = ld<v4>(radTex, x, y);
    if (var > 0.00005f && er > 0) // SSR_BILATERAL_VARIANCE_EXIT_THRESHOLD
    {
        v4 colorSum = mk4(0.0f);
        float wsum = 0.0f;
        for (int dx = -er; dx <= er; ++dx)
            for (int dy = -er; dy <= er; ++dy)
            {
                const int sx = clampi(x + dx, 0, W - 1), sy = clampi(y + dy, 0, H - 1);
                const float sd = ld<float>(depthTex, sx, sy);
                const float sr = ld<float>(roughnessTex, sx, sy);
                if (is_reflection_sample(sr, sd, k.RoughnessThreshold, k.ReversedDepth != 0))
                {
                    const v4 srad = ld<v4>(radTex, sx, sy);
                    const v3 sn   = xyz(ld<v4>(normalTex, sx, sy));
                    const float sz = depth_to_camera_z(sd, cam.proj);
                    const v2 o{float(dx), float(dy)};
                    const float ws = m_exp(fdiv(-0.5f * dot(o, o), sigma * sigma));
                    const float wz = m_exp(fdiv(-fabsf(camZ - sz), 1.0f * (fabsf(dot(o, grad)) + 1e-6f))); // SSR_BILATERAL_SIGMA_DEPTH
                    const float wn = m_pow(fmaxf(0.0f, dot(N, sn)), 128.0f);                               // SSR_BILATERAL_SIGMA_NORMAL
                    const float w  = ws * wn * wz;
                    wsum += w;
                    colorSum += w * srad;
                }
            }
        result = colorSum / fmaxf(wsum, 1.0e-6f);
    }
    st<v4>(out, x, y, v4{result.x, result.y, result.z, result.w * k.AlphaInterpolation});
}

// ------------------------------------------------------------------------------------------------ launchers
static const dim3 kBlock(64, 4, 1);
#define MIFX_LAUNCH_END()              \
    MIFX_HIP_CHECK(hipGetLastError()); \
    return MIFX_OK

mifx_status launch_ssr_hiz_pyramid(hipStream_t s, const Pyr& p, Img level0Copy, bool reversedDepth) // p.l[0] = depth; fills p.l[1 .. levels - 1] and the copy of level 0
{
    bool copied = false;
    for (int k = 1; k < p.levels;)
    {
        const int nl = pyramid_fusable_levels(p.l[k - 1].w, p.l[k - 1].h, p.levels - k);
        if (k == 1 && nl < 2)
        {
            MIFX_HIP_CHECK(hipMemcpy2DAsync(level0Copy.p, size_t(level0Copy.pitch), p.l[0].p, size_t(p.l[0].pitch), size_t(p.l[0].w) * 4u, size_t(p.l[0].h), hipMemcpyDeviceToDevice, s));
            copied = true;
        }
        if (nl >= 2)
        {
            HizOp op{};
            op.reversed = reversedDepth ? 1 : 0;
            op.src = p.l[k - 1];
            if (k == 1) { op.copy0 = level0Copy; copied = true; }
            for (int j = 0; j < nl; ++j) op.dst[j] = p.l[k + j];
            hipLaunchKernelGGL(ssr_hiz_levels_kernel, dim3((p.l[k].w + 15) / 16, (p.l[k].h + 15) / 16, 1), dim3(256, 1, 1), 0, s, op, nl);
            k += nl;
        }
        else
        {
            hipLaunchKernelGGL(ssr_hiz_mip_kernel, grid2d(p.l[k], kBlock), kBlock, 0, s, p.l[k - 1], p.l[k], reversedDepth ? 1 : 0);
            ++k;
        }
        MIFX_HIP_CHECK(hipGetLastError());
    }
    if (!copied) MIFX_HIP_CHECK(hipMemcpy2DAsync(level0Copy.p, size_t(level0Copy.pitch), p.l[0].p, size_t(p.l[0].pitch), size_t(p.l[0].w) * 4u, size_t(p.l[0].h), hipMemcpyDeviceToDevice, s));
    return MIFX_OK;
}
mifx_status launch_ssr_mask_roughness(hipStream_t s, Img material, Img depth, Img roughness, Img mask, const mifx_ssr_attribs& a, bool reversedDepth)
{
    hipLaunchKernelGGL(ssr_mask_roughness_kernel, grid2d(mask, kBlock), kBlock, 0, s, material, depth, roughness, mask, make_k(a, reversedDepth));
    MIFX_LAUNCH_END();
}
mifx_status launch_ssr_downsampled_mask(hipStream_t s, Img roughness, Img depth, Img mask, const mifx_ssr_attribs& a, bool reversedDepth)
{
    hipLaunchKernelGGL(ssr_downsampled_mask_kernel, grid2d(mask, kBlock), kBlock, 0, s, roughness, depth, mask, make_k(a, reversedDepth, true));
    MIFX_LAUNCH_END();
}
mifx_status launch_ssr_spatial(hipStream_t s, Img roughness, Img normal, Img depth, Img dirPdf, Img spec, Img mask, Img outRad, Img outVar, Img outDepth, const CamK& cam,
                               const mifx_ssr_attribs& a, bool halfResolution)
{
    const SsrK k = make_k(a, cam.reversedDepth != 0, halfResolution);
    if (halfResolution) hipLaunchKernelGGL(ssr_spatial_kernel<true>, grid2d(outRad, kBlock), kBlock, 0, s, roughness, normal, depth, dirPdf, spec, mask, outRad, outVar, outDepth, cam, k);
    else hipLaunchKernelGGL(ssr_spatial_kernel<false>, grid2d(outRad, kBlock), kBlock, 0, s, roughness, normal, depth, dirPdf, spec, mask, outRad, outVar, outDepth, cam, k);
    MIFX_LAUNCH_END();
}
mifx_status launch_ssr_temporal(hipStream_t s, Img motion, Img hitDepth, Img reprojDepth, Img currRad, Img currVar, Img prevDepth, Img prevRad, Img prevVar, Img mask, Img outRad,
                                Img outVar, const CamK& cur, const CamK& prev, const mifx_ssr_attribs& a)
{
        hipLaunchKernelGGL(ssr_temporal_kernel, grid2d(outRad, kBlock), kBlock, 0, s, motion, hitDepth, reprojDepth, currRad, currVar, prevDepth, prevRad, prevVar, mask,
                       outRad, outVar, cur, prev, make_k(a, cur.reversedDepth != 0));
    MIFX_LAUNCH_END();
}
mifx_status launch_ssr_bilateral(hipStream_t s, Img depth, Img normal, Img roughness, Img rad, Img var, Img mask, Img out, const CamK& cam, const mifx_ssr_attribs& a)
{
    hipLaunchKernelGGL(ssr_bilateral_kernel, grid2d(out, kBlock), kBlock, 0, s, depth, normal, roughness, rad, var, mask, out, cam, make_k(a, cam.reversedDepth != 0));
    MIFX_LAUNCH_END();
}
} // namespace mifx
