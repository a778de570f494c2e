// ssr_temporal.hip -- ScreenSpaceReflection passes R6 (temporal accumulation) and R7 (bilateral cleanup); R1-R5 are in ssr.hip, R4 in ssr_trace.hip.
// Math follows Shaders/PostProcess/ScreenSpaceReflection/private/SSR_ComputeTemporalAccumulation.fx and SSR_ComputeBilateralCleanup.fx.
// A file of its own so that its floating-point contraction policy can differ from R5's (diligentfx_amd/build.py: FMA_SOURCES).
#include "mifx_host.h"
#include "mifx_effects.h"
#include "mifx_pbr.h"
#include "mifx_ssr_cleanup.h"

namespace mifx
{
// ------------------------------------------------------------------------------------------------ R6: temporal accumulation (SSR_ComputeTemporalAccumulation.fx:104-275)
MIFX_D float ssr_disocclusion(float a, float b) // ComputeDisocclusion :113-118
{
    a = fabsf(a); b = fabsf(b);
    return m_exp(fdiv(-fabsf(a - b), fmaxf(fmaxf(a, b), 1e-6f)));
}
#ifndef MIFX_R6_WAVES
#define MIFX_R6_WAVES 6
#endif
__global__ __launch_bounds__(256) MIFX_WAVES(MIFX_R6_WAVES) void ssr_temporal_kernel(Img motionTex, Img hitDepthTex, Img currDepth /*reprojected*/, Img currRad, Img currVar, Img prevDepth, Img prevRad,
                                                           Img prevVar, Img mask, Img outRad, Img outVar, CamK cur, CamK prev, SsrK k)
{
    int x, y;
    if (!pixel_xy(outRad, x, y)) return;
    if (ld<float>(mask, x, y) == 0.0f)
    {
        st<v4>(outRad, x, y, mk4(0.0f));
        st<var_t>(outVar, x, y, 0.0f);
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
    const float hitDepth = ld<var_t>(hitDepthTex, x, y);
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
        // The 3x3 search keeps only WHICH candidate is the best (and its total weight); the winner's taps and weights are evaluated again afterwards -- the same
        // expressions, so the same values -- instead of carrying four weights and four coordinates through the loop: 94 -> fewer live registers, no scratch
        // spill at 6 waves per SIMD (the search runs for disoccluded pixels only, the second evaluation for those that find a candidate).
        auto candidate = [&](int dx, int dy, v4& w, Bilinear& b) __attribute__((always_inline)) {
            const v2 loc{prevCoord.x + float(dx), prevCoord.y + float(dy)};
            b = bilinear_uc(loc.x, loc.y, currDepth.w, currDepth.h);
            auto ok = [&](int px, int py) __attribute__((always_inline)) { return ssr_disocclusion(currCamZ, depth_to_camera_z(ld<float>(prevDepth, px, py), prev.proj)) > (0.9f / 2.0f) ? 1.0f : 0.0f; };
            w = v4{b.w00 * ok(b.x0, b.y0), b.w10 * ok(b.x1, b.y0), b.w01 * ok(b.x0, b.y1), b.w11 * ok(b.x1, b.y1)};
            return dot(w, mk4(1.0f));
        };
        int   best = -1;
        float bestTotal = 0.0f;
        bool  done = false;
        for (int dy = -1; dy <= 1 && !done; ++dy)
        {
            for (int dx = -1; dx <= 1; ++dx)
            {
                v4       w;
                Bilinear b;
                const float total = candidate(dx, dy, w, b);
                if (total > bestTotal)
                {
                    bestTotal = total;
                    best      = (dy + 1) * 3 + (dx + 1);
                    if (bestTotal > 0.9f) break; // BestTotalWeightEarlyExitThreshold
                }
            }
            if (bestTotal > 0.9f) done = true;
        }
        success = bestTotal > 0.1f;
        if (best >= 0)
        {
            const int bdy = best / 3 - 1, bdx = best - (bdy + 1) * 3 - 1;
            rCoord = v2{prevCoord.x + float(bdx), prevCoord.y + float(bdy)};
            if (success)
            {
                v4       bestW;
                Bilinear b;
                const float total = candidate(bdx, bdy, bestW, b);
                rColor = (ld<v4>(prevRad, b.x0, b.y0) * bestW.x + ld<v4>(prevRad, b.x1, b.y0) * bestW.y + ld<v4>(prevRad, b.x0, b.y1) * bestW.z + ld<v4>(prevRad, b.x1, b.y1) * bestW.w) / total;
            }
        }
    }
    success = success && (rCoord.x >= 0.0f && rCoord.y >= 0.0f && rCoord.x < cur.vw && rCoord.y < cur.vh);

    if (success)
    {
        const v4 cmin = mean - 2.5f * sd, cmax = mean + 2.5f * sd; // SSR_TEMPORAL_VARIANCE_GAMMA
        const v4 pr   = min4(max4(rColor, cmin), cmax);
        const float pv = sample_linear_clamp_f<var_t>(prevVar, rCoord.x * cur.ivw, rCoord.y * cur.ivh);
        st<v4>(outRad, x, y, lerp4(ld<v4>(currRad, x, y), pr, k.TemporalRadianceStabilityFactor));
        st<var_t>(outVar, x, y, lerpf(ld<var_t>(currVar, x, y), pv, k.TemporalVarianceStabilityFactor));
    }
    else
    {
        st<v4>(outRad, x, y, ld<v4>(currRad, x, y));
        st<var_t>(outVar, x, y, 1.0f);
    }
}

// ------------------------------------------------------------------------------------------------ R7: bilateral cleanup (SSR_ComputeBilateralCleanup.fx:49-103; body in mifx_ssr_cleanup.h)
__global__ __launch_bounds__(256) void ssr_bilateral_kernel(Img normalTex, SsrCleanupIn in, Img out, CamK cam)
{
    int x, y;
    if (!pixel_xy(out, x, y)) return;
    st<v4>(out, x, y, ssr_bilateral_cleanup(x, y, xyz(ld<v4>(normalTex, x, y)), normalTex, in, cam.proj, int(cam.vw), int(cam.vh)));
}

static const dim3 kBlock(64, 4, 1);
#define MIFX_LAUNCH_END()              \
    MIFX_HIP_CHECK(hipGetLastError()); \
    return MIFX_OK

mifx_status launch_ssr_temporal(hipStream_t s, Img motion, Img hitDepth, Img reprojDepth, Img currRad, Img currVar, Img prevDepth, Img prevRad, Img prevVar, Img mask, Img outRad,
                                Img outVar, const CamK& cur, const CamK& prev, const mifx_ssr_attribs& a)
{
        hipLaunchKernelGGL(ssr_temporal_kernel, grid2d(outRad, kBlock), kBlock, 0, s, motion, hitDepth, reprojDepth, currRad, currVar, prevDepth, prevRad, prevVar, mask,
                       outRad, outVar, cur, prev, make_k(a, cur.reversedDepth != 0));
    MIFX_LAUNCH_END();
}
mifx_status launch_ssr_bilateral(hipStream_t s, Img normal, const SsrCleanupIn& in, Img out, const CamK& cam)
{
    hipLaunchKernelGGL(ssr_bilateral_kernel, grid2d(out, kBlock), kBlock, 0, s, normal, in, out, cam);
    MIFX_LAUNCH_END();
}
} // namespace mifx
