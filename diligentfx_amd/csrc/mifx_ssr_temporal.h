// mifx_ssr_temporal.h -- ScreenSpaceReflection pass R6, the temporal accumulation (SSR_ComputeTemporalAccumulation.fx:104-275), for one pixel: the body of
// ssr_temporal_kernel (ssr_temporal.hip), in a header so that the test suite can also compile it for the host (tests/host_kernels/chain_host.cpp).
#pragma once
#include "mifx_device.h"
#include "mifx_effects.h"

namespace mifx
{
MIFX_D float ssr_disocclusion(float a, float b) // ComputeDisocclusion :113-118
{
    a = fabsf(a); b = fabsf(b);
    return m_exp(fdiv(-fabsf(a - b), fmaxf(fmaxf(a, b), 1e-6f)));
}

MIFX_D void ssr_temporal_pixel(int x, int y, const Img& motionTex, const Img& hitDepthTex, const Img& currDepth /*reprojected*/, const Img& currRad, const Img& currVar, const Img& prevDepth,
                               const Img& prevRad, const Img& prevVar, const Img& mask, const Img& outRad, const Img& outVar, const CamK& cur, const CamK& prev, const SsrK& k)
{
    if (ld_once<mask_t>(mask, x, y) == 0.0f) return; // (the history slot keeps what the frame before last left there: the reference's depth test skips the fragment; see ssr_spatial_kernel)
    const int W = int(cur.vw), H = int(cur.vh);
    const v2 pos{float(x) + 0.5f, float(y) + 0.5f};
    // Memory-level parallelism (round 3; the counters showed the waves of this pass parked on s_waitcnt for 76 % of their cycles at 11 % of the VALU issue roof): the
    // loads are grouped by what they depend on and each group is issued before anything of it is consumed --
    //   (1) what depends on the pixel alone: depth, hit depth, motion, the centre variance, the 3x3 neighbourhood of the resolved radiance;
    //   (2) what depends on the two candidate positions (incident point / reflection hit, both follow from (1)): the 2 x 4 history radiance texels, the history depth
    //       at both, the 2 x 4 history variance texels -- the reference samples the chosen candidate a second time (:168, :214-217), which is the same value;
    // so a pixel without disocclusion makes four or five round trips instead of ten.  The arithmetic on the fetched values is unchanged (same taps, weights, order).
    // Measured (tools/ab_gpu.sh, profiles/r03_ab_mlp.txt): 142.6 -> 127.5 us.  Forcing each group behind one wait (keep_here on all 13 / 18 values) was measured
    // too and is slower (133 us; 146 at 5 waves per SIMD): the compiler's interleaving keeps the registers for six waves.
    const float depth    = ld_once<float>(currDepth, x, y);
    const float hitDepth = ld_once<var_t>(hitDepthTex, x, y);
    const v2    mraw     = ld_once<v2>(motionTex, x, y);
    const float currVarC = ld_once<var_t>(currVar, x, y);
    // ComputePixelStatistic :122-145
    v4 m1 = mk4(0.0f), m2 = mk4(0.0f), currRadC = mk4(0.0f);
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
        {
            const v4 c = ld<v4>(currRad, clampi(x + dx, 0, W - 1), clampi(y + dy, 0, H - 1));
            if (dx == 0 && dy == 0) currRadC = c;
            m1 += c;
            m2 += c * c;
        }
    const v4 mean = m1 / 9.0f;
    const v4 sd   = sqrt4(max4((m2 / 9.0f) - (mean * mean), 0.0f));

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
    // group (2): both candidates at once
    const BilinearTaps tI = bilinear_taps<kV4Bytes>(prevRad, prevIncident.x * cur.ivw, prevIncident.y * cur.ivh), tH = bilinear_taps<kV4Bytes>(prevRad, prevHit.x * cur.ivw, prevHit.y * cur.ivh);
    const BilinearTaps vI = bilinear_taps<TexelBytes<var_t>::value>(prevVar, prevIncident.x * cur.ivw, prevIncident.y * cur.ivh),
                       vH = bilinear_taps<TexelBytes<var_t>::value>(prevVar, prevHit.x * cur.ivw, prevHit.y * cur.ivh);
    const v4 i00 = ld_at<v4>(prevRad, tI.o00), i10 = ld_at<v4>(prevRad, tI.o10), i01 = ld_at<v4>(prevRad, tI.o01), i11 = ld_at<v4>(prevRad, tI.o11);
    const v4 h00 = ld_at<v4>(prevRad, tH.o00), h10 = ld_at<v4>(prevRad, tH.o10), h01 = ld_at<v4>(prevRad, tH.o01), h11 = ld_at<v4>(prevRad, tH.o11);
    const float pdI = ld_zero_f_nb(prevDepth, int(prevIncident.x), int(prevIncident.y)), pdH = ld_zero_f_nb(prevDepth, int(prevHit.x), int(prevHit.y));
    const float a00 = ld_at<var_t>(prevVar, vI.o00), a10 = ld_at<var_t>(prevVar, vI.o10), a01 = ld_at<var_t>(prevVar, vI.o01), a11 = ld_at<var_t>(prevVar, vI.o11);
    const float b00 = ld_at<var_t>(prevVar, vH.o00), b10 = ld_at<var_t>(prevVar, vH.o10), b01 = ld_at<var_t>(prevVar, vH.o01), b11 = ld_at<var_t>(prevVar, vH.o11);
    auto blend4 = [](const BilinearTaps& b, v4 t00, v4 t10, v4 t01, v4 t11) __attribute__((always_inline)) { // == sample_linear_clamp_v4_taps on fetched texels
        MIFX_FMA_BLOCK
        return v4{t00.x * b.w00 + t10.x * b.w10 + t01.x * b.w01 + t11.x * b.w11, t00.y * b.w00 + t10.y * b.w10 + t01.y * b.w01 + t11.y * b.w11,
                  t00.z * b.w00 + t10.z * b.w10 + t01.z * b.w01 + t11.z * b.w11, t00.w * b.w00 + t10.w * b.w10 + t01.w * b.w01 + t11.w * b.w11};
    };
    const v4 cInc = blend4(tI, i00, i10, i01, i11), cHit = blend4(tH, h00, h10, h01, h11);
    const float meanLum = luminance601(xyz(mean));
    const float dInc = fabsf(luminance601(xyz(cInc)) - meanLum), dHit = fabsf(luminance601(xyz(cHit)) - meanLum);
    const bool  incident  = dInc < dHit;
    const v2    prevCoord = incident ? prevIncident : prevHit;
    // the variance at the chosen candidate (== sample_linear_clamp_f_taps on the fetched texels)
    const float pvInc = a00 * vI.w00 + a10 * vI.w10 + a01 * vI.w01 + a11 * vI.w11, pvHit = b00 * vH.w00 + b10 * vH.w10 + b01 * vH.w01 + b11 * vH.w11;
    float pv = incident ? pvInc : pvHit;

    // ComputeReprojection :147-222
    const float currCamZ = depth_to_camera_z(depth, cur.proj);
    v2   rCoord = prevCoord;
    v4   rColor = incident ? cInc : cHit; // (the reference's second SampleLevel at the chosen position)
    bool success;
    {
        const float pz = depth_to_camera_z(incident ? pdI : pdH, prev.proj);
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
            pv = sample_linear_clamp_f<var_t>(prevVar, rCoord.x * cur.ivw, rCoord.y * cur.ivh); // (the search moved the position: the variance is taken there, :214-217)
        }
    }
    success = success && (rCoord.x >= 0.0f && rCoord.y >= 0.0f && rCoord.x < cur.vw && rCoord.y < cur.vh);

    if (success)
    {
        const v4 cmin = mean - 2.5f * sd, cmax = mean + 2.5f * sd; // SSR_TEMPORAL_VARIANCE_GAMMA
        const v4 pr   = min4(max4(rColor, cmin), cmax);
        st<v4>(outRad, x, y, lerp4(currRadC, pr, k.TemporalRadianceStabilityFactor));
        st<var_t>(outVar, x, y, lerpf(currVarC, pv, k.TemporalVarianceStabilityFactor));
    }
    else
    {
        st<v4>(outRad, x, y, currRadC);
        st<var_t>(outVar, x, y, 1.0f);
    }
}
} // namespace mifx
