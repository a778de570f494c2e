// mifx_ssr_cleanup.h -- ScreenSpaceReflection pass R7, the bilateral cleanup (SSR_ComputeBilateralCleanup.fx:49-103), as a device function shared by
//   * ssr_bilateral_kernel (ssr_temporal.hip): the pass on its own, writing the effect's output plane, and
//   * composite_kernel<.., true> (pbr.hip): the chain's composite, which is the only consumer of that plane and evaluates R7 for its own pixel instead of reading it
//     (one float4 plane less written and read per frame, one launch less).
// The two translation units have different contraction policies (build.py FMA_SOURCES: pbr.hip fuses multiply-adds, ssr_temporal.hip does not) and R7's weights sit
// behind thresholds and steep exponentials (exp(-|dz| / |grad|), pow(N.N', 128)): a fused multiply-add in them moves texels.  Every multiply-add of the pass is
// therefore written out in this function's own body under `#pragma clang fp contract(off)` -- the pragma is lexical, it does not reach into helpers that are
// defined elsewhere, so the function calls only helpers without a contractible multiply-add (loads, fdiv, min / max, the hardware exp / pow wrappers).
// tests/test_gpu_chain.py: test_chain_fusion_is_bit_identical holds the two users to the same bits.
#pragma once
#include "mifx_device.h"
#include "mifx_effects.h"

namespace mifx
{
struct SsrCleanupIn // what R7 reads beside the normal: views of the SSR effect's planes of this frame (row windows do not matter: read-only)
{
    Img   depth, roughness, radiance, variance, mask;
    float RoughnessThreshold, BilateralCleanupSpatialSigmaFactor, AlphaInterpolation;
    int   ReversedDepth;
};

// R7 for the pixel (x, y); N = the G-buffer normal of the pixel, maskValue = the reflection mask at the pixel (the callers fetch both beside their own first loads: one
// dependent round trip less than fetching them here).  W, H: f4ViewportSize.xy as integers.
MIFX_D v4 ssr_bilateral_cleanup(int x, int y, v3 N, float maskValue, const Img& normalTex, const SsrCleanupIn& in, const m44& proj, int W, int H)
{
#pragma clang fp contract(off)
    if (maskValue == 0.0f) return v4{0.0f, 0.0f, 0.0f, 0.0f}; // target cleared to 0 (ScreenSpaceReflection.cpp:1099)
    auto camera_z = [&](float d) __attribute__((always_inline)) { return fdiv(proj.m[14] - d * proj.m[15], d * proj.m[11] - proj.m[10]); }; // DepthToCameraZ (ShaderUtilities.fxh:5-40)
    const float rough = ld<rough_t>(in.roughness, x, y);
    const float var   = ld<var_t>(in.variance, x, y);
    const float camZ  = camera_z(ld<float>(in.depth, x, y));
    // ddx/ddy of CameraZ (:57): fine derivatives inside the 2x2 pixel quad (right - left, bottom - top); quad lanes outside the image
    // replicate the nearest pixel.  Same convention as the oracle's quad emulation.
    auto cz = [&](int px, int py) __attribute__((always_inline)) { return camera_z(ld<float>(in.depth, px < W ? px : W - 1, py < H ? py : H - 1)); };
    const int   qx = x & ~1, qy = y & ~1;
    const float gradX = cz(qx + 1, y) - cz(qx, y), gradY = cz(x, qy + 1) - cz(x, qy);

    const float roughTarget = saturate(8.0f * rough);                                      // SSR_BILATERAL_ROUGHNESS_FACTOR
    const float radius = 0.0f + roughTarget * ((var > 0.001f ? 2.0f : 0.0f) - 0.0f);       // lerp(0, ..., roughTarget); SSS_BILATERAL_VARIANCE_ESTIMATE_THRESHOLD
    const float sigma  = in.BilateralCleanupSpatialSigmaFactor;
    const int   er     = int(fminf(2.0f * sigma, radius));
    v4 result = ld<v4>(in.radiance, x, y);
    if (var > 0.00005f && er > 0) // SSR_BILATERAL_VARIANCE_EXIT_THRESHOLD
    {
        float sumX = 0.0f, sumY = 0.0f, sumZ = 0.0f, sumW = 0.0f, wsum = 0.0f;
        for (int dx = -er; dx <= er; ++dx)
            for (int dy = -er; dy <= er; ++dy)
            {
                const int sx = clampi(x + dx, 0, W - 1), sy = clampi(y + dy, 0, H - 1);
                const float sd = ld<float>(in.depth, sx, sy);
                const float sr = ld<rough_t>(in.roughness, sx, sy);
                if (is_reflection_sample(sr, sd, in.RoughnessThreshold, in.ReversedDepth != 0))
                {
                    const v4 srad = ld<v4>(in.radiance, sx, sy);
                    const v4 sn   = ld<v4>(normalTex, sx, sy);
                    const float sz = camera_z(sd);
                    const float ox = float(dx), oy = float(dy);
                    const float ws = m_exp(fdiv(-0.5f * (ox * ox + oy * oy), sigma * sigma));
                    const float wz = m_exp(fdiv(-fabsf(camZ - sz), 1.0f * (fabsf(ox * gradX + oy * gradY) + 1e-6f))); // SSR_BILATERAL_SIGMA_DEPTH
                    const float wn = m_pow(fmaxf(0.0f, N.x * sn.x + N.y * sn.y + N.z * sn.z), 128.0f);                // SSR_BILATERAL_SIGMA_NORMAL
                    const float w  = ws * wn * wz;
                    wsum += w;
                    sumX = sumX + w * srad.x; sumY = sumY + w * srad.y; sumZ = sumZ + w * srad.z; sumW = sumW + w * srad.w;
                }
            }
        const float den = fmaxf(wsum, 1.0e-6f);
        result = v4{fdiv(sumX, den), fdiv(sumY, den), fdiv(sumZ, den), fdiv(sumW, den)};
    }
    return v4{result.x, result.y, result.z, result.w * in.AlphaInterpolation};
}
} // namespace mifx
