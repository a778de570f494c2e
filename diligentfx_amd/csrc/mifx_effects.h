// mifx_effects.h -- the attribs of the SSR / SSAO effects as the kernels take them (by value), shared by the translation units of each effect
// (ssr.hip + ssr_trace.hip, ssao.hip + ssao_ao.hip): ScreenSpaceReflectionAttribs / ScreenSpaceAmbientOcclusionAttribs plus the effect's
// compile-time options of the reference that are run-time uniforms here.
#pragma once
#include <cmath>

#include "mifx.h"
#include "mifx_device.h"

namespace mifx
{
struct SsrK
{
    float    DepthBufferThickness, RoughnessThreshold;
    unsigned MostDetailedMip;
    int      IsRoughnessPerceptual;
    unsigned RoughnessChannel, MaxTraversalIntersections;
    float    GGXImportanceSampleBias, SpatialReconstructionRadius, TemporalRadianceStabilityFactor, TemporalVarianceStabilityFactor;
    float    BilateralCleanupSpatialSigmaFactor, AlphaInterpolation;
    int      ReversedDepth; // SSR_OPTION_INVERTED_DEPTH
    int      HalfResolution; // SSR_OPTION_HALF_RESOLUTION: the ray textures and their mask are (W / 2) x (H / 2)
};
inline SsrK make_k(const mifx_ssr_attribs& a, bool reversedDepth, bool halfResolution = false)
{
    return SsrK{a.DepthBufferThickness, a.RoughnessThreshold, a.MostDetailedMip, a.IsRoughnessPerceptual, a.RoughnessChannel, a.MaxTraversalIntersections,
                a.GGXImportanceSampleBias, a.SpatialReconstructionRadius, a.TemporalRadianceStabilityFactor, a.TemporalVarianceStabilityFactor,
                a.BilateralCleanupSpatialSigmaFactor, a.AlphaInterpolation, reversedDepth ? 1 : 0, halfResolution ? 1 : 0};
}
#define SSR_MAX_MIP 6
#define SSR_FLT_EPS 5.960464478e-8f
#define SSR_FLT_MAX 3.402823466e+38f

MIFX_D bool is_reflection_sample(float roughness, float depth, float threshold, bool reversed) { return roughness <= threshold && !is_background(depth, reversed); } // SSR_Common.fxh:57-60
MIFX_D float closest_depth(float a, float b, bool reversed) { return reversed ? fmaxf(a, b) : fminf(a, b); } // ClosestDepth, SSR_Common.fxh:6-12

struct SsaoK
{
    float EffectRadius, EffectFalloffRange, RadiusMultiplier, DepthMIPSamplingOffset;
    float TemporalStabilityFactor, SpatialReconstructionRadius;
    int   ResetAccumulation;
    float AlphaInterpolation, BitmaskThickness;
    unsigned Algorithm;
    float SelfOcclusionOffset; // 1e-5, or 5e-3 with SSAO_OPTION_HALF_PRECISION_DEPTH (SSAO_ComputeAmbientOcclusion.fx:145-150)
    float UvScale;     // GetInvViewportSize() / f4ViewportSize.zw: 2 with SSAO_OPTION_HALF_RESOLUTION (SSAO_ComputeAmbientOcclusion.fx:68-75), else 1
    float MipLenSq[4]; // squared pixel distance at which the prefiltered-depth mip switches to level k + 1 (see tap_mip)
};
inline SsaoK make_k(const mifx_ssao_attribs& a, bool halfResolution, bool halfPrecisionDepth = false)
{
    SsaoK k{a.EffectRadius, a.EffectFalloffRange, a.RadiusMultiplier, a.DepthMIPSamplingOffset, a.TemporalStabilityFactor, a.SpatialReconstructionRadius,
            a.ResetAccumulation, a.AlphaInterpolation, a.BitmaskThickness, a.Algorithm, halfPrecisionDepth ? 0.005f : 0.00001f, halfResolution ? 2.0f : 1.0f, {}};
    // point-mip level = floor(clamp(log2(len) - offset, 0, 4) + 0.5) = #{k in 0..3 : log2(len) - offset >= k + 0.5}
    //                 = #{k : len^2 >= 2^(2k + 1 + 2 offset)}
    // (kernels derive level k's threshold from the first: a factor 4 per level is exact in binary floating point)
    k.MipLenSq[0] = float(exp2(1.0 + 2.0 * double(a.DepthMIPSamplingOffset)));
    for (int i = 1; i < 4; ++i) k.MipLenSq[i] = k.MipLenSq[i - 1] * 4.0f;
    return k;
}
} // namespace mifx
