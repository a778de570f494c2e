// TEST INFRASTRUCTURE ONLY (oracle/_ref).  R7: SSR_ComputeBilateralCleanup.fx (ComputeBilateralCleanupPS :49), host
// ScreenSpaceReflection.cpp:1071-1104 (target cleared to 0 :1099, masked).  ddx/ddy of the camera Z (:57) are evaluated with the
// 2x2-quad two-phase emulation of hlsl_shim.h (fine derivatives; helper lanes of masked-out quad pixels still execute, as on hardware).
#include "ref_common.h"
#ifndef SSR_OPTION_INVERTED_DEPTH // ref_*_rev.cpp builds the reversed-depth permutation of this file
#define SSR_OPTION_INVERTED_DEPTH 0
#endif
namespace hlsl { namespace r7 {
#include "ShaderDefinitions.fxh"
#include "SSR_ComputeBilateralCleanup.fx"
}}
using namespace hlsl;

// in: 0 depth, 1 normal (c=4), 2 roughness, 3 radiance history (c=4), 4 variance history, 5 mask; cam0; attribs; out[0]: SSR output (c=4, pre-filled with 0)
extern "C" int ref_ssr_bilateral_cleanup(const ref_args* a)
{
    ref_bind(r7::g_TextureDepth.s, a, 0);
    ref_bind(r7::g_TextureNormal.s, a, 1);
    ref_bind(r7::g_TextureRoughness.s, a, 2);
    ref_bind(r7::g_TextureRadiance.s, a, 3);
    ref_bind(r7::g_TextureVariance.s, a, 4);
    const ref_img& mask = a->in[5][0];
    std::memcpy(&r7::g_Camera, a->cam0, sizeof(r7::CameraAttribs));
    std::memcpy(&r7::g_SSRAttribs, a->attribs, sizeof(r7::ScreenSpaceReflectionAttribs));
    const ref_img& o = a->out[0];
    const int W = o.w, H = o.h;
#pragma omp parallel for schedule(dynamic, 2)
    for (int qy = 0; qy < (H + 1) / 2; ++qy)
        for (int qx = 0; qx < (W + 1) / 2; ++qx)
        {
            for (int phase = 0; phase < 2; ++phase)
                for (int lane = 0; lane < 4; ++lane)
                {
                    // lanes outside the image replicate the nearest pixel (helper invocations)
                    int x = std::min(qx * 2 + (lane & 1), W - 1), y = std::min(qy * 2 + (lane >> 1), H - 1);
                    r7::FullScreenTriangleVSOutput vs;
                    vs.f4PixelPos     = float4(float(x) + 0.5f, float(y) + 0.5f, 0.0f, 1.0f);
                    vs.f2NormalizedXY = float2(2.0f * (float(x) + 0.5f) / float(W) - 1.0f, 1.0f - 2.0f * (float(y) + 0.5f) / float(H));
                    vs.uInstID        = 0u;
                    g_ctx.discarded  = false;
                    g_ctx.quad_phase = phase;
                    g_ctx.quad_lane  = lane;
                    g_ctx.call_idx   = 0;
                    float4 r = r7::ComputeBilateralCleanupPS(vs);
                    bool real = (qx * 2 + (lane & 1) < W) && (qy * 2 + (lane >> 1) < H);
                    if (phase == 1 && real && mask.data[size_t(y) * mask.w + x] != 0.0f) ref_store(o, x, y, r);
                }
            g_ctx.quad_phase = -1;
        }
    return 0;
}
