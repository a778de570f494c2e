// TEST INFRASTRUCTURE ONLY (oracle/_ref).  R6: SSR_ComputeTemporalAccumulation.fx (ComputeTemporalAccumulationPS :224),
// host ScreenSpaceReflection.cpp:1033-1069: g_TextureCurrDepth = PostFX reprojected depth (:1050), previous depth (:1053),
// history ping-pong (:1054-1055), linear-clamp samplers (:688-690), masked.
#include "ref_common.h"
#define SSR_OPTION_INVERTED_DEPTH 0
namespace hlsl { namespace r6 {
#include "ShaderDefinitions.fxh"
#include "SSR_ComputeTemporalAccumulation.fx"
}}
using namespace hlsl;

// in: 0 motion (c=2), 1 hit depth, 2 reprojected depth, 3 curr radiance (c=4), 4 curr variance, 5 prev depth, 6 prev radiance (c=4),
//     7 prev variance, 8 mask; cam0, cam1; attribs.  out: 0 radiance history (c=4), 1 variance history (pre-filled with 0)
extern "C" int ref_ssr_temporal_accumulation(const ref_args* a)
{
    ref_bind(r6::g_TextureMotion.s, a, 0);
    ref_bind(r6::g_TextureHitDepth.s, a, 1);
    ref_bind(r6::g_TextureCurrDepth.s, a, 2);
    ref_bind(r6::g_TextureCurrRadiance.s, a, 3);
    ref_bind(r6::g_TextureCurrVariance.s, a, 4);
    ref_bind(r6::g_TexturePrevDepth.s, a, 5);
    ref_bind(r6::g_TexturePrevRadiance.s, a, 6);
    ref_bind(r6::g_TexturePrevVariance.s, a, 7);
    const ref_img& mask = a->in[8][0];
    r6::g_TexturePrevDepth_sampler = r6::g_TexturePrevRadiance_sampler = r6::g_TexturePrevVariance_sampler = Sam_LinearClamp;
    std::memcpy(&r6::g_CurrCamera, a->cam0, sizeof(r6::CameraAttribs));
    std::memcpy(&r6::g_PrevCamera, a->cam1, sizeof(r6::CameraAttribs));
    std::memcpy(&r6::g_SSRAttribs, a->attribs, sizeof(r6::ScreenSpaceReflectionAttribs));
    const ref_img &o0 = a->out[0], &o1 = a->out[1];
    ref_fullscreen<r6::FullScreenTriangleVSOutput>(o0.w, o0.h, 0u, [&](r6::FullScreenTriangleVSOutput& vs, int x, int y) {
        if (mask.data[size_t(y) * mask.w + x] == 0.0f) return;
        r6::PSOutput r = r6::ComputeTemporalAccumulationPS(vs);
        ref_store(o0, x, y, r.Radiance);
        ref_store(o1, x, y, r.Variance);
    });
    return 0;
}
