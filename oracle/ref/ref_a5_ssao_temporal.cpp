// TEST INFRASTRUCTURE ONLY (oracle/_ref).  A5: SSAO_ComputeTemporalAccumulation.fx (ComputeTemporalAccumulationPS :151),
// host: ScreenSpaceAmbientOcclusion.cpp:1032-1073 -- g_TextureCurrDepth = PostFX reprojected depth (:1053), g_TexturePrevDepth = previous
// depth (:1054), g_TextureMotion = closest motion (:1055); both targets cleared to 1.0 (:1059-1068).
#include "ref_common.h"
#ifndef SSAO_OPTION_INVERTED_DEPTH // ref_*_rev.cpp builds the reversed-depth permutation of this file
#define SSAO_OPTION_INVERTED_DEPTH 0
#endif
namespace hlsl { namespace a5 {
#include "ShaderDefinitions.fxh"
#include "SSAO_ComputeTemporalAccumulation.fx"
}}
using namespace hlsl;

// in: 0 curr AO, 1 prev AO history, 2 prev history length, 3 reprojected depth, 4 prev depth, 5 closest motion; cam0, cam1; attribs
// out: 0 AO history, 1 history length (both pre-filled with 1)
extern "C" int ref_ssao_temporal_accumulation(const ref_args* a)
{
    ref_bind(a5::g_TextureCurrOcclusion.s, a, 0);
    ref_bind(a5::g_TexturePrevOcclusion.s, a, 1);
    ref_bind(a5::g_TextureHistory.s, a, 2);
    ref_bind(a5::g_TextureCurrDepth.s, a, 3);
    ref_bind(a5::g_TexturePrevDepth.s, a, 4);
    ref_bind(a5::g_TextureMotion.s, a, 5);
    std::memcpy(&a5::g_CurrCamera, a->cam0, sizeof(a5::CameraAttribs));
    std::memcpy(&a5::g_PrevCamera, a->cam1, sizeof(a5::CameraAttribs));
    std::memcpy(&a5::g_SSAOAttribs, a->attribs, sizeof(a5::ScreenSpaceAmbientOcclusionAttribs));
    const ref_img& o0 = a->out[0];
    const ref_img& o1 = a->out[1];
    ref_fullscreen<a5::FullScreenTriangleVSOutput>(o0.w, o0.h, 0u, [&](a5::FullScreenTriangleVSOutput& vs, int x, int y) {
        a5::PSOutput r = a5::ComputeTemporalAccumulationPS(vs);
        if (!g_ctx.discarded) { ref_store(o0, x, y, r.Occlusion); ref_store(o1, x, y, r.History); }
    });
    return 0;
}
