// TEST INFRASTRUCTURE ONLY (oracle/_ref).  A7: SSAO_ComputeResampledHistory.fx (ComputeResampledHistoryPS :56),
// host: ScreenSpaceAmbientOcclusion.cpp:1257-1286; samplers g_TextureDepth = Sam_LinearClamp, g_TextureOcclusion = Sam_PointClamp (:735-736).
#include "ref_common.h"
#ifndef SSAO_OPTION_INVERTED_DEPTH // ref_*_rev.cpp builds the reversed-depth permutation of this file
#define SSAO_OPTION_INVERTED_DEPTH 0
#endif
namespace hlsl { namespace a7 {
#include "ShaderDefinitions.fxh"
#include "SSAO_ComputeResampledHistory.fx"
}}
using namespace hlsl;

// in: 0 AO pyramid (5 mips), 1 depth pyramid (5 mips), 2 history length, 3 normal (c=4); cam0; out[0]: resampled AO
extern "C" int ref_ssao_resampled_history(const ref_args* a)
{
    ref_bind(a7::g_TextureOcclusion.s, a, 0);
    ref_bind(a7::g_TextureDepth.s, a, 1);
    ref_bind(a7::g_TextureHistory.s, a, 2);
    ref_bind(a7::g_TextureNormal.s, a, 3);
    a7::g_TextureDepth_sampler     = SamplerState{true, false, ADDR_CLAMP}; // integer mip levels only
    a7::g_TextureOcclusion_sampler = Sam_PointClamp;
    std::memcpy(&a7::g_Camera, a->cam0, sizeof(a7::CameraAttribs));
    const ref_img& o = a->out[0];
    ref_fullscreen<a7::FullScreenTriangleVSOutput>(o.w, o.h, 0u, [&](a7::FullScreenTriangleVSOutput& vs, int x, int y) { ref_store(o, x, y, a7::ComputeResampledHistoryPS(vs)); });
    return 0;
}
