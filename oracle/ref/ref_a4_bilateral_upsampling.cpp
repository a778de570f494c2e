// TEST INFRASTRUCTURE ONLY (oracle/_ref).  A4: SSAO_ComputeBilateralUpsampling.fx (ComputeBilateralUpsamplingPS :73), host ScreenSpaceAmbientOcclusion.cpp:985-1008
// (FEATURE_FLAG_HALF_RESOLUTION only); g_TextureDepth = the input depth, linear CLAMP where linear depth sampling is supported (:614), g_TextureOcclusion linear CLAMP (:615).
#include "ref_common.h"
#ifndef SSAO_OPTION_INVERTED_DEPTH
#define SSAO_OPTION_INVERTED_DEPTH 0
#endif
namespace hlsl { namespace a4 {
#include "ShaderDefinitions.fxh"
#include "SSAO_ComputeBilateralUpsampling.fx"
}}
using namespace hlsl;

// in: 0 depth (full resolution), 1 occlusion (half resolution); cam0; attribs; out[0]: upsampled occlusion (full resolution)
extern "C" int ref_ssao_bilateral_upsampling(const ref_args* a)
{
    ref_bind(a4::g_TextureDepth.s, a, 0);
    ref_bind(a4::g_TextureOcclusion.s, a, 1);
    a4::g_TextureDepth_sampler = a4::g_TextureOcclusion_sampler = Sam_LinearClamp;
    std::memcpy(&a4::g_Camera, a->cam0, sizeof(a4::CameraAttribs));
    std::memcpy(&a4::g_SSAOAttribs, a->attribs, sizeof(a4::ScreenSpaceAmbientOcclusionAttribs));
    const ref_img& o = a->out[0];
    ref_fullscreen<a4::FullScreenTriangleVSOutput>(o.w, o.h, 0u, [&](a4::FullScreenTriangleVSOutput& vs, int x, int y) { ref_store(o, x, y, a4::ComputeBilateralUpsamplingPS(vs)); });
    return 0;
}
