// TEST INFRASTRUCTURE ONLY (oracle/_ref).  A2: SSAO_ComputePrefilteredDepthBuffer.fx (ComputePrefilteredDepthBufferPS :79),
// host: ScreenSpaceAmbientOcclusion.cpp:843-959 (SRV path: previous mip bound as g_TextureLastMip; mip 0 = CopyTextureDepth :864).
#include "ref_common.h"
#define SSAO_OPTION_INVERTED_DEPTH 0
#define SUPPORTED_SHADER_SRV 1
namespace hlsl { namespace a2 {
#include "ShaderDefinitions.fxh"
#include "SSAO_ComputePrefilteredDepthBuffer.fx"
}}
using namespace hlsl;

// in[0]: previous mip (1 level); cam0; attribs: ScreenSpaceAmbientOcclusionAttribs; out[0]: next mip
extern "C" int ref_ssao_prefiltered_depth_mip(const ref_args* a)
{
    ref_bind(a2::g_TextureLastMip.s, a, 0);
    std::memcpy(&a2::g_Camera, a->cam0, sizeof(a2::CameraAttribs));
    std::memcpy(&a2::g_SSAOAttribs, a->attribs, sizeof(a2::ScreenSpaceAmbientOcclusionAttribs));
    const ref_img& o = a->out[0];
    ref_fullscreen<a2::FullScreenTriangleVSOutput>(o.w, o.h, unsigned(a->ival[0]), [&](a2::FullScreenTriangleVSOutput& vs, int x, int y) { ref_store(o, x, y, a2::ComputePrefilteredDepthBufferPS(vs)); });
    return 0;
}
extern "C" int ref_sizeof_ssao_attribs() { return int(sizeof(a2::ScreenSpaceAmbientOcclusionAttribs)); }
